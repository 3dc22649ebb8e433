"""One measurement of the 4K / 256-light shade (ms per launch: isolated launches and 50 back-to-back), for sweeps
over environment knobs (each setting needs its own process): python tools/shade_ms.py [label]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from direct12pbrrenderer_amd import scene, synth  # noqa: E402
from direct12pbrrenderer_amd.api import PbrContext  # noqa: E402
from direct12pbrrenderer_amd.pipeline import DeferredFrame, tile_for_rank  # noqa: E402

ctx = PbrContext(0)
lut, env, sh = bench.build_ibl(ctx)
W, H = 3840, 2160
cam = scene.Camera.reference_default(W, H)
g = scene.make_global(cam, W, H, sh_pack=sh)
fr = DeferredFrame(ctx, tile_for_rank(0, 1, W, H), g, synth.lights_in_view_box(256, cam), lut, 512, env, 512, 5)
fr.upload_gbuffer(synth.gbuffer_tile(0, 0, W, H, W, H))
fr.clustered()
iso = bench.time_stage(fr.shade, 30)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
best = 1e9
for _ in range(3):
    e0.record()
    for _ in range(50):
        fr.shade()
    e1.record()
    e1.synchronize()
    best = min(best, e0.elapsed_time(e1) / 50)
frame = bench.time_stage(fr.render, 30)
fr.shade()
b_nohist = bench.time_stage(fr.bloom, 30, pre=fr.shade)
b_hist = bench.time_stage(fr.bloom_histogram, 30, pre=fr.shade)
print(f"   bloom {b_nohist:.4f} ms, bloom+histogram {b_hist:.4f} ms (each incl. nothing else; shade re-run before every sample)", flush=True)
print(f"{sys.argv[1] if len(sys.argv) > 1 else ''}: shade isolated {iso:.4f} ms, back-to-back {best:.4f} ms, frame {frame:.4f} ms", flush=True)
