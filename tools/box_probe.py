"""What differs between two boxes of the pool when the same shade takes 0.336 ms on one and 0.353 ms on the other?  Prints, for THIS box: the
shader clock and issue rate of packed fp32 under a SUSTAINED load (back-to-back pbr_valubench launches, 3-6 ms each, sampled along ~150 ms), the
same for plain fp32, then the 4K / 256-light shade's time in a settled frame loop and the clock-normalised figure.  python tools/box_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from direct12pbrrenderer_amd import scene, synth  # noqa: E402
from direct12pbrrenderer_amd.api import PbrContext  # noqa: E402
from direct12pbrrenderer_amd.pipeline import DeferredFrame, tile_for_rank  # noqa: E402

ctx = PbrContext(0)
cus = torch.cuda.get_device_properties(0).multi_processor_count
blocks = cus * 5
st = torch.zeros((blocks * 4, 4), dtype=torch.int64, device="cuda")


def sustained(op, iters, launches):
    out = []
    for k in range(launches):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ctx.valubench(op, blocks, iters, st)
        e1.record()
        e1.synchronize()
        s = st.cpu().numpy()
        clock = float(np.median((s[:, 1] - s[:, 0]) / np.maximum(s[:, 3] - s[:, 2], 1)) * 100e6)
        out.append((e0.elapsed_time(e1), clock / 1e9, blocks * 4 * iters * 8 / (e0.elapsed_time(e1) * 1e-3) / 1e9))
    return out


for name, op, iters in (("packed v_pk_fma_f32", 2, 60000), ("plain v_mul_f32", 0, 60000)):
    r = sustained(op, iters, 30)
    pick = [0, 2, 5, 10, 20, 29]
    print(f"{name}: launch ms / clock GHz / Ginst/s after k launches: " + "  ".join(f"k={k}: {r[k][0]:.2f} / {r[k][1]:.3f} / {r[k][2]:.0f}" for k in pick), flush=True)

lut, env, sh = bench.build_ibl(ctx)
W, H = 3840, 2160
cam = scene.Camera.reference_default(W, H)
g = scene.make_global(cam, W, H, sh_pack=sh)
fr = DeferredFrame(ctx, tile_for_rank(0, 1, W, H), g, synth.lights_in_view_box(256, cam), lut, 512, env, 512, 5)
fr.upload_gbuffer(synth.gbuffer_tile(0, 0, W, H, W, H))
fr.set_prev_luminance(0.18)
for _ in range(400):
    fr.render()
torch.cuda.synchronize()
ev = []
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(200):
    pair = [] if i % 5 == 0 else None
    fr.render(pair)
    if pair:
        ev += pair
e1.record()
e1.synchronize()
shade = sum(a.elapsed_time(b) for a, b in ev) / len(ev)
print(f"frame {e0.elapsed_time(e1) / 200:.4f} ms, shade in frame {shade:.4f} ms", flush=True)
# the clock right after the frame loop (the chip in the thermal / power state the frames left it in)
r = sustained(2, 60000, 3)
print(f"packed clock right after the frames: {r[0][1]:.3f} / {r[2][1]:.3f} GHz -> shade = {shade * 1e-3 * r[2][1] * 1e9 * cus * 4 / (W * H):.1f} SIMD-cycles per pixel", flush=True)
ctx.close()
