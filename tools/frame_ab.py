"""Interleaved A/B of DeferredFrame options in ONE process on the 4K / 256-light bench frame (ms per frame, HIP events over
back-to-back frames): python tools/frame_ab.py   — currently: fused_exposure (average + tone-map as one launch) on / off."""
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
g = scene.make_global(cam, W, H, sh_pack=sh, delta_time=1.0 / 60.0)
lights = synth.lights_in_view_box(256, cam)
gb = synth.gbuffer_tile(0, 0, W, H, W, H)
frames = {}
for name, fused in (("two launches", False), ("one launch", True)):
    fr = DeferredFrame(ctx, tile_for_rank(0, 1, W, H), g, lights, lut, 512, env, 512, 5, fused_exposure=fused)
    fr.upload_gbuffer(gb)
    fr.set_prev_luminance(0.18)
    frames[name] = fr
for fr in frames.values():
    for _ in range(300):
        fr.render()
torch.cuda.synchronize()
for rnd in range(3):
    for name, fr in frames.items():
        for _ in range(20):
            fr.render()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            fr.render()
        e1.record()
        e1.synchronize()
        print(f"round {rnd}: average + tone-map as {name}: {e0.elapsed_time(e1) / 200:.4f} ms/frame", flush=True)
a, b = frames["two launches"], frames["one launch"]
assert torch.equal(a.ldr, b.ldr) and float(a.avg.cpu()[0]) == float(b.avg.cpu()[0]), "the two variants rendered different frames"
print("same LDR image and adapted luminance after the same number of frames")
