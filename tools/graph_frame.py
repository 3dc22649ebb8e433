"""Experiment: the in-order 4K frame (12 launches) captured ONCE in a HIP graph and replayed, against the same frame enqueued launch by
launch (ms per frame, 200 frames each, interleaved).  python tools/graph_frame.py"""
import os
import sys
import time

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
fr = DeferredFrame(ctx, tile_for_rank(0, 1, W, H), g, synth.lights_in_view_box(256, cam), lut, 512, env, 512, 5)
fr.upload_gbuffer(synth.gbuffer_tile(0, 0, W, H, W, H))
fr.set_prev_luminance(0.18)
for _ in range(300):
    fr.render()
torch.cuda.synchronize()
ref_avg = None

side = torch.cuda.Stream()
graph = torch.cuda.CUDAGraph()
with torch.cuda.stream(side):
    ctx.bind_torch_stream()
    fr.render()                      # warm the capture stream
    torch.cuda.synchronize()
    with torch.cuda.graph(graph, stream=side):
        ctx.bind_torch_stream()
        fr.render()
ctx.bind_torch_stream()
torch.cuda.synchronize()


def timed(fn, n=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for rnd in range(3):
    a = timed(fr.render)
    b = timed(graph.replay)
    print(f"round {rnd}: launch by launch {a:.4f} ms/frame, graph replay {b:.4f} ms/frame", flush=True)
# same frames: adapted luminance after N more frames either way
fr.set_prev_luminance(0.18); fr.hist.zero_(); torch.cuda.synchronize()
for _ in range(5):
    fr.render()
torch.cuda.synchronize(); x = (float(fr.avg.cpu()[0]), int(fr.ldr.to(torch.int64).sum()))
fr.set_prev_luminance(0.18); fr.hist.zero_(); torch.cuda.synchronize()
for _ in range(5):
    graph.replay()
torch.cuda.synchronize(); y = (float(fr.avg.cpu()[0]), int(fr.ldr.to(torch.int64).sum()))
print("same frames:", x == y)
