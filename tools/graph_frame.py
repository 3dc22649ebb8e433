"""Experiment: the in-order frame (12 launches) captured ONCE in a HIP graph and replayed, against the same frame enqueued launch by
launch (ms per frame, 200 frames each, interleaved), back to back and with a fence wait after every frame (the reference's loop).
python tools/graph_frame.py [W H [n_lights]]      (default 3840 2160 256; `1440 960 8` = the reference's operating point)"""
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
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3840, 2160)
NL = int(sys.argv[3]) if len(sys.argv) > 3 else 256
cam = scene.Camera.reference_default(W, H)
g = scene.make_global(cam, W, H, sh_pack=sh, delta_time=1.0 / 60.0)
fr = DeferredFrame(ctx, tile_for_rank(0, 1, W, H), g, synth.lights_in_view_box(NL, cam), lut, 512, env, 512, 5)
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


def fenced(fn):
    def f():
        fn()
        torch.cuda.synchronize()
    return f


for rnd in range(3):
    a = timed(fr.render)
    b = timed(graph.replay)
    c = timed(fenced(fr.render))
    d = timed(fenced(graph.replay))
    print(f"round {rnd} {W}x{H}/{NL}: back to back: launch by launch {a:.4f} ms/frame, graph replay {b:.4f}; fence per frame: launch by launch {c:.4f}, graph replay {d:.4f}", flush=True)
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
