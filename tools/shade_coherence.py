"""Upper bound of what binning pixels by cluster could buy the shade: the same 4K / 256-light
workload on G-buffers whose surface samples are shared by cell x cell pixel patches.  cell = 1 is the BASELINE workload
(every pixel its own random depth: the 64 lanes of a wave fall into ~8 z-slices, the wave walks the LONGEST of their
lists).  cell = 64 gives every wave one cluster and one light list (lane-uniform LDS reads, no wave-max waste) without
any of the costs a real binning pass would add (LDS exchange of pixel data, partially filled waves)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from direct12pbrrenderer_amd import scene, synth  # noqa: E402
from direct12pbrrenderer_amd.api import PbrContext  # noqa: E402
from direct12pbrrenderer_amd.pipeline import DeferredFrame, tile_for_rank  # noqa: E402

ctx = PbrContext(0)
lut, env, sh = bench.build_ibl(ctx)
W, H = 3840, 2160
cam = scene.Camera.reference_default(W, H)
g = scene.make_global(cam, W, H, sh_pack=sh)
lights = synth.lights_in_view_box(256, cam)
frames = {}
for cell in (1, 16, 64, 256):
    fr = DeferredFrame(ctx, tile_for_rank(0, 1, W, H), g, lights, lut, 512, env, 512, 5)
    gb = synth.gbuffer_tile(0, 0, W, H, W, H, cell=cell)
    fr.upload_gbuffer(gb)
    fr.clustered()
    frames[cell] = (fr, bench.mean_lights_per_pixel(g, gb, fr.spec, fr.clusters))
for _ in range(200):           # clocks up before anything is timed
    frames[1][0].shade()
for rnd in range(3):           # three interleaved rounds: a drifting clock shows up as a spread, not as a trend over cells
    for cell, (fr, ll) in frames.items():
        ms = bench.time_stage(fr.shade, 50)
        print(f"round {rnd} cell {cell:3d}: shade {ms:.4f} ms; mean list length per pixel {ll:.2f}", flush=True)
