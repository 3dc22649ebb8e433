"""Runs one stage of the 4K frame a few times (for rocprofv3 --pmc passes).  usage: tools/profile_stage.py shade|bloom|all [n_lights]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from direct12pbrrenderer_amd import scene, synth
from direct12pbrrenderer_amd.api import PbrContext
from direct12pbrrenderer_amd.pipeline import DeferredFrame, tile_for_rank
import bench
stage = sys.argv[1] if len(sys.argv) > 1 else "shade"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
ctx = PbrContext(0)
lut, env, sh = bench.build_ibl(ctx)
W, H = 3840, 2160
spec = tile_for_rank(0, 1, W, H)
cam = scene.Camera.reference_default(W, H)
g = scene.make_global(cam, W, H, sh_pack=sh)
fr = DeferredFrame(ctx, spec, g, synth.lights_in_view_box(n, cam), lut, 512, env, 512, 5)
fr.upload_gbuffer(synth.gbuffer_tile(0, 0, W, H, W, H))
fr.set_prev_luminance(0.18)
fr.render()
torch.cuda.synchronize()
for _ in range(5):
    if stage == "shade":
        fr.shade()
    elif stage == "bloom":
        fr.bloom()
    else:
        fr.render()
torch.cuda.synchronize()
