"""Runs one stage of a frame a few times (for rocprofv3 --pmc passes).  usage: tools/profile_stage.py shade|bloom|all [n_lights [W H]]
(default: the 4K / 256-light bench frame; `shade 1 1920 1080` = BASELINE cfg2 with the reference scene's light_1)"""
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
W, H = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (3840, 2160)
spec = tile_for_rank(0, 1, W, H)
cam = scene.Camera.reference_default(W, H)
g = scene.make_global(cam, W, H, sh_pack=sh)
fr = DeferredFrame(ctx, spec, g, synth.reference_scene_light() if n == 1 else synth.lights_in_view_box(n, cam), lut, 512, env, 512, 5)
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
