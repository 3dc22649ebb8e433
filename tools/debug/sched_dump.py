"""dump the shaded HDR buffer of one tile for the current schedule (knobs build): python tools/debug/sched_dump.py out.npy W H [FW FH] [lights] [repeat]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from direct12pbrrenderer_amd import scene, synth
from direct12pbrrenderer_amd.api import PbrContext
from direct12pbrrenderer_amd.pipeline import DeferredFrame, TileSpec
out, W, H = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
FW, FH = (int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (W, H)
n = int(sys.argv[6]) if len(sys.argv) > 6 else 256
rep = int(sys.argv[7]) if len(sys.argv) > 7 else 3
ctx = PbrContext(0)
lut, env, sh = bench.build_ibl(ctx)
cam = scene.Camera.reference_default(FW, FH)
g = scene.make_global(cam, FW, FH, sh_pack=sh)
spec = TileSpec(0, 0, W, H, FW, FH, 0)
fr = DeferredFrame(ctx, spec, g, synth.reference_scene_light() if n == 1 else synth.lights_in_view_box(n, cam), lut, 512, env, 512, 5)
fr.upload_gbuffer(synth.gbuffer_tile(spec.sx0, spec.sy0, spec.sw, spec.sh, FW, FH))
fr.clustered()
res = []
for _ in range(rep):
    fr.hdr.zero_()
    fr.shade()
    torch.cuda.synchronize()
    res.append(fr.hdr.view(torch.int16).cpu().numpy().copy())
for i in range(1, rep):
    print("repeat", i, "differs from repeat 0 in", int((res[i] != res[0]).any(-1).sum()), "pixels")
np.save(out, res[0])
