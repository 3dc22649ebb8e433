"""debug: dump the GPU fp32 shade of the bench-IBL bands (+ the GPU-built LUT/env) for offline analysis against the oracle."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, common, bench
from oracle import binding as orc
from direct12pbrrenderer_amd.api import PbrContext
ctx = PbrContext(0)
lut_d, env_d, sh = bench.build_ibl(ctx)
lut = lut_d.cpu().view(torch.int16).numpy().view(np.float16)
env = env_d.cpu().view(torch.int16).numpy().view(np.float16)
out = {"lut": lut, "env": env, "sh": sh}
for (w, h, rows) in ((1920, 1080, 32), (3840, 2160, 32)):
    y0 = (h - rows) // 2 // 8 * 8
    cam, g, lights, gb, tile = common.shade_scene(w, rows, 256, sh, full=(w, h), x0=0, y0=y0, rough_min=48, coverage_mask=False)
    cl = orc.cluster_build(g); orc.cluster_cull(g, lights, cl)
    gbd = {k: ctx.upload(v) for k, v in gb.items()}
    o = ctx.zeros((rows, w, 4), torch.float32)
    envp = ctx.env_pad(env_d, 512, 5)
    ctx.deferred_shade_f32(g, tile, gbd, w, lut_d, 512, envp, 512, 5, ctx.upload(cl), ctx.upload(lights), len(lights), o, w)
    out[f"got_{w}"] = o.cpu().numpy()
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "band_f32.npz"), **out)
print("saved")
