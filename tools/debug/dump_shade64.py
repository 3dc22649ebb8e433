"""debug: dump the GPU fp32 + fp16 shade of the 64x64 / 256-light golden scene (full roughness range)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, common
from oracle import binding as orc
from direct12pbrrenderer_amd.api import PbrContext
ctx = PbrContext(0)
sky, env, lut, sh = common.small_ibl(orc)
def dev_half(a): return ctx.upload(np.ascontiguousarray(a, dtype=np.float16).view(np.uint16)).view(torch.float16)
cam, g, lights, gb, tile = common.shade_scene(64, 64, 256, sh)
cl = orc.cluster_build(g); orc.cluster_cull(g, lights, cl)
gbd = {k: ctx.upload(v) for k, v in gb.items()}
envp = ctx.env_pad(dev_half(env), 16, 5)
o32 = ctx.zeros((64, 64, 4), torch.float32)
ctx.deferred_shade_f32(g, tile, gbd, 64, dev_half(lut), 32, envp, 16, 5, ctx.upload(cl), ctx.upload(lights), len(lights), o32, 64)
o16 = ctx.zeros((64, 64, 4), torch.float16)
ctx.deferred_shade(g, tile, gbd, 64, dev_half(lut), 32, envp, 16, 5, ctx.upload(cl), ctx.upload(lights), len(lights), o16, 64)
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "shade64.npz"), f32=o32.cpu().numpy(), f16=o16.cpu().view(torch.int16).numpy())
print("saved")
