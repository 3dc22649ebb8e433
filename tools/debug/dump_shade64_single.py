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
def run(clm):
    o32 = ctx.zeros((64, 64, 4), torch.float32)
    ctx.deferred_shade_f32(g, tile, gbd, 64, dev_half(lut), 32, envp, 16, 5, ctx.upload(clm), ctx.upload(lights), len(lights), o32, 64)
    return o32.cpu().numpy()[27, 25, :3]
x, y, ci = 25, 27, 1804
base = run(cl)
c0 = cl.copy(); c0["NumLights"][:] = 0
ibl = run(c0)
print("gpu total", base, "ibl", ibl)
w0, w032 = orc.deferred_shade(g, tile, gb, lut, env, 16, 5, c0, lights, want_f32=True)
print("orc ibl", w032[y, x, :3])
cc = cl[ci]
for i in range(cc["NumLights"]):
    clm = c0.copy(); clm[ci]["NumLights"] = 1; clm[ci]["LightIndex"][0] = cc["LightIndex"][i]
    gpu = run(clm) - ibl
    t, t32 = orc.deferred_shade(g, tile, gb, lut, env, 16, 5, clm, lights, want_f32=True)
    o = t32[y, x, :3] - w032[y, x, :3]
    if np.abs(o - gpu).max() > 2e-5:
        print("light", cc["LightIndex"][i], "slot", i, "gpu", gpu, "orc", o)
# pairs as the kernel sees them
for i in range(0, cc["NumLights"], 2):
    clm = c0.copy(); clm[ci]["NumLights"] = 2; clm[ci]["LightIndex"][0] = cc["LightIndex"][i]; clm[ci]["LightIndex"][1] = cc["LightIndex"][i + 1]
    gpu = run(clm) - ibl
    t, t32 = orc.deferred_shade(g, tile, gb, lut, env, 16, 5, clm, lights, want_f32=True)
    o = t32[y, x, :3] - w032[y, x, :3]
    if np.abs(o - gpu).max() > 2e-5: print("pair", cc["LightIndex"][i], cc["LightIndex"][i + 1], "gpu", gpu, "orc", o)

# prefixes of the real list
for k in range(1, cc["NumLights"] + 1):
    clm = c0.copy(); clm[ci]["NumLights"] = k; clm[ci]["LightIndex"][:] = cc["LightIndex"]
    gpu = run(clm) - ibl
    t, t32 = orc.deferred_shade(g, tile, gb, lut, env, 16, 5, clm, lights, want_f32=True)
    o = t32[y, x, :3] - w032[y, x, :3]
    print("prefix", k, "last", cc["LightIndex"][k - 1], "gpu", gpu, "orc", o)
