"""debug: where does the 1024-light 256x144 shade differ from the oracle?"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, common
from oracle import binding as orc
from direct12pbrrenderer_amd.api import PbrContext
ctx = PbrContext(0)
sky, env, lut, sh = common.small_ibl(orc)
def dev_half(a): return ctx.upload(np.ascontiguousarray(a, dtype=np.float16).view(np.uint16)).view(torch.float16)
for nl in (1024, 600, 300, 256):
    cam, g, lights, gb, tile = common.shade_scene(256, 144, nl, sh, rough_min=48)
    cl = orc.cluster_build(g); orc.cluster_cull(g, lights, cl)
    want, w32, sens = orc.deferred_shade(g, tile, gb, lut, env, 16, 5, cl, lights, want_f32=True, want_sens=True)
    gbd = {k: ctx.upload(v) for k, v in gb.items()}
    out = ctx.zeros((144, 256, 4), torch.float32)
    envp = ctx.env_pad(dev_half(env), 16, 5)
    ctx.deferred_shade_f32(g, tile, gbd, 256, dev_half(lut), 32, envp, 16, 5, ctx.upload(cl), ctx.upload(lights), len(lights), out, 256)
    got = out.cpu().numpy()
    on = gb["stencil"] > 0
    err = np.abs(got[..., :3] - w32[..., :3]) * on[..., None]
    scale = np.abs(w32[on][:, :3]).max()
    allow = 1e-4 * scale + 8 * 2.0**-24 * sens
    r = err / allow
    y, x, c = np.unravel_index(np.argmax(r), r.shape)
    print(f"n_lights {nl}: scale {scale:.4f} worst ratio {r.max():.3f} at ({x},{y}) ch {c}: got {got[y,x,:3]} want {w32[y,x,:3]} sens {sens[y,x]} ")
    print("   A %08x B %08x C %08x depth %.7f" % (gb["A"][y,x], gb["B"][y,x], gb["C"][y,x], gb["depth"][y,x]))
    bad = (r.max(axis=2) > 1)
    print("   pixels over the bound:", int(bad.sum()), "of", int(on.sum()), "; rel L-inf", err.max()/scale)
    ys, xs = np.nonzero(bad)
    for yy, xx in list(zip(ys, xs))[:8]:
        print("     ", xx, yy, got[yy, xx, :3], w32[yy, xx, :3], "rough", gb["C"][yy,xx] & 255, "metal", (gb["C"][yy,xx] >> 8) & 255)
