"""the failing leg of test_deferred_shade_full_4k_frame_is_linear_in_the_light_colours, dumped: python f32_tile_repro.py out_prefix"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np, torch
import bench, common
from direct12pbrrenderer_amd.api import PbrContext
from direct12pbrrenderer_amd.structs import Tile
ctx = PbrContext(0)
lut_d, env_d, sh = bench.build_ibl(ctx)
W, H = 3840, 2160
cam, g, lights, gb, tile = common.shade_scene(W, H, 256, sh, rough_min=48, coverage_mask=False)
from direct12pbrrenderer_amd import pipeline
cld = ctx.zeros((3072 * 39,), torch.int32)
ctx.clustered(g, ctx.upload(lights), len(lights), cld)
gbd = {k: ctx.upload(v) for k, v in gb.items()}
envp = ctx.env_pad(env_d, 512, 5)
def shade(t, sub, hh, f32):
    out = ctx.zeros((hh, W, 4), torch.float32 if f32 else torch.float16)
    if f32:
        ctx.deferred_shade_f32(g, t, sub, W, lut_d, 512, envp, 512, 5, cld, ctx.upload(lights), len(lights), out, W)
    else:
        ctx.deferred_shade(g, t, sub, W, lut_d, 512, envp, 512, 5, cld, ctx.upload(lights), len(lights), out, W)
    ctx.sync()
    return out
for f32 in (True, False):
    B = shade(tile, gbd, H, f32)
    for (y0, hh) in ((0, 1080), (1080, 1080)):
        t = Tile(0, y0, W, hh, W, H)
        sub = {k: ctx.upload(np.ascontiguousarray(v[y0:y0 + hh])) for k, v in gb.items()}
        out = shade(t, sub, hh, f32)
        d = (out[..., :3] != B[y0:y0 + hh, :, :3]).any(-1)
        print("f32" if f32 else "f16", "tile", y0, "differs from the whole frame in", int(d.sum()), "pixels; zero pixels in tile:", int((out[..., :3] == 0).all(-1).sum()),
              "in frame:", int((B[y0:y0 + hh, :, :3] == 0).all(-1).sum()))
        if d.any():
            ys, xs = torch.nonzero(d, as_tuple=True)
            rows = torch.bincount(ys, minlength=hh).cpu().numpy()
            nz = np.nonzero(rows)[0]
            print("  rows with differences:", len(nz), "first", nz[:24].tolist(), "counts", rows[nz[:24]].tolist())
            print("  per 256-col block:", torch.bincount(xs // 256, minlength=15).cpu().tolist())
        np.save(f"{sys.argv[1]}_{'f32' if f32 else 'f16'}_{y0}.npy", out.cpu().numpy().view(np.uint32 if f32 else np.uint16))
    np.save(f"{sys.argv[1]}_{'f32' if f32 else 'f16'}_whole.npy", B.cpu().numpy().view(np.uint32 if f32 else np.uint16))
