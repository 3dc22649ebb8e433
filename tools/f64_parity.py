"""GPU shade (fp32 probe, pbr_deferred_shade_f32) and the fp32 oracle, each against the DOUBLE-precision truth interval
(oracle/pbr_oracle_f64.cpp): error distributions in units of the image's scale, and the per-pixel criterion
|gpu - f64| <= 1e-4 * scale + 4 * |oracle_f32 - f64|.  Test infrastructure (loads oracle/): run on the GPU box.
usage: python tools/f64_parity.py [out.json]"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import common  # noqa: E402
from direct12pbrrenderer_amd.api import PbrContext  # noqa: E402
from oracle import binding as orc  # noqa: E402

ctx = PbrContext(0)


def dev_half(a):
    return ctx.upload(np.ascontiguousarray(a, dtype=np.float16).view(np.uint16)).view(torch.float16)


def to_np_half(t):
    return t.cpu().view(torch.int16).numpy().view(np.float16)


def gpu_f32(g, tile, gb, dlut, lut_res, envp, env_size, env_mips, cl, lights):
    h, w = gb["A"].shape
    gbd = {k: ctx.upload(v) for k, v in gb.items()}
    out = ctx.zeros((h, w, 4), torch.float32)
    ctx.deferred_shade_f32(g, tile, gbd, w, dlut, lut_res, envp, env_size, env_mips, ctx.upload(cl), ctx.upload(lights) if len(lights) else None, len(lights), out, w)
    ctx.sync()
    return out.cpu().numpy()


def report(name, g, tile, gb, lut, env, env_size, env_mips, lights, dlut, envp):
    cl = orc.cluster_build(g)
    orc.cluster_cull(g, lights, cl)
    _, o32, sens = orc.deferred_shade(g, tile, gb, lut, env, env_size, env_mips, cl, lights, want_f32=True, want_sens=True)
    lo, hi, fl = orc.deferred_shade_f64(g, tile, gb, lut, env, env_size, env_mips, cl, lights)
    got = gpu_f32(g, tile, gb, dlut, lut.shape[0], envp, env_size, env_mips, cl, lights)
    ok = fl == 0
    scale = float(np.abs(hi[ok]).max())
    dg = orc.truth_distance(got, lo, hi)[ok].max(axis=-1) / scale
    do = orc.truth_distance(o32, lo, hi)[ok].max(axis=-1) / scale
    dgo = np.abs(got[..., :3].astype(np.float64) - o32[..., :3])[ok].max(axis=-1) / scale
    crit = dg <= 1e-4 + 4.0 * do
    q = lambda a: {"max": float(a.max()), "q99.99": float(np.quantile(a, .9999)), "q99.9": float(np.quantile(a, .999)), "q99": float(np.quantile(a, .99)),
                   "median": float(np.median(a)), "n_above_1e-4": int((a > 1e-4).sum())}
    r = {"case": name, "pixels": int(ok.sum()), "flagged": int((fl > 1).sum()), "scale": scale, "gpu_vs_f64": q(dg), "oracle_f32_vs_f64": q(do), "gpu_vs_oracle_f32": q(dgo),
         "criterion_failures": int((~crit).sum()), "worst_ratio": float((dg / (1e-4 + 4.0 * do)).max())}
    bad = np.argsort(dg)[-3:][::-1]
    r["worst_pixels"] = [{"gpu": float(dg[i]), "oracle": float(do[i]), "sens_allow": float(8 * 2.0 ** -24 * sens[0][ok][i].max() / scale)} for i in bad]
    print(json.dumps(r), flush=True)
    return r


out = []
sky, env, lut, sh = common.small_ibl(orc)
dlut, envp = dev_half(lut), ctx.env_pad(dev_half(env), common.ENV_SIZE, common.ENV_MIPS)
for nl in (0, 1, 256, 1024):
    for rm in (0, 48):
        cam, g, lights, gb, tile = common.shade_scene(64, 64, nl, sh, rough_min=rm)
        out.append(report(f"64x64, {nl} lights, rough_min {rm}, test IBL", g, tile, gb, lut, env, common.ENV_SIZE, common.ENV_MIPS, lights, dlut, envp))
cam, g, lights, gb, tile = common.shade_scene(200, 37, 256, sh, full=(640, 360), x0=328, y0=91, rough_min=0)
out.append(report("ragged 200x37 tile of 640x360, 256 lights", g, tile, gb, lut, env, common.ENV_SIZE, common.ENV_MIPS, lights, dlut, envp))

lut_d, env_d, sh_b = bench.build_ibl(ctx)
lut_b, env_b = to_np_half(lut_d), to_np_half(env_d)
envp_b = ctx.env_pad(env_d, 512, 5)
for (w, h, rows) in ((1920, 1080, 32), (3840, 2160, 32), (7680, 4320, 16)):
    y0 = (h - rows) // 2 // 8 * 8
    cam, g, lights, gb, tile = common.shade_scene(w, rows, 256, sh_b, full=(w, h), x0=0, y0=y0, rough_min=48, coverage_mask=False)
    out.append(report(f"{w}x{rows} band of {w}x{h}, 256 lights, bench IBL", g, tile, gb, lut_b, env_b, 512, 5, lights, lut_d, envp_b))
rng = np.random.default_rng(12345)
for k in range(6):   # soak: random tiles of the 4K frame, full roughness range on every other one
    x0, y0 = int(rng.integers(0, 3840 - 512)) // 8 * 8, int(rng.integers(0, 2160 - 128)) // 8 * 8
    cam, g, lights, gb, tile = common.shade_scene(512, 128, 256, sh_b, full=(3840, 2160), x0=x0, y0=y0, rough_min=48 if k % 2 == 0 else 0, coverage_mask=False)
    out.append(report(f"soak tile {k}: 512x128 at ({x0},{y0}) of 3840x2160, rough_min {48 if k % 2 == 0 else 0}", g, tile, gb, lut_b, env_b, 512, 5, lights, lut_d, envp_b))
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
ctx.close()
