"""pbr_prefilter_env (table-driven kernels) against the per-dispatch sequential kernel and the oracle on small cubes, per mip:
where do they differ, and by how many fp16 ULPs?  Diagnostic for the cube-face tie rule.  python tools/prefilter_small_diag.py"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import common
from direct12pbrrenderer_amd import synth
from direct12pbrrenderer_amd.api import PbrContext
from direct12pbrrenderer_amd.structs import cube_mip_offset
from oracle import binding as orc
ctx = PbrContext(0)
h = lambda t: t.cpu().view(torch.int16).numpy().view(np.float16)
for size in (16, 32, 64):
    mips = int(np.log2(size)) + 1
    sky = synth.env_cube(size, mips); orc.cube_gen_mips(sky, size, mips)
    for name, chain in (("fp32", sky), ("half", sky.astype(np.float16).astype(np.float32))):
        d = ctx.upload(chain)
        fast = h(ctx.prefilter_env(d, size, mips, size, 5)); seq = h(ctx.prefilter_env_dispatches(d, size, mips, size, 5)); want = orc.prefilter_env(chain, size, mips, size, 5)
        for m in range(5):
            a, b = cube_mip_offset(size, m), cube_mip_offset(size, m + 1)
            s = size >> m
            df = common.half_ulp_diff(fast[a:b, :3], want[a:b, :3]).max(axis=1); ds = common.half_ulp_diff(seq[a:b, :3], want[a:b, :3]).max(axis=1)
            rel = (np.abs(fast[a:b, :3].astype(np.float32) - want[a:b, :3].astype(np.float32)) / np.maximum(np.abs(want[a:b, :3].astype(np.float32)), 1e-9)).max(axis=1)
            bad = np.nonzero((df > 1) & (rel > 1e-3))[0]
            where = [(int(t // (s * s)), int((t // s) % s), int(t % s)) for t in bad[:6]]
            print(f"size {size:3d} {name} mip {m} ({s}x{s}): table-driven vs oracle max {int(df.max())} ulp (rel {rel.max():.2e}), {len(bad)} texels beyond 1 ulp AND 1e-3 {where}; sequential vs oracle max {int(ds.max())} ulp", flush=True)
ctx.close()
