"""One-off soak of the shade's parity beyond the committed cases: random tiles of several frame sizes x light counts, the fp32
probe against the oracle with the test suite's own checker (tests/test_gpu_parity.py::_check_shade_f32).
usage (GPU box): python tools/debug/parity_soak.py [n_cases]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402,F401
import common  # noqa: E402
import test_gpu_parity as T  # noqa: E402
from oracle import binding as orc  # noqa: E402
from direct12pbrrenderer_amd.api import PbrContext  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
ctx = PbrContext(0)
sky, env, lut, sh = common.small_ibl(orc)
dlut, denv = T.dev_half(ctx, lut), ctx.env_pad(T.dev_half(ctx, env), common.ENV_SIZE, common.ENV_MIPS)
rng = np.random.default_rng(20261004)
frames = [(1920, 1080), (3840, 2160), (1280, 720), (2560, 1440), (7680, 4320), (640, 360)]
worst = 0.0
for case in range(n_cases):
    fw, fh = frames[case % len(frames)]
    w, h = int(rng.integers(40, 320)), int(rng.integers(8, 96))
    w, h = min(w, fw), min(h, fh)
    x0, y0 = int(rng.integers(0, fw - w + 1)), int(rng.integers(0, fh - h + 1))
    n_lights = [7, 256, 1024, 64][case % 4]
    cam, g, lights, gb, tile = common.shade_scene(w, h, n_lights, sh, full=(fw, fh), x0=x0, y0=y0, rough_min=int(rng.integers(0, 64)))
    cl = orc.cluster_build(g)
    orc.cluster_cull(g, lights, cl)
    _, want_f32, sens = orc.deferred_shade(g, tile, gb, lut, env, common.ENV_SIZE, common.ENV_MIPS, cl, lights, want_f32=True, want_sens=True)
    got = T._shade_f32_on_gpu(ctx, g, tile, gb, dlut, lut.shape[0], denv, common.ENV_SIZE, common.ENV_MIPS, cl, lights)
    rel, frac = T._check_shade_f32(got, want_f32, sens, gb["stencil"], f"case {case}: {w}x{h}+{x0}+{y0} of {fw}x{fh}, {n_lights} lights")
    worst = max(worst, rel)
    print(f"case {case:2d}: {w:3d}x{h:2d} at ({x0},{y0}) of {fw}x{fh}, {n_lights:4d} lights: plain relative L-inf {rel:.3g}, {frac * 100:.3f} % of pixels above 1e-4 (inside the conditioning allowance)", flush=True)
print(f"all {n_cases} cases inside the bound; worst plain relative L-inf {worst:.3g}")
