"""cfg3 timing: GGX prefilter of the 512^2 sky (5 mips x 1024 spp), per mip: python tools/cfg3_ms.py [label]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import bench  # noqa: E402
from direct12pbrrenderer_amd import synth  # noqa: E402
from direct12pbrrenderer_amd.api import PbrContext  # noqa: E402
from direct12pbrrenderer_amd.structs import ENV_MIPS  # noqa: E402

ctx = PbrContext(0)
sky_mips = 10
sky = ctx.upload(synth.env_cube(512, sky_mips))
ctx.cube_gen_mips(sky, 512, sky_mips)
envbuf = ctx.prefilter_env(sky, 512, sky_mips, 512, ENV_MIPS)
ms = bench.time_stage(lambda: ctx.prefilter_env(sky, 512, sky_mips, 512, ENV_MIPS, out=envbuf), 5)
sky_h = sky.half().float()   # every texel a half value (what BC6H_UF16 assets decode to): the exact half-precision copy is sampled
ms_h = bench.time_stage(lambda: ctx.prefilter_env(sky_h, 512, sky_mips, 512, ENV_MIPS, out=envbuf), 5)
print(f"{sys.argv[1] if len(sys.argv) > 1 else ''}: prefilter 512^2 x 5 mips x 1024 spp: fp32 source {ms:.3f} ms, half-representable source {ms_h:.3f} ms", flush=True)
