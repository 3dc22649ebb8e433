#!/usr/bin/env python3
"""Generates tests/golden/golden_v2.npz from the CPU oracle: the full-size IBL fixtures SURVEY 8c asks for.

  * LUT 256^2 and 512^2 (precompute_brdf.hlsl): CRC32 of the whole oracle plane + 64 sampled texels each;
  * prefiltered env at cfg3's real size (512^2 x 5 mips x 1024 spp, env_map_gen.hlsl): 4096 seeded random texels
    spread over the five mips (oracle/orc_prefilter_env_texels), from the bench's synthetic sky;
  * SH9 pack of that 512^2 sky (deterministic quadrature).

PARITY UNPINNED, like golden_v1: the reference holds no fixture for this path, so these pin the ORACLE at the commit
that generated them; the GPU tests compare the HIP kernels with them and with the live oracle on the same inputs.
Inputs are regenerated from seeds (direct12pbrrenderer_amd/synth.py); only indices and expected outputs are stored.

Run from the repo root:  python tests/golden/make_golden_v2.py        (about a minute on 8 cores)
"""
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from direct12pbrrenderer_amd import synth  # noqa: E402
from oracle import binding as orc  # noqa: E402

ENV_SIZE, ENV_MIPS, SKY_MIPS = 512, 5, 10
TEXELS_PER_MIP = [1024, 1024, 1024, 512, 512]      # 4096 in all


def crc(a):
    return np.uint32(zlib.crc32(np.ascontiguousarray(a).tobytes()))


def bench_sky():
    """The sky bench.py builds: synth.env_cube(512) with 2x2 box mips (oracle's cube_gen_mips == pbr_cube_gen_mips)."""
    sky = synth.env_cube(ENV_SIZE, SKY_MIPS)
    orc.cube_gen_mips(sky, ENV_SIZE, SKY_MIPS)
    return sky


def prefilter_indices():
    rng = np.random.default_rng(0x5EED0040)
    return [np.sort(rng.choice(6 * (ENV_SIZE >> m) ** 2, size=n, replace=False)).astype(np.uint32) for m, n in enumerate(TEXELS_PER_MIP)]


def lut_samples(res):
    rng = np.random.default_rng(0x5EED0041 + res)
    idx = rng.choice(res * res, size=64, replace=False).astype(np.uint32)
    idx[:4] = [0, res - 1, res * (res - 1), res * res - 1]      # the four corners ride along
    return idx


def main():
    out = {}
    for res in (256, 512):
        lut = orc.brdf_lut(res)
        out[f"lut{res}_crc"] = crc(lut)
        idx = lut_samples(res)
        out[f"lut{res}_idx"] = idx
        out[f"lut{res}_texels"] = lut.reshape(-1, 2)[idx]
    sky = bench_sky()
    out["sky512_crc"] = crc(sky)
    for m, idx in enumerate(prefilter_indices()):
        out[f"env512_m{m}_idx"] = idx
        out[f"env512_m{m}_texels"] = orc.prefilter_env_texels(sky, ENV_SIZE, SKY_MIPS, ENV_SIZE, ENV_MIPS, m, idx)
    out["sh512"] = orc.sh9_project(sky, ENV_SIZE)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden_v2.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", len(out), "arrays")


if __name__ == "__main__":
    main()
