#!/usr/bin/env python3
"""Generates tests/golden/golden_v1.npz from the CPU oracle (oracle/pbr_oracle.cpp).

PARITY UNPINNED: the reference ships no golden vector for this path, so
these fixtures pin the ORACLE's behaviour at the commit that generated them: they catch drift in
the oracle or in the synthetic-input generators, and the GPU tests compare the HIP path with
them on the same seeded inputs.  Inputs are regenerated from seeds (tests/common.py); only
expected outputs are stored.

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import common  # noqa: E402
from direct12pbrrenderer_amd import synth  # noqa: E402
from oracle import binding as orc  # noqa: E402

LUT_ROWS = {256: [0, 100, 255], 512: [0, 255, 511]}


def crc(a):
    return np.uint32(zlib.crc32(np.ascontiguousarray(a).tobytes()))


def main():
    out = {}
    sky, env, lut, sh = common.small_ibl(orc)
    out["sky_crc"] = crc(sky)
    out["lut32"] = lut
    for res, rows in LUT_ROWS.items():
        for r in rows:
            out[f"lut{res}_row{r}"] = orc.brdf_lut_rows(res, r, 1)[0]
    out["env16"] = env
    out["sh16"] = sh
    for n in (0, 1, 256):
        cam, g, lights, gb, tile = common.shade_scene(64, 64, n, sh)
        cl = orc.cluster_build(g)
        orc.cluster_cull(g, lights, cl)
        hdr, _ = orc.deferred_shade(g, tile, gb, lut, env, common.ENV_SIZE, common.ENV_MIPS, cl, lights)
        out[f"shade64_l{n}"] = hdr
        if n == 256:
            out["clusters_l256_numlights"] = cl["NumLights"].copy()
            out["clusters_l256_crc"] = crc(cl["LightIndex"][np.arange(32)[None, :] < cl["NumLights"][:, None]])
            out["gbuffer_crc"] = crc(np.stack([gb["A"], gb["B"], gb["C"]]))
    img = synth.hdr_noise_image(128, 72)
    out["bloom_in_crc"] = crc(img)
    hdr = img.copy()
    a, b = orc.bloom(hdr)
    out["bloom_chain_a"] = a
    out["bloom_chain_b"] = b
    out["bloom_hdr"] = hdr
    hist = orc.lum_histogram(hdr)
    out["hist"] = hist.copy()
    out["avg_bin"] = np.float32(orc.lum_average_bin(hist, 128 * 72))
    avg = orc.lum_average(hist, 128 * 72, 1.0 / 60.0, 0.18)
    out["avg"] = np.float32(avg)
    out["ldr"] = orc.tonemap(hdr, avg)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden_v1.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", len(out), "arrays")


if __name__ == "__main__":
    main()
