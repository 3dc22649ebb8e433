"""Tile + apron / tile + halo path on real hardware: two, three and four (2x2) ranks (all on cuda:0 — the test box has one GPU; the
collective is gloo on the host copy of the 256-bin histogram) against the single-GPU frame.
What runs on the GPU is exactly what bench.py runs per rank: DeferredFrame on an apron-extended
tile with global-pixel addressing, interior histogram, full-frame PixelCount."""
import os
import socket
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import common

pytestmark = pytest.mark.gpu
N_LIGHTS = 256


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _frame(ctx, spec, ibl_dev, sh, allreduce=None):
    from direct12pbrrenderer_amd.pipeline import DeferredFrame
    lut, env = ibl_dev
    cam, g, lights, gb, _ = common.shade_scene(spec.ew, spec.eh, N_LIGHTS, sh, full=(spec.full_w, spec.full_h),
                                               x0=spec.ex0, y0=spec.ey0, rough_min=48, coverage_mask=False)
    fr = DeferredFrame(ctx, spec, g, lights, lut, common.LUT_RES, env, common.ENV_SIZE, common.ENV_MIPS, allreduce=allreduce)
    fr.upload_gbuffer(gb)
    fr.set_prev_luminance(0.18)
    fr.render()
    ctx.sync()
    return fr


def _ibl_dev(ctx, ibl):
    sky, env, lut, sh = ibl
    up = lambda a: ctx.upload(np.ascontiguousarray(a, dtype=np.float16).view(np.uint16)).view(torch.float16)
    return up(lut), up(env)


def _worker(rank, world, port, outdir, layout, halo, tile_w, tile_h, overlap=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from direct12pbrrenderer_amd.api import PbrContext
    from direct12pbrrenderer_amd.pipeline import DeferredFrame, HaloTransport, tile_for_rank
    from oracle import binding as orc
    ibl = common.small_ibl(orc)
    ctx = PbrContext(0)

    def allreduce(hist):   # host-side gloo stand-in for the RCCL all-reduce of bench.py
        ctx.sync()
        t = hist.cpu()
        dist.all_reduce(t)
        hist.copy_(t)

    specs = [tile_for_rank(r, world, tile_w, tile_h, layout=layout, halo=halo) for r in range(world)]
    spec = specs[rank]
    lut, env = _ibl_dev(ctx, ibl)
    cam, g, lights, gb, _ = common.shade_scene(spec.sw, spec.sh, N_LIGHTS, ibl[3], full=(spec.full_w, spec.full_h),
                                               x0=spec.sx0, y0=spec.sy0, rough_min=48, coverage_mask=False)
    fr = DeferredFrame(ctx, spec, g, lights, lut, common.LUT_RES, env, common.ENV_SIZE, common.ENV_MIPS, allreduce=allreduce,
                       all_specs=specs, rank=rank, halo_transport=HaloTransport("host", dist) if halo else None, overlap=overlap)
    assert (fr.split is not None) == overlap
    fr.upload_gbuffer(gb)
    fr.set_prev_luminance(0.18)
    if halo:
        fr.level1.fill_(777.0)   # poison: every level-1 texel of E must come from the prefilter or a neighbour
    fr.render()
    ctx.sync()
    if halo:
        assert not bool((fr.level1 == 777.0).any()), "halo exchange left level-1 texels unfilled"
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), hdr=fr.hdr_interior().view(np.uint16), ldr=fr.ldr_numpy(),
             avg=fr.avg.cpu().numpy(), rect=np.array([spec.x0, spec.y0, spec.w, spec.h]))
    ctx.close()
    dist.barrier()
    dist.destroy_process_group()


# world 3: the middle rank has a neighbour (and an apron) on both sides; 2x2: aprons on two axes incl. the corner
# (the way BASELINE cfg5 tiles its 8K frame); halo: level-1 strips from the neighbours instead of the shaded apron
# overlap: the border ring is shaded first, its level-1 strips travel while the core is shaded (needs tiles > 2 x 264 px)
CASES = [(2, None, False, 512, 288, False), (3, None, False, 512, 288, False), (4, (2, 2), False, 384, 288, False),
         (4, (2, 2), True, 384, 288, False), (2, None, True, 512, 288, False), (4, (2, 2), True, 1024, 640, True),
         (3, None, True, 832, 400, True)]


# the last two cases: the ranks load the knobs build with PBR_BLOOM_WIDE=1, so every 2x-up level of their (small) tiles runs the
# POLYPHASE kernel — its merge rectangle / buffer origin / interior-only histogram (pbr_bloom_tiled, pbr_bloom_histogram with a
# rect), which the frame sizes of this file would otherwise leave to k_blur_hv; the single frame they are compared with runs in this
# process on the product library (shader-order kernels at this size): <= 2 fp16 ULP as everywhere in this file
CASES += [(2, None, True, 512, 288, "poly"), (4, (2, 2), False, 384, 288, "poly")]


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world,layout,halo,tile_w,tile_h,overlap", CASES)
def test_ranks_match_single_gpu_frame(ctx, ibl, world, layout, halo, tile_w, tile_h, overlap):
    from direct12pbrrenderer_amd.pipeline import TileSpec, grid_for_world
    cols, rows = grid_for_world(world, layout)
    poly, overlap = overlap == "poly", overlap is True
    saved = {k: os.environ.get(k) for k in ("PBR_HIP_LIB", "PBR_BLOOM_WIDE")}
    if poly:   # spawned ranks inherit the environment; this process has loaded the product library long ago
        os.environ["PBR_HIP_LIB"] = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "direct12pbrrenderer_amd", "libpbr_hip_knobs.so")
        os.environ["PBR_BLOOM_WIDE"] = "1"
    try:
        with tempfile.TemporaryDirectory() as d:
            mp.spawn(_worker, args=(world, _free_port(), d, layout, halo, tile_w, tile_h, overlap), nprocs=world, join=True)
            ranks = [dict(np.load(os.path.join(d, f"rank{r}.npz"))) for r in range(world)]
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    W, H = tile_w * cols, tile_h * rows
    fr = _frame(ctx, TileSpec(0, 0, W, H, W, H, 0), _ibl_dev(ctx, ibl), ibl[3])
    full_hdr = fr.hdr_interior()
    full_ldr = fr.ldr_numpy()
    assert all(r["avg"][0] == ranks[0]["avg"][0] for r in ranks)            # bit-identical exposure on every rank
    assert ranks[0]["avg"][0] == pytest.approx(float(fr.avg.cpu()[0]), rel=1e-6)
    for r in ranks:
        x0, y0, w, h = r["rect"]
        d_ulp = common.half_ulp_diff(r["hdr"].view(np.float16), full_hdr[y0:y0 + h, x0:x0 + w])
        assert d_ulp.max() <= 2 and (d_ulp > 0).mean() < 2e-3, (d_ulp.max(), (d_ulp > 0).mean())
        a, b = r["ldr"], full_ldr[y0:y0 + h, x0:x0 + w]
        for k in range(3):
            assert np.abs(((a >> (8 * k)) & 255).astype(np.int32) - ((b >> (8 * k)) & 255).astype(np.int32)).max() <= 1


_SHRINK = r"""
import hashlib, sys
sys.path.insert(0, %r)
import numpy as np, torch
from direct12pbrrenderer_amd.api import PbrContext
from direct12pbrrenderer_amd.structs import bloom_level_offset
ctx = PbrContext(0)
rng = np.random.default_rng(20261005)
out = []
for case in range(%d):
    ew, eh = 16 * int(rng.integers(8, 80)), 16 * int(rng.integers(8, 80))          # extended tile: multiples of 16, as tile layouts make them (level 4 may be odd-sized)
    # the rank's shaded rectangle inside E and the interior it merges inside that (even origins and sizes, as tile layouts have)
    hx, hy = 2 * int(rng.integers(0, ew // 8)), 2 * int(rng.integers(0, eh // 8))
    hw, hh = 2 * int(rng.integers(8, (ew - hx) // 2 + 1)), 2 * int(rng.integers(8, (eh - hy) // 2 + 1))
    mx, my = hx + 2 * int(rng.integers(0, hw // 8 + 1)), hy + 2 * int(rng.integers(0, hh // 8 + 1))
    mw, mh = 2 * int(rng.integers(4, (hx + hw - mx) // 2 + 1)), 2 * int(rng.integers(4, (hy + hh - my) // 2 + 1))
    hdr_np = (rng.random((hh, hw, 4), dtype=np.float32) * 3.0).astype(np.float16)
    hdr_np[..., 3] = 1.0
    l1 = (rng.random(((eh // 2) * (ew // 2), 4), dtype=np.float32) * 2.0).astype(np.float16)
    l1[:, 3] = 1.0
    hdr = ctx.upload(hdr_np.view(np.uint16)).view(torch.float16)
    A, B = ctx.alloc_bloom_chain(ew, eh), ctx.alloc_bloom_chain(ew, eh)
    B.fill_(777.0)                                                                    # stale chain contents must not reach the merged interior
    o = bloom_level_offset(ew, eh, 1)
    A[o:o + l1.shape[0]] = ctx.upload(l1.view(np.uint16)).view(torch.float16)
    hist = ctx.zeros((256,), torch.int32)
    ctx.bloom_tiled(hdr, hw, (hx, hy, hw, hh), ew, eh, A, B, (mx, my, mw, mh), hist)
    ctx.sync()
    got = hdr.cpu().view(torch.int16).numpy()
    assert int(hist.sum()) == mw * mh, (case, int(hist.sum()), mw * mh)
    out.append(hashlib.sha1(got.tobytes()).hexdigest()[:16] + hashlib.sha1(hist.cpu().numpy().tobytes()).hexdigest()[:8])
print("tiled bloom", " ".join(out))
"""


@pytest.mark.timeout(600)
def test_tiled_bloom_up_pass_rectangles_leave_the_merged_interior_bit_identical():
    """Round 6: pbr_bloom_tiled runs the up-pass of levels 1-3 only on the rectangles the merged interior depends on (bloom.hip,
    bloom_pyramid(need0): need / 2 +- margins per level, clipped, whole tiles).  40 random extended tiles (128 .. 1264 on a side, multiples of 16), shaded
    rectangles and merge rectangles: the HDR buffer and the interior's histogram of the product library equal, bit for bit, those of the
    knobs build with the rectangles switched off (PBR_BLOOM_SHRINK=0: every level on the whole extended tile) — with chain B pre-filled
    with a sentinel, so a level that reads a texel its producer skipped cannot pass by luck.  Own processes (knobs are read once)."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def run(env_extra):
        env = dict(os.environ)
        for k in ("PBR_BLOOM_SHRINK", "PBR_HIP_LIB"):
            env.pop(k, None)
        if env_extra:
            env.update(env_extra)
            env["PBR_HIP_LIB"] = os.path.join(root, "direct12pbrrenderer_amd", "libpbr_hip_knobs.so")
        r = subprocess.run(["timeout", "-k", "10", "500", sys.executable, "-c", _SHRINK % (root, 40)], capture_output=True, text=True, env=env)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("tiled bloom")]
        assert r.returncode == 0 and lines, (r.returncode, r.stdout[-1500:], r.stderr[-3000:])
        return lines[-1].split()

    product, full = run(None), run({"PBR_BLOOM_SHRINK": "0"})
    differing = [i for i, (a, b) in enumerate(zip(product[2:], full[2:])) if a != b]
    assert not differing, f"cases {differing} differ between the shrinking rectangles and the full up-pass"
    assert run({"PBR_BLOOM_SHRINK": "1"}) == product
