"""world_size-2 and -3 rehearsal of the multi-GPU path on CPU (gloo): tiling with aprons + the 256-bin
histogram all-reduce.  The per-rank arithmetic is the ORACLE here (no GPU in this
container); what is under test is the host-side sharding logic that bench.py uses on the GPUs:
tile_for_rank / TileSpec, global-pixel addressing, interior histogram, all-reduce, full-frame
PixelCount, identical average luminance on every rank."""
import os
import socket
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import common

TILE_W, TILE_H, APRON, N_LIGHTS = 512, 64, 256, 64


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _render_rank(orc, spec, ibl):
    """Oracle pipeline on one rank's extended tile; returns interior HDR (post-bloom) and its histogram."""
    from direct12pbrrenderer_amd.structs import Tile
    sky, env, lut, sh = ibl
    cam, g, lights, gb, _ = common.shade_scene(spec.ew, spec.eh, N_LIGHTS, sh, full=(spec.full_w, spec.full_h),
                                               x0=spec.ex0, y0=spec.ey0, rough_min=48, coverage_mask=False)
    cl = orc.cluster_build(g)
    orc.cluster_cull(g, lights, cl)
    hdr, _ = orc.deferred_shade(g, Tile(spec.ex0, spec.ey0, spec.ew, spec.eh, spec.full_w, spec.full_h), gb, lut, env,
                                common.ENV_SIZE, common.ENV_MIPS, cl, lights)
    orc.bloom(hdr)
    interior = np.ascontiguousarray(hdr[spec.iy:spec.iy + spec.h, spec.ix:spec.ix + spec.w])
    return g, interior, orc.lum_histogram(interior)


def _worker(rank, world, port, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import binding as orc
    from direct12pbrrenderer_amd.pipeline import tile_for_rank
    orc.set_num_threads(2)
    spec = tile_for_rank(rank, world, TILE_W, TILE_H, APRON)
    g, interior, hist = _render_rank(orc, spec, common.small_ibl(orc))
    t = torch.from_numpy(hist.view(np.int32).copy())
    dist.all_reduce(t)                                    # int32 sum == uint32 sum bit for bit
    hist_all = t.numpy().view(np.uint32).copy()
    avg = orc.lum_average(hist_all.copy(), spec.full_w * spec.full_h, float(g.DeltaTime), 0.18)
    ldr = orc.tonemap(interior, avg)
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), interior=interior, hist=hist, hist_all=hist_all, avg=np.float32(avg), ldr=ldr,
             rect=np.array([spec.x0, spec.y0, spec.w, spec.h]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [2, 3])      # 3: the middle rank carries an apron on both sides (the N >= 3 layout of bench.py)
def test_tiling_matches_single_frame(orc, ibl, world):
    from direct12pbrrenderer_amd.pipeline import TileSpec, tile_for_rank
    if world == 3:
        mid = tile_for_rank(1, 3, TILE_W, TILE_H, APRON)
        assert (mid.ex0, mid.ew, mid.ix) == (TILE_W - APRON, TILE_W + 2 * APRON, APRON)
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(world, _free_port(), d), nprocs=world, join=True)
        ranks = [dict(np.load(os.path.join(d, f"rank{r}.npz"))) for r in range(world)]
    W, H = TILE_W * world, TILE_H
    g, full, hist_full = _render_rank(orc, TileSpec(0, 0, W, H, W, H, 0), ibl)
    avg_full = orc.lum_average(hist_full.copy(), W * H, float(g.DeltaTime), 0.18)
    # every rank holds the same all-reduced histogram = sum of the per-rank ones, and the same average
    for r in ranks[1:]:
        assert np.array_equal(ranks[0]["hist_all"], r["hist_all"]) and ranks[0]["avg"] == r["avg"]
    assert np.array_equal(ranks[0]["hist_all"], sum(r["hist"] for r in ranks))
    assert ranks[0]["hist_all"].sum() == W * H
    # apron sufficiency: interiors equal the single-frame result (fp32 uv rounding may flip an fp16 ulp)
    for r in ranks:
        x0, y0, w, h = r["rect"]
        ref = full[y0:y0 + h, x0:x0 + w]
        d = common.half_ulp_diff(r["interior"], ref)
        assert d.max() <= 2 and (d > 0).mean() < 2e-3, (d.max(), (d > 0).mean())
    assert np.abs(ranks[0]["hist_all"].astype(np.int64) - hist_full.astype(np.int64)).sum() <= 8
    assert float(ranks[0]["avg"]) == pytest.approx(avg_full, rel=1e-6)
    ldr_full = orc.tonemap(full, avg_full)
    for r in ranks:
        x0, y0, w, h = r["rect"]
        a, b = r["ldr"], ldr_full[y0:y0 + h, x0:x0 + w]
        for k in range(3):
            assert np.abs(((a >> (8 * k)) & 255).astype(np.int32) - ((b >> (8 * k)) & 255).astype(np.int32)).max() <= 1


def test_apron_too_small_is_visible(orc, ibl):
    """Negative control: without an apron the tile seam shows up in the bloomed interior."""
    from direct12pbrrenderer_amd.pipeline import TileSpec
    W, H = 256, 64
    _, full, _ = _render_rank(orc, TileSpec(0, 0, W, H, W, H, 0), ibl)
    _, left, _ = _render_rank(orc, TileSpec(0, 0, W // 2, H, W, H, 0), ibl)
    assert common.half_ulp_diff(left, full[:, : W // 2]).max() > 2
