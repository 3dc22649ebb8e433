"""bench.py's N = 8 arithmetic without a GPU: the grids, tiles and halo plans the driver's 8-GPU launch will use.  (The launch itself
cannot be rehearsed on the one-GPU pool — more than 6 processes holding one card open are killed there, launcher included — so what is specific to N = 8 is pinned
here, the per-rank arithmetic of a 2x4 grid in tests/test_multigpu_gloo.py, and bench.py's orchestration as a 2x2 rehearsal in
tests/test_gpu_bench.py.)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_bench_n8_weak_and_cfg5_layouts():
    import bench
    from direct12pbrrenderer_amd.pipeline import grid_for_world, halo_plan, tile_for_rank, tile_of_frame
    # the weak-scaling headline at N = 8: a 16:9 frame of 8 x 8.3 Mpixel cut 4 x 2 (cols x rows), one 4K-equivalent tile per rank
    cols, rows = grid_for_world(8, None)
    assert (cols, rows) == (4, 2)
    tw, th = bench.weak_tile(8, cols, rows, 3840, 2160)
    assert (tw, th) == (2720, 3056) and tw % 16 == 0 and th % 16 == 0
    assert abs(tw * th / (3840 * 2160) - 1.0) < 0.01                       # per-GPU work fixed as N grows ("weak")
    assert abs((cols * tw) / (rows * th) - 16 / 9) < 0.02                   # the assembled frame stays 16:9
    # BASELINE configs[4]: 7680x4320 "tiled 2x4" = 2 rows of 4 tiles of 1920x2160
    assert bench.CFG5_GRID[8] == (4, 2) and bench.CFG5_FRAME == (7680, 4320)
    for make, fw, fh in ((lambda r, halo: tile_for_rank(r, 8, tw, th, layout=(cols, rows), halo=halo), cols * tw, rows * th),
                         (lambda r, halo: tile_of_frame(r, 8, 7680, 4320, layout=bench.CFG5_GRID[8], halo=halo), 7680, 4320)):
        for halo in (False, True):
            specs = [make(r, halo) for r in range(8)]
            cover = np.zeros((fh // 16, fw // 16), dtype=np.int32)          # every interior pixel owned exactly once (16-px granularity)
            for s in specs:
                assert s.full_w == fw and s.full_h == fh and s.w % 16 == 0 and s.h % 16 == 0
                cover[s.y0 // 16:(s.y0 + s.h) // 16, s.x0 // 16:(s.x0 + s.w) // 16] += 1
            assert (cover == 1).all()
            if halo:
                plans = [halo_plan(r, 8, specs) for r in range(8)]
                # what rank a sends to rank b is what b expects from a (both derive the rectangles from the same specs: no sizes travel)
                sends = {(r, pr): send for r, p in enumerate(plans) for (pr, send, _recv) in p if send}
                recvs = {(pr, r): recv for r, p in enumerate(plans) for (pr, _send, recv) in p if recv}
                assert sends == recvs and len(sends) >= 2 * (3 * 2 + 4)     # at least the 10 shared edges of a 4x2 grid, both ways
                for r, s in enumerate(specs):                               # the strips a rank receives + its interior tile E exactly
                    e = np.zeros(((s.ey1 - s.ey0) // 2, (s.ex1 - s.ex0) // 2), dtype=np.int32)
                    e[(s.y0 - s.ey0) // 2:(s.y0 + s.h - s.ey0) // 2, (s.x0 - s.ex0) // 2:(s.x0 + s.w - s.ex0) // 2] += 1
                    for (_pr, _send, recv) in plans[r]:
                        if recv:
                            e[recv[1] - s.ey0 // 2:recv[3] - s.ey0 // 2, recv[0] - s.ex0 // 2:recv[2] - s.ex0 // 2] += 1
                    assert (e == 1).all()

def test_bench_refuses_a_rehearsal_beyond_five_ranks():
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1"], capture_output=True, text=True, cwd=ROOT,
                       env=dict(os.environ, CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES=""))
    assert r.returncode == 2 and "at most 5 ranks" in r.stderr and r.stdout.strip() == ""


def _dry_run(extra, timeout=600):
    import json
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dry-run", "--steps", "2", "--warmup", "1", "--settle", "2",
                        "--deadline", "120"] + extra, capture_output=True, text=True, cwd=ROOT, timeout=timeout,
                       env=dict(os.environ, CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES=""))
    lines = [ln for ln in r.stdout.split("\n") if ln.strip()]
    return r, lines, (json.loads(lines[-1]) if lines else None)


def test_bench_n8_launch_dry_run_on_cpu():
    """`bench.py --gpus 8 --dry-run`: the 8-rank launch with every device call replaced by a host stand-in (bench_dryrun.py) — the real
    spawn_ranks / torch.distributed.run start, rendezvous, unique-id broadcast, 2x4 grids, halo plans through DeferredFrame, verification
    frames (the all-reduce counts every pixel of the frame once; every level-1 texel of every extended tile arrives with its sender's
    checksum), timed loops with their barriers, the overlapped-tail check, the cfg5 sub-record with its single-GPU denominator, teardown
    and the ONE JSON line — everything of the launch that has never run on hardware, but the kernels.  No device is visible."""
    r, lines, d = _dry_run([])
    assert r.returncode == 0, r.stderr[-3000:]
    assert len(lines) == 1, lines
    assert d["dry_run"] is True and d["n_gpus"] == 8 and d["steps"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    c = d["config"]
    assert c["layout_rows_x_cols"] == "2x4" and c["frame"] == [10880, 6112] and c["tile"] == [0, 0, 2720, 3056]
    assert c["bloom_borders"] == "halo" and "rccl-capi" in c["collectives"] and "dry run" in c["collectives"]
    assert "ms_per_step_tail_overlapped" in c          # the overlapped-tail variant was checked against the plain order and timed
    c5 = c["cfg5"]
    assert "error" not in c5, c5
    assert c5["frame"] == [7680, 4320] and c5["layout_rows_x_cols"] == "2x4" and c5["tile"] == [0, 0, 1920, 2160] and c5["scaling"] == "strong"
    assert c5["single_gpu_ms_per_step"] > 0 and "speedup_vs_single_gpu" in c5
    assert "oracle" not in r.stderr.lower()           # the product path of the launch does not touch the checker


def test_bench_n8_dry_run_every_rank_takes_the_next_candidate_when_one_rank_fails_setup():
    """A rank that cannot set a candidate up (here: rank 3, halo exchange through the C ABI's communicators) must not leave the others
    inside that candidate's collectives: all eight agree and move to the next one (halo through torch.distributed) — for the headline
    and for the cfg5 sub-record — and the launch ends with exit code 0 and a note in the record, instead of hanging until a deadline."""
    r, lines, d = _dry_run(["--dry-fail", "3:halo/capi"])
    assert r.returncode == 0, r.stderr[-3000:]
    assert len(lines) == 1
    for c in (d["config"], d["config"]["cfg5"]):
        assert "rccl-torch" in c["collectives"] and c["bloom_borders"] == "halo" and c["layout_rows_x_cols"] == "2x4"
        assert any("halo/capi could not be set up" in n for n in c["notes"])


def test_bench_n8_dry_run_failed_verification_sends_every_rank_to_the_next_candidate():
    """One rank's tile counts one pixel too many in the first verification frame: the all-reduced histogram then disagrees with the frame's
    pixel count on EVERY rank, all eight drop the candidate together (no rank is left inside its collectives) and the next one passes."""
    r, lines, d = _dry_run(["--dry-fail", "5:miscount", "--no-cfg5"])
    assert r.returncode == 0, r.stderr[-3000:]
    assert len(lines) == 1 and "rccl-torch" in d["config"]["collectives"]
    assert any("failed verification" in n and "histogram counts 66498561 pixels, frame has 66498560" in n for n in d["config"]["notes"])


def test_bench_n8_dry_run_a_rank_that_never_arrives_ends_the_launch_at_the_deadline():
    """A rank that never reaches a collective (here: rank 5 sleeps in front of its first histogram all-reduce) cannot be waited for: every
    phase with collectives runs under the watchdog's deadline, each rank gives up on its own, the launch ends with a non-zero exit code
    within the deadline (+ start-up) instead of hanging, and no half-written record is taken for a result."""
    import time
    t0 = time.monotonic()
    r, lines, d = _dry_run(["--dry-fail", "5:hang", "--deadline", "15", "--no-cfg5"], timeout=240)
    assert r.returncode != 0 and time.monotonic() - t0 < 180
    assert "did not finish before its deadline" in r.stderr and lines == []
