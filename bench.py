#!/usr/bin/env python3
"""bench.py — shaded Mpixel/s of the full deferred frame on synthetic 4K G-buffers (BASELINE.json).

A "step" is one frame of the hot path over one rank's tile: cluster build + cull, deferred shade
(256 clustered lights + IBL), the bloom chain (16 reference dispatches, 8 fused launches) with the luminance
histogram in its last kernel (+ RCCL all-reduce when N > 1), average, ACES tone-map — i.e. BASELINE.json configs[3] at N = 1.  Inputs (G-buffer,
lights, LUT, prefiltered env, SH) are resident in HBM before the timed region.  Weak scaling: each
rank owns one 3840x2160 tile of an N x 1 tile frame (tiles side by side: short edges shared) and shades a
256-px apron towards its neighbours so bloom needs no halo exchange; `value` counts interior pixels only.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from direct12pbrrenderer_amd import scene, synth  # noqa: E402
from direct12pbrrenderer_amd.api import PbrContext  # noqa: E402
from direct12pbrrenderer_amd.pipeline import DeferredFrame, tile_for_rank  # noqa: E402
from direct12pbrrenderer_amd.structs import ENV_MIPS, Tile  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
FP32_VALU_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md chip table
# algorithmic bytes per pixel: unique bytes a reference pass must read + write once
BYTES_PER_PX = {"shade": 25.0, "bloom": 75.75, "histogram": 8.0, "tonemap": 12.0}
ENV_SIZE, LUT_RES, N_LIGHTS = 512, 512, 256


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=50)
    p.add_argument("--warmup", type=int, default=5)
    p.add_argument("--width", type=int, default=3840)
    p.add_argument("--height", type=int, default=2160)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-kernel-timing", action="store_true")
    return p.parse_args()


def build_ibl(ctx):
    """One-shot IBL precompute with the HIP kernels (not timed): LUT 512^2, sky 512^2 -> env 5 mips, SH9."""
    sky_mips = int(np.log2(ENV_SIZE)) + 1
    sky = ctx.upload(synth.env_cube(ENV_SIZE, sky_mips))
    ctx.cube_gen_mips(sky, ENV_SIZE, sky_mips)
    lut = ctx.brdf_lut(LUT_RES)
    env = ctx.prefilter_env(sky, ENV_SIZE, sky_mips, ENV_SIZE, ENV_MIPS)
    sh = ctx.sh9_project(sky, ENV_SIZE, sky_mips)
    ctx.sync()
    return lut, env, sh.cpu().numpy()


def pmc_traffic(kernel_prefix, workload_px):
    """HBM-side bytes per launch of `kernel_prefix` from the committed rocprofv3 --pmc summary
    (profiles/pmc_traffic_latest.json, produced by tools/summarize_pmc.py from separate FETCH_SIZE and
    WRITE_SIZE passes).  Unit/correction per MI355X_MICROARCH.md: both counters are in KiB and gfx950's
    FETCH_SIZE tallies 128-B requests at 64 B, so reads are doubled.  None if no matching profile."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic_latest.json")
    try:
        d = json.load(open(path))
    except Exception:
        return None
    if d.get("_workload_pixels") != workload_px:
        return None
    for k, v in d.items():
        if isinstance(v, dict) and kernel_prefix in k and "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            return int((2.0 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024)
    return None


def pmc_valu(kernel_prefix, workload_px):
    """VALU wave-instructions per launch of `kernel_prefix` from the committed SQ counter summary."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "pmc_sq_latest.json")))
    except Exception:
        return None
    if d.get("_workload_pixels") != workload_px:
        return None
    for k, v in d.items():
        if isinstance(v, dict) and kernel_prefix in k and "SQ_INSTS_VALU" in v:
            return v["SQ_INSTS_VALU"]
    return None


def time_stage(fn, iters, pre=None):
    """Average device time of `fn` in ms, HIP events on the stream the kernels run on
    (the ctx is bound to torch's current stream, so torch.cuda.Event sees them)."""
    if pre:
        pre()
    fn()
    torch.cuda.synchronize()
    total = 0.0
    for _ in range(iters):
        if pre:
            pre()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        total += e0.elapsed_time(e1)
    return total / iters


def cpu_baseline(g, lights, lut_np, env_np, width, height, budget_s=12.0):
    """The oracle ("port") on a bounded band of the same workload, all host threads.  A 64-row probe
    sizes the band so the timed sample costs about `budget_s` seconds of CPU work (<= the whole frame)."""
    from oracle import binding as orc

    def run(rows):
        rows = max(16, min(height, rows) // 16 * 16)
        y0 = (height - rows) // 2
        gb = synth.gbuffer_tile(0, y0, width, rows, width, height)
        cl = orc.cluster_build(g)
        orc.cluster_cull(g, lights, cl)
        t0 = time.perf_counter()
        hdr, _ = orc.deferred_shade(g, Tile(0, y0, width, rows, width, height), gb, lut_np, env_np, ENV_SIZE, ENV_MIPS, cl, lights)
        orc.bloom(hdr)
        hist = orc.lum_histogram(hdr)
        avg = orc.lum_average(hist, width * rows, 1.0 / 60.0, 0.18)
        orc.tonemap(hdr, avg)
        return rows, y0, time.perf_counter() - t0

    rows, _, dt = run(64)
    rows, y0, dt = run(int(rows * budget_s / max(dt, 1e-3)))
    reps = 1
    if rows >= height // 16 * 16 and dt < 0.7 * budget_s:   # many-core host: the whole frame is too short a sample, repeat it
        reps = max(1, min(32, int(budget_s / max(dt, 1e-3))))
        t0 = time.perf_counter()
        for _ in range(reps):
            run(rows)
        dt = (time.perf_counter() - t0) / reps
    return {"value": round(width * rows / dt / 1e6, 4), "unit": "Mpixel/s", "cores": orc.num_threads(), "kind": "port",
            "sample": f"{width}x{rows} band (rows {y0}..{y0 + rows - 1}) of the {width}x{height} frame: shade(256 lights+IBL)"
                      f"+bloom+histogram+average+tonemap, oracle/pbr_oracle.cpp with OpenMP on {orc.num_threads()} threads, "
                      f"{reps} x {dt:.2f} s (input synthesis included)"}


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # PBR_BENCH_REHEARSAL=1: every rank on cuda:0 with a gloo collective — exercises this file's
    # multi-rank path on a one-GPU box; not a measurement
    rehearsal = os.environ.get("PBR_BENCH_REHEARSAL", "0") == "1"
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            local_rank = 0
            torch.cuda.set_device(0)
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist = None
        torch.cuda.set_device(0)
    assert a.gpus == world, f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
    dev = local_rank if world > 1 else 0
    ctx = PbrContext(dev)

    spec = tile_for_rank(rank, world, a.width, a.height)
    lut, env, sh = build_ibl(ctx)
    cam = scene.Camera.reference_default(spec.full_w, spec.full_h)
    g = scene.make_global(cam, spec.full_w, spec.full_h, sh_pack=sh, delta_time=1.0 / 60.0)
    lights = synth.lights_in_view_box(N_LIGHTS, cam)

    use_capi_rccl = os.environ.get("PBR_ALLREDUCE", "torch") == "capi"
    allreduce = None
    if world > 1:
        if use_capi_rccl:
            from direct12pbrrenderer_amd.api import comm_unique_id
            ids = [comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(ids, src=0)
            ctx.comm_init(world, rank, ids[0])
            allreduce = ctx.allreduce_hist
        elif rehearsal:
            def allreduce(h):
                t = h.cpu()
                dist.all_reduce(t)
                h.copy_(t)
        else:
            allreduce = lambda h: dist.all_reduce(h)   # RCCL, int32 sum == uint32 sum bit for bit  # noqa: E731

    frame = DeferredFrame(ctx, spec, g, lights, lut, LUT_RES, env, ENV_SIZE, ENV_MIPS, allreduce=allreduce)
    frame.upload_gbuffer(synth.gbuffer_tile(spec.ex0, spec.ey0, spec.ew, spec.eh, spec.full_w, spec.full_h))
    frame.set_prev_luminance(0.18)
    torch.cuda.synchronize()

    for _ in range(a.warmup):
        frame.render()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    # HIP events bracket the dominant kernel's launch inside the timed region (the ctx launches on torch's
    # current stream, so torch events sit on the right stream).  Two marker packets cost ~0.8 % of a frame, so only
    # every fifth frame carries them.
    shade_events = None if a.no_kernel_timing else []
    t0 = time.perf_counter()
    for i in range(a.steps):
        frame.render(shade_events if (shade_events is not None and i % 5 == 0) else None)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearsal else f"cuda:{dev}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / a.steps * 1e3
    total_px = spec.full_w * spec.full_h
    value = total_px / (ms_per_step * 1e-3) / 1e6

    out = {
        "metric": "shaded Mpixel/s at 4K G-buffer (full deferred frame: clustered shade + bloom + auto-exposure + ACES)",
        "value": round(value, 2), "unit": "Mpixel/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{a.width}x{a.height} G-buffer per GPU ({spec.full_w}x{spec.full_h} frame), {N_LIGHTS} clustered "
                               f"lights + IBL (env {ENV_SIZE}^2 x{ENV_MIPS} mips, LUT {LUT_RES}^2, SH9), auto-exposure + ACES + 9-tap bloom",
                   "tile": [spec.x0, spec.y0, spec.w, spec.h], "apron": spec.apron,
                   "allreduce": ("none" if world == 1 else ("gloo-rehearsal" if rehearsal else ("rccl-capi" if use_capi_rccl else "rccl-torch")))},
    }

    if rank == 0 and not a.no_kernel_timing:
        iters = max(5, min(a.steps, 20))
        ext_px = spec.ew * spec.eh
        # bloom and the luminance histogram run fused in the frame (pbr_bloom_histogram); the two
        # un-fused stage calls are timed as well for reference
        stages = {
            "cluster": (frame.clustered, None, 0.0),
            "shade": (frame.shade, None, BYTES_PER_PX["shade"] * ext_px),
            "bloom+histogram": (frame.bloom_histogram, None, BYTES_PER_PX["bloom"] * ext_px + BYTES_PER_PX["histogram"] * spec.w * spec.h),
            "bloom": (frame.bloom, None, BYTES_PER_PX["bloom"] * ext_px),
            "histogram": (frame.histogram, None, BYTES_PER_PX["histogram"] * spec.w * spec.h),
            "average": (frame.average, frame.histogram, 0.0),
            "tonemap": (frame.tonemap, None, BYTES_PER_PX["tonemap"] * spec.w * spec.h),
        }
        frame.hist.zero_()
        kern = {}
        for name, (fn, pre, nbytes) in stages.items():
            ms = time_stage(fn, iters, pre)
            kern[name] = {"ms": round(ms, 4), "GB/s": round(nbytes / (ms * 1e-3) / 1e9, 1) if nbytes else None}
            frame.hist.zero_()
        # dominant kernel: the shade.  Its launch duration is the mean over the frames of the timed region; the
        # per-stage figures below come from separate isolated launches after it
        shade_ms_in_frame = sum(e0.elapsed_time(e1) for e0, e1 in shade_events) / max(len(shade_events), 1)
        kern["shade(in frame)"] = {"ms": round(shade_ms_in_frame, 4), "GB/s": round(BYTES_PER_PX["shade"] * ext_px / (shade_ms_in_frame * 1e-3) / 1e9, 1)}
        dom = "shade"
        achieved = kern["shade(in frame)"]["GB/s"]
        out["roofline"] = {"bound": "hbm", "kernel": {"shade": "k_deferred_shade", "bloom+histogram": "bloom chain (8 launches, histogram fused)",
                                                     "tonemap": "k_tonemap"}[dom],
                           "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                           "traffic": pmc_traffic("k_deferred_shade", spec.ew * spec.eh) if dom == "shade" else None, "stage_ms": {k: v["ms"] for k, v in kern.items()},
                           "stage_GBps": {k: v["GB/s"] for k, v in kern.items()},
                           "note": "shade with 256 clustered lights is FP32-VALU-bound (SURVEY D6): its HBM fraction is structurally low; "
                                   "traffic (when present) is the committed rocprofv3 PMC figure for this workload, IBL gathers served by L2/MALL included"}
        nv = pmc_valu("k_deferred_shade", spec.ew * spec.eh)
        if nv:
            # supplementary compute roofline for the VALU-bound shade: measured issue cost of a plain fp32 VALU
            # instruction on gfx950 is ~4 cycles per SIMD (tools/valu_rate2.hip); 1024 SIMDs at the 2.4 GHz max clock
            peak = 1024 * 2.4e9 / 4.0
            out["roofline"]["valu"] = {"wave_insts_per_launch": nv, "achieved_Ginst_s": round(nv / (shade_ms_in_frame * 1e-3) / 1e9, 1),
                                       "peak_Ginst_s": round(peak / 1e9, 1), "frac": round(nv / (shade_ms_in_frame * 1e-3) / peak, 3),
                                       "source": "profiles/pmc_sq_latest.json (rocprofv3 SQ_INSTS_VALU) + tools/valu_rate2.hip"}
    if rank == 0 and not a.no_cpu_baseline and world == 1:
        try:
            out["cpu_baseline"] = cpu_baseline(g, lights, lut.cpu().view(torch.int16).numpy().view(np.float16),
                                               env.cpu().view(torch.int16).numpy().view(np.float16), a.width, a.height)
        except Exception as e:   # the baseline is reporting only; never fail the GPU measurement on it
            out["cpu_baseline"] = {"value": None, "unit": "Mpixel/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
