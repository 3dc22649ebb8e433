#!/usr/bin/env python3
"""bench.py — shaded Mpixel/s of the full deferred frame on synthetic 4K G-buffers (BASELINE.json).

A "step" is one frame of the hot path over one rank's tile: cluster build + cull, deferred shade
(256 clustered lights + IBL), the bloom chain (16 reference dispatches, 8 fused launches) with the luminance
histogram in its last kernel (+ RCCL all-reduce when N > 1), average, ACES tone-map — i.e. BASELINE.json configs[3]
at N = 1.  Inputs (G-buffer, lights, LUT, prefiltered env, SH) are resident in HBM before the timed region.

Multi-GPU (one process per GPU; `--gpus N` from a bare shell starts its own ranks through torch.distributed.run).  ONE launch
measures both multi-GPU workloads:
  * headline = WEAK scaling of the cfg4 scene: the same camera, lights and cluster grid rendered at a 16:9 frame of
    N x 8.3 Mpixel, cut into the most square tile grid (2x1, 2x2, 4x2 ...), one 4K-equivalent tile per GPU — per-pixel
    light lists are identical at every N because the scene is fixed in uv;
  * `cfg5` sub-record = BASELINE configs[4] (STRONG scaling): the 7680x4320 frame cut over the N ranks (N = 8: 2 rows x 4
    cols of 1920x2160), with the single-GPU time of the same 8K frame measured on rank 0 as the denominator.
  `--cfg5` (= `--frame 7680x4320`) makes the strong-scaling frame the headline instead.
Frames are timed IN ORDER at every N (the like-for-like figure `value` is computed from); for N > 1 the same frames with
their tail — histogram all-reduce, average, tone-map — on a side stream beside the next frame's shade are timed as well
and reported next to it (`ms_per_step_tail_overlapped`).
Bloom across tile borders: `--mode halo` (default) shades interior + 4 px and exchanges prefiltered half-res strips
(pbr_halo_exchange: RCCL send/recv over xGMI); `--mode apron` shades a 256-px apron instead (no data-path collective).
The only other collective is the 256-bin histogram all-reduce (on its own communicator).  `value` counts interior pixels.
`host_graph` = the same frames driven by the C++ pass graph (libpbr_host.so: RenderScheduler -> FrameGraph -> passes -> C ABI).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import hashlib
import json
import math
import os
import socket
import subprocess
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from direct12pbrrenderer_amd import scene, synth  # noqa: E402
from direct12pbrrenderer_amd.api import PbrContext  # noqa: E402
from direct12pbrrenderer_amd.pipeline import (DeferredFrame, HaloTransport, grid_for_world, parse_layout,  # noqa: E402
                                              tile_for_rank, tile_of_frame)
from direct12pbrrenderer_amd.structs import CLUSTER_DTYPE, CLUSTER_X, CLUSTER_Y, CLUSTER_Z, ENV_MIPS, Tile  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec; the measured streaming-read rate is reported beside it
FP32_VALU_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md chip table (packed FMA on every lane every cycle)
N_SIMD, CLOCK_HZ = 1024, 2.4e9
# tools/valu_rate3.hip on gfx950 at 5 waves per SIMD (profiles/r02_valu_rate3.txt): cycles per wave-instruction per SIMD
MEASURED_CYCLES = {"plain": 3.0, "fma": 2.6, "packed": 4.7, "trans": 8.5}
MEASURED_CLOCK_HZ = 2.1e9
# algorithmic bytes per pixel: unique bytes a reference pass must read + write once
BYTES_PER_PX = {"shade": 25.0, "bloom": 75.75, "histogram": 8.0, "tonemap": 12.0}
ENV_SIZE, LUT_RES, N_LIGHTS = 512, 512, 256
ROUGH_MIN = 48                  # synth.gbuffer_tile default; see config.workload
CFG5_FRAME = (7680, 4320)       # BASELINE.json configs[4]
CFG5_GRID = {8: (4, 2)}         # (cols, rows): "tiled 2x4" = 2 rows of 4 tiles; other N: the most square grid


DRY = False   # --dry-run: no device call is made anywhere below (bench_dryrun.py)


def dev_sync():
    if not DRY:
        torch.cuda.synchronize()


def dev_empty_cache():
    if not DRY:
        torch.cuda.empty_cache()


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=50)
    p.add_argument("--warmup", type=int, default=5)
    p.add_argument("--width", type=int, default=3840, help="single-GPU frame / per-GPU tile budget (weak scaling)")
    p.add_argument("--height", type=int, default=2160)
    p.add_argument("--frame", default=None, help="WxH: strong scaling of this frame over the ranks becomes the headline")
    p.add_argument("--cfg5", action="store_true", help="= --frame 7680x4320 (BASELINE configs[4]; at N = 8: --layout 2x4)")
    p.add_argument("--no-cfg5", action="store_true", help="N > 1, weak-scaling headline: skip the cfg5 (8K strong-scaling) sub-record")
    p.add_argument("--layout", default=None, help="RxC tile grid, rows x cols (cfg5: 2x4); default: most square")
    p.add_argument("--mode", choices=["halo", "apron"], default="halo", help="bloom across tile borders (N > 1)")
    p.add_argument("--transport", choices=["capi", "torch"], default="capi",
                   help="collectives through the C ABI's own RCCL communicators or through torch.distributed (also RCCL)")
    p.add_argument("--overlap", action="store_true",
                   help="halo mode: shade the tile's border ring first (side stream) and exchange its strips while the core is shaded. "
                        "Off by default: on one GPU the split costs 73 us (cfg5 tile) / 120 us (4K-equivalent tile) because the ring's "
                        "264-px-wide bands shade at half the whole tile's rate — it pays only where the exchange takes longer than that")
    p.add_argument("--settle", type=int, default=300,
                   help="untimed frames rendered before the W warm-up steps so that the device's clock has ramped (DVFS: a fresh process "
                        "runs its first ~100 ms at a lower clock; with the default K the timed region is only ~25 ms).  Reported in config")
    p.add_argument("--no-tail-overlap", action="store_true", help="N > 1: do not time the overlapped-tail variant")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-kernel-timing", action="store_true")
    p.add_argument("--no-host-graph", action="store_true", help="skip the C++ pass-graph leg (host_graph block)")
    p.add_argument("--no-shade-paths", action="store_true", help="skip the shade_ms_by_path A/B (N = 1)")
    p.add_argument("--no-configs", action="store_true", help="skip the `configs` block: BASELINE configs[0..2] and the reference's own operating point (N = 1)")
    p.add_argument("--deadline", type=float, default=240.0, help="seconds a collective phase may take before the rank gives up (N > 1)")
    p.add_argument("--dry-run", action="store_true",
                   help="N > 1 on a box without N GPUs (or without any): the real rank processes, rendezvous (gloo), tile / halo plans, candidate "
                        "fall-back, verification frames, timed loops, cfg5 record and JSON assembly with every device call replaced by a host "
                        "stand-in (bench_dryrun.py) — the orchestration of the 8-GPU launch, rehearsed where 8 processes are allowed.  Not a measurement")
    p.add_argument("--dry-fail", default=None,
                   help="dry run only, fault injection: '<rank>:<mode>/<transport>' — that rank's set-up of that candidate raises; '<rank>:miscount' — its first "
                        "verification frame counts one pixel too many (every rank must fall back together); '<rank>:hang' — it never arrives at its first "
                        "all-reduce (the watchdog's --deadline must end the launch)")
    return p.parse_args()


def spawn_ranks(a):
    """`python bench.py --gpus N` from a bare shell: start N fresh rank processes BEFORE anything here touches the GPU
    (never re-exec a process that has), relay their output, return their exit code.  With fewer than N devices on the
    box the ranks share cuda:0 in rehearsal mode (gloo + host copies): a functional run, not a measurement."""
    n_dev = torch.cuda.device_count()   # does not initialise the GPU (and, measured, does not count as a holder of the device either)
    env = dict(os.environ)
    if a.dry_run:
        pass                            # no rank touches a device
    elif n_dev < a.gpus:
        if n_dev < 1 or a.gpus > 5:
            # (the GPU pool this was developed on kills a job with more than 6 processes holding one card open — its "process guard" — and
            #  torch.distributed.run's agent is one of them (measured: 6 ranks = 7 holders), so the N = 8 launch itself cannot
            #  be rehearsed there: its layout arithmetic runs on CPU/gloo in tests/test_multigpu_gloo.py (2x4) and tests/test_bench_cpu.py,
            #  its orchestration — grid, cfg5 record, host-graph leg — as a 2x2 rehearsal in tests/test_gpu_bench.py)
            print(f"bench.py: --gpus {a.gpus} but {n_dev} device(s) visible (a rehearsal on one GPU takes at most 5 ranks)", file=sys.stderr)
            return 2
        env["PBR_BENCH_REHEARSAL"] = "1"
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


class Watchdog:
    """A rank blocked in a collective whose peers never arrive (RCCL kernel spinning, ncclCommInitRank waiting) cannot be
    unblocked from inside: each collective phase runs under a deadline, and a rank that misses it exits the process —
    after rank 0 has written whatever record is complete.  Every rank arms the same phases, so all of them leave."""

    def __init__(self):
        self.lock = threading.Lock()
        self.deadline, self.what, self.on_fire = None, "", None
        t = threading.Thread(target=self._run, daemon=True)
        t.start()

    def _run(self):
        while True:
            time.sleep(0.5)
            with self.lock:
                d, what, fire = self.deadline, self.what, self.on_fire
            if d is not None and time.monotonic() > d:
                print(f"bench.py: '{what}' did not finish before its deadline — giving up on this rank", file=sys.stderr, flush=True)
                code = 3
                try:
                    if fire:
                        code = fire(what)
                finally:
                    os._exit(code)

    def phase(self, seconds, what):
        """Context manager arming a deadline.  Phases may nest: leaving an inner one re-arms the enclosing one's deadline."""
        wd = self

        class _P:
            def __enter__(self_inner):
                with wd.lock:
                    self_inner.prev = (wd.deadline, wd.what)
                    wd.deadline, wd.what = time.monotonic() + seconds, what

            def __exit__(self_inner, *exc):
                with wd.lock:
                    wd.deadline, wd.what = self_inner.prev
                return False
        return _P()


def build_ibl(ctx, want_sky=False):
    """One-shot IBL precompute with the HIP kernels (not timed): LUT 512^2, sky 512^2 -> env 5 mips, SH9.
    want_sky: also return the host copy of the sky cube (the C++ pass graph builds its own IBL from it)."""
    sky_mips = int(np.log2(ENV_SIZE)) + 1
    sky_np = synth.env_cube(ENV_SIZE, sky_mips)
    sky = ctx.upload(sky_np)
    ctx.cube_gen_mips(sky, ENV_SIZE, sky_mips)
    lut = ctx.brdf_lut(LUT_RES)
    env = ctx.prefilter_env(sky, ENV_SIZE, sky_mips, ENV_SIZE, ENV_MIPS)
    sh = ctx.sh9_project(sky, ENV_SIZE, sky_mips)
    ctx.sync()
    return (lut, env, sh.cpu().numpy(), sky_np) if want_sky else (lut, env, sh.cpu().numpy())


def src_stamp():
    """sha1 of the sources the committed counter profiles describe; a profile with another stamp is stale."""
    h = hashlib.sha1()
    for f in ("shade.hip", "pbr_device.hpp"):
        h.update(open(os.path.join(ROOT, "direct12pbrrenderer_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:12]


def load_profile(name, workload_px):
    """A committed rocprofv3 --pmc summary (tools/summarize_pmc.py), or None when it describes another workload or
    another version of the shade kernel."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", name)))
    except Exception:
        return None
    if d.get("_workload_pixels") != workload_px or d.get("_src_stamp") != src_stamp():
        return None
    return d


def kernel_entry(d, prefix, need):
    for k, v in (d or {}).items():
        if isinstance(v, dict) and prefix in k and all(n in v for n in need):
            return v
    return None


def pmc_traffic(kernel_prefix, workload_px):
    """HBM-side bytes per launch from separate FETCH_SIZE and WRITE_SIZE passes.  Unit/correction per
    MI355X_MICROARCH.md: both counters are in KiB and gfx950's FETCH_SIZE tallies 128-B requests at 64 B, so reads
    are doubled.  None if no matching profile."""
    v = kernel_entry(load_profile("pmc_traffic_latest.json", workload_px), kernel_prefix, ("FETCH_SIZE", "WRITE_SIZE"))
    return int((2.0 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024) if v else None


def time_stage(fn, iters, pre=None):
    """Average device time of `fn` in ms, HIP events on the stream the kernels run on
    (the ctx is bound to torch's current stream, so torch.cuda.Event sees them)."""
    if pre:
        pre()
    fn()
    dev_sync()
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


def measure_hbm_read(ctx):
    """Streaming-read bandwidth of this device (GB/s): pbr_membench_read over a 2 GiB buffer (8 x the Infinity Cache),
    best of 5 launches, HIP events.  Not part of the timed step."""
    n = 2 << 30
    buf = torch.empty(n // 4, dtype=torch.int32, device=ctx.torch_device)
    buf.fill_(0x01020304)
    blocks = 256 * 16
    sink = torch.zeros(blocks, dtype=torch.int32, device=ctx.torch_device)
    best = 0.0
    for i in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ctx.membench_read(buf, sink, blocks)
        e1.record()
        e1.synchronize()
        if i >= 2:
            best = max(best, n / (e0.elapsed_time(e1) * 1e-3) / 1e9)
    del buf
    return best


def measure_valu(ctx, waves_per_simd=5, iters=60000, launches=12):
    """Issue rate of the VALU and the shader clock it SUSTAINS on this box, per instruction class (pbr_valubench: the loop of
    tools/valu_rate3.hip): every SIMD of the chip holds `waves_per_simd` waves (the shade's occupancy), each issuing iters x 8
    independent instructions between two reads of the shader-cycle counter and of the 100 MHz counter.  A class is loaded for `launches`
    back-to-back launches of 3-10 ms each (a short burst reads the clock on its way up from idle: ~1.75 GHz in the first launch against
    ~1.98 sustained under packed fp32, profiles/r05_af_box_probe_*.txt); reported: the median of the last four.  Per class: Ginst_s =
    wave-instructions of a launch / its HIP-event duration, clock_GHz = median over the waves of shader cycles / real time inside the
    loop, cycles_per_inst_per_simd = clock x SIMDs / rate.  Not part of the timed step (~0.4 s)."""
    cus = torch.cuda.get_device_properties(ctx.torch_device).multi_processor_count
    blocks = cus * waves_per_simd
    stamps = torch.zeros((blocks * 4, 4), dtype=torch.int64, device=ctx.torch_device)
    out = {"waves_per_simd": waves_per_simd, "compute_units": cus, "sustained_ms_per_class": None}
    spent = []
    for name, op in (("plain_v_mul_f32", 0), ("v_fma_f32", 1), ("packed_v_pk_fma_f32", 2), ("trans_v_rcp_f32", 3)):
        runs = []
        for _ in range(launches):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ctx.valubench(op, blocks, iters, stamps)
            e1.record()
            e1.synchronize()
            st = stamps.cpu().numpy()
            cyc, ticks = (st[:, 1] - st[:, 0]).astype(np.float64), (st[:, 3] - st[:, 2]).astype(np.float64)
            runs.append((e0.elapsed_time(e1), float(np.median(cyc / np.maximum(ticks, 1.0)) * 100e6), float(np.median(cyc)) / (iters * 8)))
        spent.append(round(sum(r[0] for r in runs), 1))
        last = runs[-4:]
        ms, clock, per_wave = (float(np.median([r[k] for r in last])) for k in range(3))
        rate = blocks * 4 * iters * 8 / (ms * 1e-3)
        out[name] = {"Ginst_s": round(rate / 1e9, 1), "clock_GHz": round(clock / 1e9, 3), "cycles_per_inst_per_simd": round(clock * cus * 4 / rate, 2),
                     "per_wave_cycles_per_inst": round(per_wave, 2), "clock_GHz_first_launch": round(runs[0][1] / 1e9, 3)}
    out["sustained_ms_per_class"] = spent
    return out


def mean_lights_per_pixel(g, gb_np, spec, clusters_dev):
    """Mean length of the light list a pixel of this rank's interior walks (ClusterIndex of clustered.hlsli:45-60 on
    the synthetic depth + the culled cluster table): the per-pixel work statistic that must not drift with N."""
    cl = clusters_dev.cpu().numpy().view(CLUSTER_DTYPE)
    depth = gb_np["depth"][spec.siy:spec.siy + spec.h, spec.six:spec.six + spec.w].astype(np.float64)
    near, far = float(g.Near), float(g.Far)
    z_vs = near * far / (far - depth * (far - near))
    sz = np.clip((CLUSTER_Z * np.log(np.clip(z_vs, near, far) / near) / np.log(far / near)).astype(np.int64), 0, CLUSTER_Z - 1)
    xs = (np.arange(spec.x0, spec.x0 + spec.w) + 0.5) / spec.full_w
    ys = (np.arange(spec.y0, spec.y0 + spec.h) + 0.5) / spec.full_h
    sx = np.clip(np.floor(xs * CLUSTER_X).astype(np.int64), 0, CLUSTER_X - 1)[None, :]
    sy = np.clip(np.floor((1.0 - ys) * CLUSTER_Y).astype(np.int64), 0, CLUSTER_Y - 1)[:, None]
    idx = sz + sx * CLUSTER_Z + sy * CLUSTER_X * CLUSTER_Z
    return float(np.minimum(cl["NumLights"], 32)[idx].mean())


def cpu_baseline(g, lights, lut_np, env_np, width, height, rows=1024, reps=5, parity_probe=None):
    """The oracle ("port") on a FIXED band of the single-GPU frame, all host threads: the `rows` centre rows of the frame (the same
    sample on every box and in every round, whatever the host's speed — round 3 sized the band with a 64-row probe and the figure
    wandered 4.3 -> 6.9 -> 5.1); inputs are synthesised ONCE outside the clock; the timed region is the oracle calls only (cluster
    build + cull, shade, bloom, histogram, average, tone-map); value = median of `reps` repetitions after one warm-up.
    parity_probe(orc): while the checker is loaded, the caller's GPU-vs-truth comparison (the `parity` block) runs here — this
    leg is the only place bench.py touches oracle/."""
    from oracle import binding as orc

    rows = max(16, min(height, rows) // 16 * 16)
    y0 = (height - rows) // 2 // 8 * 8
    gb = synth.gbuffer_tile(0, y0, width, rows, width, height)

    def run():
        t0 = time.perf_counter()
        cl = orc.cluster_build(g)
        orc.cluster_cull(g, lights, cl)
        hdr, _ = orc.deferred_shade(g, Tile(0, y0, width, rows, width, height), gb, lut_np, env_np, ENV_SIZE, ENV_MIPS, cl, lights)
        orc.bloom(hdr)
        hist = orc.lum_histogram(hdr)
        avg = orc.lum_average(hist, width * rows, 1.0 / 60.0, 0.18)
        orc.tonemap(hdr, avg)
        return time.perf_counter() - t0

    run()
    times = sorted(run() for _ in range(reps))
    dt = times[len(times) // 2]
    base = {"value": round(width * rows / dt / 1e6, 4), "unit": "Mpixel/s", "cores": orc.num_threads(), "kind": "port",
            "sample": f"{width}x{rows} band (rows {y0}..{y0 + rows - 1}, fixed) of the {width}x{height} frame: cluster build+cull, shade(256 lights+IBL), "
                      f"bloom, histogram, average, tonemap — oracle/pbr_oracle.cpp, OpenMP on {orc.num_threads()} threads, median of {reps} x {dt:.2f} s "
                      f"(min {times[0]:.2f}, max {times[-1]:.2f}; inputs synthesised outside the clock)"}
    parity = None
    if parity_probe is not None:
        try:
            parity = parity_probe(orc)
        except Exception as e:   # noqa: BLE001 — reporting only
            parity = {"error": f"{type(e).__name__}: {e}"}
    return base, parity


def shade_parity_probe(ctx, g, lights, lut_dev, env_dev, lut_np, env_np, width, height, rows=32):
    """The `parity` block: the GPU shade BEFORE its fp16 store (pbr_deferred_shade_f32, same kernel body) on a width x rows band of the
    bench frame against the DOUBLE-precision evaluation of the reference's formulas (oracle/pbr_oracle_f64.cpp), as plain relative
    L-inf — north_star's "<= 1e-4" with nothing allowed on top — next to the same figure for the fp32 CPU restatement of the shader
    (the error both share is the conditioning of GGX at highlights).  Returns a closure for cpu_baseline."""
    def probe(orc):
        y0 = (height - rows) // 2 // 8 * 8
        tile = Tile(0, y0, width, rows, width, height)
        gb = synth.gbuffer_tile(0, y0, width, rows, width, height)
        cl = orc.cluster_build(g)
        orc.cluster_cull(g, lights, cl)
        _, o32 = orc.deferred_shade(g, tile, gb, lut_np, env_np, ENV_SIZE, ENV_MIPS, cl, lights, want_f32=True)
        lo, hi, fl = orc.deferred_shade_f64(g, tile, gb, lut_np, env_np, ENV_SIZE, ENV_MIPS, cl, lights)
        gbd = {k: ctx.upload(v) for k, v in gb.items()}
        out = ctx.zeros((rows, width, 4), torch.float32)
        envp = ctx.env_pad(env_dev, ENV_SIZE, ENV_MIPS)
        ctx.deferred_shade_f32(g, tile, gbd, width, lut_dev, LUT_RES, envp, ENV_SIZE, ENV_MIPS, ctx.upload(cl), ctx.upload(lights), len(lights), out, width)
        ctx.sync()
        got = out.cpu().numpy()
        ok = fl == 0
        scale = float(np.abs(hi[ok]).max())
        dg = orc.truth_distance(got, lo, hi)[ok].max(axis=-1) / scale
        do = orc.truth_distance(o32, lo, hi)[ok].max(axis=-1) / scale
        worst = int(np.argmax(dg))
        # the literal reading of north_star: the GPU against the fp32 restatement of the shader itself (every pixel of the band: no truth,
        # so no pixel is "not comparable"), relative to the restatement's own scale
        s32 = float(np.abs(o32[..., :3]).max())
        dr = np.abs(got[..., :3] - o32[..., :3]).max(axis=-1) / s32
        restatement = {"what": "max |gpu - fp32 CPU restatement| / max |restatement| over every pixel of the band", "pixels": int(dr.size),
                       "max": float(f"{dr.max():.3e}"), "pixels_above_1e-4": int((dr > 1e-4).sum()), "q99.99": float(f"{np.quantile(dr, 0.9999):.3e}")}
        return {"gpu_vs_restatement_f32": restatement, "what": f"fp32 shade (pbr_deferred_shade_f32) vs f64 truth on the {width}x{rows} band at row {y0} of the bench frame; relative to scale = max |truth|",
                "bound": "north_star: <= 1e-4 relative L-inf (plain, nothing allowed on top)", "pixels": int(ok.sum()),
                "pixels_not_comparable": int((~ok).sum()), "scale": round(scale, 3),
                "gpu_pixels_above_1e-4": int((dg > 1e-4).sum()), "gpu_worst": float(f"{dg.max():.3e}"), "gpu_q99.99": float(f"{np.quantile(dg, 0.9999):.3e}"),
                "cpu_fp32_restatement_pixels_above_1e-4": int((do > 1e-4).sum()), "cpu_fp32_restatement_worst": float(f"{do.max():.3e}"),
                "cpu_fp32_restatement_at_gpu_worst_pixel": float(f"{do[worst]:.3e}"),
                "criterion_failures": int((dg > 1e-4 + 4.0 * do).sum()),
                "note": "pixels above 1e-4 are GGX highlights at roughness ~0.2 where the fp32 rounding of N.H is amplified by 4/t; the fp32 restatement of "
                        "the shader misses the bound at the same pixels by the same amount (tests assert |gpu - truth| <= 1e-4 scale + 4 |restatement - truth|)"}
    return probe


def weak_tile(world, cols, rows, base_w, base_h):
    """Per-rank tile of the weak-scaling frame: a 16:9 frame of world x (base_w x base_h) pixels cut cols x rows,
    rounded to the multiple of 16 the bloom pyramid needs (N = 2, 8: 2720 x 3056 = 8.31 Mpixel vs 8.29 at N = 1, 4)."""
    if world == 1:
        return base_w, base_h
    s = math.sqrt(world)
    return int(round(base_w * s / cols / 16)) * 16, int(round(base_h * s / rows / 16)) * 16


class Job:
    """What every workload of one launch shares: the context, the process group, the IBL and the communicators."""

    def __init__(self, a, ctx, dist, rank, world, rehearsal, flag_dev, wd):
        self.a, self.ctx, self.dist, self.rank, self.world, self.rehearsal, self.flag_dev, self.wd = a, ctx, dist, rank, world, rehearsal, flag_dev, wd
        self.notes = []
        self.capi_comm = False

    def all_agree(self, ok):
        """True iff every rank says ok (fallback decisions must be taken by all ranks together)."""
        if not self.dist:
            return bool(ok)
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=self.flag_dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
        return bool(t.item())

    def max_over_ranks(self, v):
        if not self.dist:
            return float(v)
        t = torch.tensor([v], dtype=torch.float64, device=self.flag_dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def barrier(self):
        dev_sync()
        if self.dist:
            self.dist.barrier()
        dev_sync()

    def make_allreduce(self, kind):
        if self.world == 1:
            return None
        if kind == "capi":
            return self.ctx.allreduce_hist
        dist, ctx = self.dist, self.ctx
        if self.rehearsal:
            def f(h):
                ctx.sync()
                t = h.cpu()
                dist.all_reduce(t)
                h.copy_(t)
            return f
        return lambda h: dist.all_reduce(h)   # RCCL, int32 sum == uint32 sum bit for bit


def build_frame(job, make_spec, g, lights, mode, transport, overlap):
    """LOCAL half of a candidate configuration: this rank's DeferredFrame (tile layout, halo plan, staging, buffers) and
    its G-buffer.  Raises on anything inconsistent — before any rank has entered a collective for this candidate."""
    world, rank = job.world, job.rank
    halo = mode == "halo" and world > 1
    specs = [make_spec(r, halo) for r in range(world)]
    spec = specs[rank]
    ht = HaloTransport("host" if job.rehearsal else transport, job.dist) if halo else None
    fr = DeferredFrame(job.ctx, spec, g, lights, job.lut, LUT_RES, job.env, ENV_SIZE, ENV_MIPS, allreduce=job.make_allreduce(transport),
                       all_specs=specs, rank=rank, halo_transport=ht, overlap=overlap)
    if halo:   # what pbr_halo_exchange would refuse, found here instead of inside the group call
        pitch, rows = spec.ew // 2, spec.eh // 2
        for peer, snd, rcv in fr.halo_plan_local:
            for q in (snd, rcv):
                if q and (q[0] + q[2] > pitch or q[1] + q[3] > rows):
                    raise RuntimeError(f"halo rectangle {q} of peer {peer} leaves the {pitch}x{rows} level-1 plane")
            if not (0 <= peer < world) or peer == rank:
                raise RuntimeError(f"bad halo peer {peer}")
    if DRY:
        if job.a.dry_fail == f"{rank}:{mode}/{transport}":
            raise RuntimeError("injected set-up failure (--dry-fail)")
        gb_np = {k: np.zeros((spec.sh, spec.sw), dtype=d) for k, d in (("A", np.uint32), ("B", np.uint32), ("C", np.uint32), ("depth", np.float32), ("stencil", np.uint8))}
    else:
        gb_np = synth.gbuffer_tile(spec.sx0, spec.sy0, spec.sw, spec.sh, spec.full_w, spec.full_h)
    fr.upload_gbuffer(gb_np)
    fr.set_prev_luminance(0.18)
    return fr, gb_np


def verify_frame(job, fr):
    """COLLECTIVE half: one frame of the very sequence the timed frames run, then: the all-reduced histogram must count
    every pixel of the whole frame exactly once, and in halo mode every level-1 texel of E must have arrived, with the
    sender's checksum.  Every rank runs the same collectives whatever it finds; a finding is raised at the end."""
    spec, dist, rank, world = fr.spec, job.dist, job.rank, job.world
    halo = spec.halo
    if halo:
        fr.level1.fill_(777.0)
    fr.clustered()
    if fr.split is not None:
        fr.shade_and_bloom_overlapped()
    else:
        fr.shade()
        fr.bloom_histogram()
    if fr.allreduce is not None:
        fr.allreduce(fr.hist)
    job.ctx.sync()
    problem = None
    counted = int(fr.hist.cpu().to(torch.int64).sum())
    if counted != spec.full_w * spec.full_h:
        problem = f"histogram counts {counted} pixels, frame has {spec.full_w * spec.full_h}"
    if halo:
        if bool((fr.level1 == 777.0).any()):
            problem = problem or "halo exchange left level-1 texels of the extended tile unfilled"
        pw = spec.ew // 2
        l1 = fr.level1.view(spec.eh // 2, pw, 4).view(torch.int16).to(torch.int64)

        def csum(r):
            return int(l1[r[1]:r[1] + r[3], r[0]:r[0] + r[2]].sum())
        mine = {}
        for peer, snd, rcv in fr.halo_plan_local:
            if snd:
                mine[("s", rank, peer)] = csum(snd)
            if rcv:
                mine[("r", peer, rank)] = csum(rcv)
        every = [None] * world
        dist.all_gather_object(every, mine)
        sent = {k[1:]: v for d in every for k, v in d.items() if k[0] == "s"}
        for k, v in mine.items():
            if k[0] == "r" and sent.get(k[1:]) != v:
                problem = problem or f"halo strip {k[1]} -> {k[2]} arrived with another checksum"
    if problem:
        raise RuntimeError(problem)
    fr.hist.zero_()
    fr.set_prev_luminance(0.18)


def run_workload(job, name, scaling, make_spec, full_w, full_h, cols, rows, steps, warmup, settle, want_stage_timing):
    """Set up, verify and time one workload on all ranks.  Returns a dict (rank 0 reports it)."""
    a, ctx, dist, rank, world, rehearsal = job.a, job.ctx, job.dist, job.rank, job.world, job.rehearsal
    cam = scene.Camera.reference_default(full_w, full_h)
    g = scene.make_global(cam, full_w, full_h, sh_pack=job.sh, delta_time=1.0 / 60.0)
    lights = synth.lights_in_view_box(N_LIGHTS, cam)
    notes = []

    want_overlap = a.mode == "halo" and a.overlap
    if world == 1:
        candidates = [("single", "none", False)]
    elif rehearsal:
        candidates = [(a.mode, "torch", want_overlap)] + ([(a.mode, "torch", False)] if want_overlap else []) + \
                     ([("apron", "torch", False)] if a.mode == "halo" else [])
    else:
        first = "capi" if job.capi_comm else "torch"
        candidates = [(a.mode, first, want_overlap)]
        if want_overlap:
            candidates.append((a.mode, first, False))
        if first == "capi":
            candidates.append((a.mode, "torch", False))
        if a.mode == "halo":
            candidates.append(("apron", "torch", False))
    frame = None
    for mode, transport, overlap in candidates:
        label = f"{mode}/{transport}{'/overlap' if overlap else ''}"
        ok, err = True, None
        try:
            frame, gb_np = build_frame(job, make_spec, g, lights, mode, transport, overlap)
        except Exception as e:   # noqa: BLE001
            ok, err = False, f"{label} could not be set up on rank {rank}: {e}"
        if not job.all_agree(ok):    # nobody has entered a collective of this candidate yet
            notes.append(err or f"{label} could not be set up on another rank")
            frame = None
            continue
        with job.wd.phase(a.deadline, f"{name}: verification frame of {label}"):
            try:
                verify_frame(job, frame)
            except Exception as e:   # noqa: BLE001
                ok, err = False, f"{label} failed verification on rank {rank}: {e}"
            if job.all_agree(ok):
                break
        notes.append(err or f"{label} failed on another rank")
        frame = None
        dev_sync()
    if frame is None:
        raise SystemExit(f"bench.py: no configuration of '{name}' passed its verification frame: " + "; ".join(notes))
    spec = frame.spec
    lights_px = 0.0 if DRY else mean_lights_per_pixel(g, gb_np, spec, frame.clusters)
    dev_sync()

    def timed(n_settle, n_warm, n_steps, shade_events=None):
        with job.wd.phase(a.deadline, f"{name}: timed frames"):
            for _ in range(n_settle + n_warm):   # the same count on every rank: frames carry collectives
                frame.render()
            frame.finish()
            job.barrier()
            t0 = time.perf_counter()
            for i in range(n_steps):
                frame.render(shade_events if (shade_events is not None and i % 5 == 0) else None)
            frame.finish()
            dev_sync()
            if dist:
                dist.barrier()
            dev_sync()
            return job.max_over_ranks(time.perf_counter() - t0)

    # ---- the headline: frames in order (at N = 1 there is nothing else)
    # HIP events bracket the dominant kernel's launch inside the timed region (the ctx launches on torch's current stream, so
    # torch events sit on the right stream).  Two marker packets cost ~0.8 % of a frame, so only every fifth frame carries them.
    shade_events = [] if (want_stage_timing and not a.no_kernel_timing) else None
    dt = timed(settle, warmup, steps, shade_events)
    ms_per_step = dt / steps * 1e3

    # ---- the same frames with the tail beside the next frame's shade (multi-GPU, C-ABI collectives): hides the all-reduce's
    # latency.  Checked on every rank against the plain order (three frames each: same adapted luminance, same LDR image).
    ms_overlapped = None
    if world > 1 and not rehearsal and transport == "capi" and frame.split is None and not a.no_tail_overlap:
        try:
            with job.wd.phase(a.deadline, f"{name}: overlapped-tail check"):
                def three_frames():
                    frame.set_prev_luminance(0.18)
                    frame.hist.zero_()
                    for _ in range(3):
                        frame.render()
                    frame.finish()
                    dev_sync()
                    return float(frame.avg.cpu()[0]), int(frame.ldr.to(torch.int64).sum().item())
                plain = three_frames()
                ok = True
                try:
                    frame.enable_tail_overlap(capi_allreduce=True)   # allocations only: no rank enters a collective the others do not
                except Exception as e:   # noqa: BLE001
                    ok = False
                    notes.append(f"tail overlap not set up on rank {rank}: {e}")
                good = job.all_agree(ok) and job.all_agree(three_frames() == plain)
            if good:
                ms_overlapped = timed(20, warmup, steps) / steps * 1e3
            else:
                notes.append("overlapped frame tail not timed (not available, or it did not reproduce the plain order's frames)")
            frame._tail_overlap = False
            frame.set_prev_luminance(0.18)
            frame.hist.zero_()
            dev_sync()
        except Exception as e:   # noqa: BLE001 — the in-order figure above stands; this variant is an extra
            notes.append(f"overlapped frame tail failed on rank {rank}: {type(e).__name__}: {e}")
            ms_overlapped = None
            frame._tail_overlap = False

    # ---- N = 1: the same throughput mode (frame i's tail beside frame i + 1's shade; no collective in it).  Two depths: the tail
    # alone (average + tone-map) and everything behind the shade (bloom chain + average + tone-map).  Each is checked against the plain
    # order first (three frames: same adapted luminance, same LDR image).  Reported NEXT TO the in-order figures, never instead.
    overlap_n1 = None
    if world == 1 and want_stage_timing and not a.no_tail_overlap:
        overlap_n1 = {}
        try:
            def three_frames():
                frame.set_prev_luminance(0.18)
                frame.hist.zero_()
                for _ in range(3):
                    frame.render()
                frame.finish()
                ctx.sync()
                return float(frame.avg.cpu()[0]), int(frame.ldr.to(torch.int64).sum().item())
            plain = three_frames()
            for key, from_bloom in (("tail", False), ("post_shade", True)):
                frame.enable_tail_overlap(from_bloom=from_bloom)
                same = three_frames() == plain
                ms = timed(60, warmup, steps) / steps * 1e3 if same else None
                overlap_n1[key] = {"reproduces_in_order_frames": same, "ms_per_step": round(ms, 4) if ms else None,
                                   "value": round(spec.full_w * spec.full_h / (ms * 1e-3) / 1e6, 2) if ms else None}
                frame.finish()
                ctx.sync()
                frame._tail_overlap = False
            frame.set_prev_luminance(0.18)
            frame.hist.zero_()
            ctx.sync()
        except Exception as e:   # noqa: BLE001 — an extra; the in-order figure stands
            overlap_n1["error"] = f"{type(e).__name__}: {e}"
            frame._tail_overlap = False

    lp = [None] * world
    if dist:
        dist.all_gather_object(lp, round(lights_px, 3))
    else:
        lp = [round(lights_px, 3)]
    total_px = spec.full_w * spec.full_h
    res = {"name": name, "scaling": scaling, "frame_obj": frame, "g": g, "lights": lights, "gb_np": gb_np, "spec": spec, "cols": cols, "rows": rows,
           "ms_per_step": ms_per_step, "value": total_px / (ms_per_step * 1e-3) / 1e6, "ms_overlapped": ms_overlapped,
           "mode": mode, "transport": transport, "notes": notes, "lights_px": lp, "shade_events": shade_events, "total_px": total_px,
           "overlap_n1": overlap_n1}
    return res


def two_preset_lights(cam):
    """The bench lights with every second one re-cast as radius 1 / intensity 40: the same positions, colours and CULLING radius
    (radius * 1.814 * sqrt(intensity): 2 sqrt(10) = 1 sqrt(40)), hence the same cluster lists and trip counts — but a second attenuation preset
    ((1, 4.5, 75) for radius <= 1, Scene.cpp:132-165), which switches the shade's shared-polynomial path off."""
    a = synth.lights_in_view_box(N_LIGHTS, cam)
    b = synth.lights_in_view_box(N_LIGHTS, cam, radius=1.0, intensity=40.0)
    a[1::2] = b[1::2]
    return a


def shade_paths_ab(job, res, frames=30):
    """`shade_ms_by_path`: the in-frame shade time (HIP events around the launch inside whole rendered frames) and the frame time of the bench
    workload as shipped and with the inputs that take the shade's other light-walk instantiations (shade.hip: walk(QSAFE, TSAFE, ATT)):
    SURVEY 8d's full roughness range (a wave of 64 independent pixels is then never all >= 0.157, so the GGX floor max(pi t^2, 1e-6) is
    evaluated: TSAFE off) and two attenuation presets (ATT off).  Outside the timed region; N = 1 only.  The full-range frames hold fp16 inf /
    NaN after bloom (the reason for the floor on roughness) — their time is what is reported, not their image."""
    ctx, spec, g = job.ctx, res["spec"], res["g"]
    cam = scene.Camera.reference_default(spec.full_w, spec.full_h)
    out = {}
    variants = [("as_shipped", ROUGH_MIN, False), ("roughness_0_255", 0, False), ("two_attenuation_presets", ROUGH_MIN, True),
                ("roughness_0_255_and_two_presets", 0, True)]
    for name, rough_min, two in variants:
        lights = two_preset_lights(cam) if two else res["lights"]
        fr = DeferredFrame(ctx, spec, g, lights, job.lut, LUT_RES, job.env, ENV_SIZE, ENV_MIPS)
        gb_np = res["gb_np"] if rough_min == ROUGH_MIN else synth.gbuffer_tile(spec.sx0, spec.sy0, spec.sw, spec.sh, spec.full_w, spec.full_h, rough_min=rough_min)
        fr.upload_gbuffer(gb_np)
        fr.set_prev_luminance(0.18)
        for _ in range(60):
            fr.render()
        dev_sync()
        ev = []
        t0 = time.perf_counter()
        for i in range(frames):
            fr.render(ev if i % 3 == 0 else None)
        dev_sync()
        frame_ms = (time.perf_counter() - t0) / frames * 1e3
        shade_ms = sum(a.elapsed_time(b) for a, b in ev) / len(ev)
        out[name] = {"shade_ms": round(shade_ms, 4), "frame_ms": round(frame_ms, 4),
                     "mean_lights_per_pixel": round(mean_lights_per_pixel(g, gb_np, spec, fr.clusters), 3)}
        del fr, gb_np
        dev_empty_cache()
    base = out["as_shipped"]["shade_ms"]
    for v in out.values():
        v["shade_vs_as_shipped"] = round(v["shade_ms"] / base, 4)
    out["note"] = ("same frame size, lights' positions / colours / culling radii and cluster lists in all four (mean_lights_per_pixel); only the walk "
                   "instantiation differs: as_shipped = walk(QSAFE, TSAFE, ATT) in every wave; roughness_0_255 = TSAFE off (+1 packed mul, +2 v_max per "
                   "light pair); two presets = ATT off (per-light coefficients: 19 instead of 13 LDS dwords per pair).  The roughness_0_255 frames contain "
                   "fp16 inf (GGX peak > 65504 below roughness ~0.17) — timing only")
    return out


def time_batch(fn, n=30, reps=3):
    """ms per call of `fn` over back-to-back batches of n launches (best of reps), HIP events on the kernels' stream."""
    fn()
    dev_sync()
    best = float("inf")
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best


def cpu_median(fn, reps=3):
    """median wall time (s) of `fn` over `reps` runs after one warm-up (the oracle runs on every host thread: OpenMP)"""
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2]


def configs_block(job):
    """`configs`: the single-GPU BASELINE.json configs other than the headline — cfg1 (split-sum LUT), cfg2 (1080p, one point light),
    cfg3 (GGX prefilter + SH9) — and the reference's own operating point (1440x960, Asset/Scene/main.json's 8 lights, LUT 512^2,
    fence per frame, through the C++ pass graph), each with its figure of merit and, unless --no-cpu-baseline, the CPU leg of BASELINE.md
    section 2 (the oracle, kind "port", all host threads, on the whole config or a stated sample).  Outside the timed region; N = 1 only.
    (The same legs, with the cfg5 tiles, as a stand-alone tool: tools/bench_configs.py.)"""
    ctx, a = job.ctx, job.a
    cpu = not a.no_cpu_baseline
    orc = None
    if cpu:
        from oracle import binding as orc   # the checker, timed beside the product: never on the product's path
    out = {}

    # ---- cfg1: "256x256 split-sum BRDF LUT" (the reference's own LUT is 512x512 — both)
    c1 = {"workload": "split-sum BRDF LUT, 1 024 Hammersley / GGX samples per texel (precompute_brdf.hlsl); 256^2 = BASELINE configs[0], 512^2 = the reference's own size",
          "bound": "FP32 VALU issue (no input, 256 KiB / 1 MiB out)"}
    for res in (256, 512):
        buf = ctx.empty((res, res, 2), torch.float16)
        ms = time_batch(lambda: ctx.brdf_lut(res, out=buf), 10)
        r = {"ms": round(ms, 4), "Msamples_per_s": round(res * res * 1024 / ms / 1e3, 1)}
        if cpu:
            dt = cpu_median(lambda: orc.brdf_lut(res))
            r["cpu_baseline"] = {"value": round(res * res * 1024 / dt / 1e6, 1), "unit": "Msamples/s", "ms": round(dt * 1e3, 2), "cores": orc.num_threads(), "kind": "port",
                                 "sample": f"the whole {res}x{res} plane (orc_brdf_lut, OpenMP), median of 3"}
        c1[f"lut{res}"] = r
    out["cfg1_brdf_lut"] = c1

    # ---- cfg2: 1920x1080, 1 point light, deferred shade (headline of cfg2 = the shade alone)
    W2, H2 = 1920, 1080
    cam = scene.Camera.reference_default(W2, H2)
    g2 = scene.make_global(cam, W2, H2, sh_pack=job.sh, delta_time=1.0 / 60.0)
    light1 = synth.reference_scene_light()
    from direct12pbrrenderer_amd.pipeline import TileSpec
    fr = DeferredFrame(ctx, TileSpec(0, 0, W2, H2, W2, H2, 0), g2, light1, job.lut, LUT_RES, job.env, ENV_SIZE, ENV_MIPS)
    gb2 = synth.gbuffer_tile(0, 0, W2, H2, W2, H2)
    fr.upload_gbuffer(gb2)
    fr.set_prev_luminance(0.18)
    for _ in range(60):
        fr.render()
    shade_ms = time_batch(fr.shade)
    frame_ms = time_batch(fr.render, 20)
    px2 = W2 * H2
    algo = BYTES_PER_PX["shade"] * px2
    prof = load_profile("pmc_cfg2_latest.json", px2)
    v = kernel_entry(prof, "k_deferred_shade", ("FETCH_SIZE", "WRITE_SIZE"))
    traffic = int((2.0 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024) if v else None
    c2 = {"workload": "1920x1080 synthetic G-buffer (SURVEY 8d, roughness [48,255]), 1 point light (the reference scene's light_1) + IBL: the deferred shade alone",
          "shade_ms": round(shade_ms, 4), "value": round(px2 / shade_ms / 1e3, 1), "unit": "Mpixel/s", "frame_ms_all_passes": round(frame_ms, 4),
          "roofline": {"bound": "hbm", "achieved": round(algo / (shade_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "frac": round(algo / (shade_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "algorithmic_bytes": int(algo), "traffic": traffic,
                       "traffic_over_algorithmic": round(traffic / algo, 2) if traffic else None,
                       "traffic_GBps": round(traffic / (shade_ms * 1e-3) / 1e9, 1) if traffic else None,
                       "traffic_source": "profiles/pmc_cfg2_latest.json (tools/pmc_cfg.sh: separate FETCH_SIZE / WRITE_SIZE passes, gfx950 corrections), stamp-checked against shade.hip"
                                         if traffic else "no counter profile of this shade.hip committed",
                       "kernel_limited_by": "cache lines pulled by the IBL footprint gathers (texture addresser / L1 line handling, 65-80 % of the streaming rate in fabric traffic; not occupancy, not the walk): DESIGN.md section 4"}}
    if v and prof:
        for k in ("TA_BUSY_avr", "TCP_PENDING_STALL_CYCLES_sum", "TCC_HIT_sum", "TCC_MISS_sum", "GRBM_GUI_ACTIVE", "SQ_INSTS_VALU"):
            if k in v:
                c2["roofline"].setdefault("counters", {})[k] = v[k]
    if cpu:
        lut_np = job.lut.cpu().view(torch.int16).numpy().view(np.float16)
        env_np = job.env.cpu().view(torch.int16).numpy().view(np.float16)

        def shade_cpu():
            cl = orc.cluster_build(g2)
            orc.cluster_cull(g2, light1, cl)
            orc.deferred_shade(g2, Tile(0, 0, W2, H2, W2, H2), gb2, lut_np, env_np, ENV_SIZE, ENV_MIPS, cl, light1)
        dt = cpu_median(shade_cpu)
        c2["cpu_baseline"] = {"value": round(px2 / dt / 1e6, 2), "unit": "Mpixel/s", "ms": round(dt * 1e3, 1), "cores": orc.num_threads(), "kind": "port",
                              "sample": "the whole 1920x1080 frame: cluster build + cull + deferred shade (oracle, OpenMP), inputs synthesised outside the clock, median of 3"}
    out["cfg2_1080p_1_light"] = c2
    del fr
    dev_empty_cache()

    # ---- cfg3: 512^2 cube, GGX prefilter (5 mips, 1 024 spp) + SH9
    sky_mips = int(np.log2(ENV_SIZE)) + 1
    sky = ctx.upload(job.sky_np)
    ctx.cube_gen_mips(sky, ENV_SIZE, sky_mips)
    envbuf = ctx.prefilter_env(sky, ENV_SIZE, sky_mips, ENV_SIZE, ENV_MIPS)
    texels = 6 * sum((ENV_SIZE >> m) ** 2 for m in range(ENV_MIPS))
    ms_f = time_batch(lambda: ctx.prefilter_env(sky, ENV_SIZE, sky_mips, ENV_SIZE, ENV_MIPS, out=envbuf), 5)
    sky_h = sky.half().float()   # what the reference's BC6H_UF16 sky assets decode to: the kernel then samples its exact half copy
    ms_h = time_batch(lambda: ctx.prefilter_env(sky_h, ENV_SIZE, sky_mips, ENV_SIZE, ENV_MIPS, out=envbuf), 5)
    shbuf = ctx.empty((28,), torch.float32)
    ms_sh = time_batch(lambda: ctx.sh9_project(sky, ENV_SIZE, sky_mips, out=shbuf), 10)
    sh_bytes = 6 * ENV_SIZE * ENV_SIZE * 16
    c3 = {"workload": "512^2 fp32 RGBA cube (SURVEY 8d sky): GGX prefilter, 5 mips x 1 024 samples per texel (env_map_gen.hlsl), + SH9 projection (deterministic quadrature)",
          "prefilter_fp32_source": {"ms": round(ms_f, 3), "Gsamples_per_s_reference_equivalent": round(texels * 1024 / ms_f / 1e6, 2), "bound": "texture addresser (L2-resident gathers)"},
          "prefilter_half_representable_source": {"ms": round(ms_h, 3), "Gsamples_per_s_reference_equivalent": round(texels * 1024 / ms_h / 1e6, 2), "bound": "FP32 VALU issue",
                                                  "note": "every reference sky asset is BC6H_UF16: its texels are half values"},
          "reference_samples": texels * 1024,
          "sh9": {"ms": round(ms_sh, 4), "GBps": round(sh_bytes / ms_sh / 1e6, 1), "frac_of_8TBps": round(sh_bytes / ms_sh / 1e6 / HBM_PEAK_GBS, 4),
                  "bound": "launch latency (two dependent launches over 25 MB)"}}
    if cpu:
        sky_np = sky.cpu().numpy()
        rng = np.random.default_rng(7)
        n_s = 1024
        picks = {m: rng.integers(0, 6 * (ENV_SIZE >> m) ** 2, n_s).astype(np.uint32) for m in range(ENV_MIPS)}
        orc.prefilter_env_texels(sky_np, ENV_SIZE, sky_mips, ENV_SIZE, ENV_MIPS, 1, picks[1][:256])   # warm-up
        est = 0.0
        for m in range(ENV_MIPS):   # each mip's sample scaled to the mip's texel count (the shader loops 1 024 times per texel on every mip)
            t0 = time.perf_counter()
            orc.prefilter_env_texels(sky_np, ENV_SIZE, sky_mips, ENV_SIZE, ENV_MIPS, m, picks[m])
            est += (time.perf_counter() - t0) / n_s * 6 * (ENV_SIZE >> m) ** 2
        c3["prefilter_fp32_source"]["cpu_baseline"] = {"value": round(texels * 1024 / est / 1e9, 3), "unit": "Gsamples/s (reference-equivalent)",
                                                       "ms_estimated_whole_config": round(est * 1e3, 1), "cores": orc.num_threads(), "kind": "port",
                                                       "sample": f"{n_s} random texels of each of the 5 mips (orc_prefilter_env_texels, OpenMP), each mip's time scaled to its texel count"}
        dt = cpu_median(lambda: orc.sh9_project(sky_np, ENV_SIZE))
        c3["sh9"]["cpu_baseline"] = {"value": round(sh_bytes / dt / 1e9, 2), "unit": "GB/s", "ms": round(dt * 1e3, 2), "cores": orc.num_threads(), "kind": "port",
                                     "sample": "the whole 512^2 cube (orc_sh9_project quadrature, OpenMP), median of 3"}
    out["cfg3_prefilter_sh9"] = c3
    del sky, sky_h, envbuf
    dev_empty_cache()

    # ---- the reference's own operating point (App.h:77-78, Asset/Scene/main.json, D3D12Device.cpp:993-1003) through libpbr_host.so
    try:
        import tempfile
        from direct12pbrrenderer_amd.host import HostRenderer
        WR, HR = 1440, 960
        recs = np.load(os.path.join(ROOT, "tests", "golden", "scene_lights.npz"))   # the 8 light records of main.json (data; tests/golden/make_scene_lights.py)
        gbr = synth.gbuffer_tile(0, 0, WR, HR, WR, HR)
        rp = {"workload": "1440x960 (the reference's default target, App.h:77-78), reference camera, the 8 point lights of Asset/Scene/main.json through the scene-file "
                          "reader + CPU light cull, LUT 512^2, env 512^2 x 5, synthetic G-buffer; every pass of the frame graph, fence wait per frame (D3D12Device.cpp:993-1003)"}
        with tempfile.TemporaryDirectory() as td:
            path = os.path.join(td, "main.json")
            with open(path, "w") as f:
                f.write(scene.scene_file_text(recs))
            for key, fused in (("dispatch_by_dispatch", False), ("fused_passes", True)):
                r = HostRenderer(ctx.device if hasattr(ctx, "device") else 0, WR, HR, ENV_SIZE, LUT_RES)
                try:
                    r.set_fused(fused)
                    r.set_skybox(job.sky_np, ENV_SIZE)
                    r.load_scene_lights(path)
                    r.set_gbuffer(gbr)
                    r.set_initial_luminance(0.18)
                    r.render()                      # the one-shot passes (env prefilter, LUT) run here
                    r.render_n(100)                 # clocks up
                    n0 = r.dispatch_count()
                    ms = min(r.render_n(100) for _ in range(3))
                    rp[key] = {"ms_per_frame": round(ms, 4), "Mpixel_per_s": round(WR * HR / ms / 1e3, 1), "dispatches_per_frame": n0}
                finally:
                    r.close()
        out["reference_operating_point"] = rp
    except Exception as e:   # noqa: BLE001 — reporting only
        out["reference_operating_point"] = {"error": f"{type(e).__name__}: {e}"}
    return out


def workload_config(job, r, settle):
    spec, world, rows, cols = r["spec"], job.world, r["rows"], r["cols"]
    if world == 1:
        workload = f"{spec.full_w}x{spec.full_h} G-buffer"
    elif r["scaling"] == "weak":
        workload = (f"weak scaling of the cfg4 scene: {spec.full_w}x{spec.full_h} frame (16:9, {world} x 8.3 Mpixel) as {rows} rows x {cols} cols of "
                    f"{spec.w}x{spec.h} tiles, one per GPU")
    else:
        workload = f"strong scaling: {spec.full_w}x{spec.full_h} frame as {rows} rows x {cols} cols of {spec.w}x{spec.h} tiles, one per GPU"
    workload += (f"; {N_LIGHTS} clustered lights + IBL (env {ENV_SIZE}^2 x{ENV_MIPS} mips, LUT {LUT_RES}^2, SH9), auto-exposure + ACES + 9-tap bloom"
                 f"; synthetic G-buffer per SURVEY 8d EXCEPT roughness u8 uniform on [{ROUGH_MIN},255] (8d: [0,255]; below ~0.17 the GGX peak of an "
                 f"intensity-10 light overflows fp16 and one inf turns the bloom pyramid into NaNs), every pixel an independent surface sample, stencil 1 "
                 f"everywhere; all {N_LIGHTS} lights radius 2 / intensity 10 => ONE attenuation preset (1, 0.7, 1.8) — with these two properties every wave of "
                 f"the shade takes its all-fast-paths walk (GGX floor off, shared attenuation polynomial): see shade_ms_by_path for the other paths")
    fr = r["frame_obj"]
    cfg = {"workload": workload, "frame": [spec.full_w, spec.full_h], "layout_rows_x_cols": f"{rows}x{cols}",
           "tile": [spec.x0, spec.y0, spec.w, spec.h], "shaded_rect": [spec.sx0, spec.sy0, spec.sw, spec.sh],
           "bloom_rect": [spec.ex0, spec.ey0, spec.ew, spec.eh],
           "bloom_borders": "none" if world == 1 else (r["mode"] + (" (ring first: exchange overlaps the core's shade)" if fr.split is not None else "")),
           "collectives": "none" if world == 1 else ("gloo-rehearsal" if job.rehearsal else ("dry run: gloo stand-in for " if DRY else "") + f"rccl-{r['transport']}"
                                                     + (" (halo exchange and histogram all-reduce on separate communicators)" if r["transport"] == "capi" else "")),
           "mean_lights_per_pixel_by_rank": r["lights_px"], "clock_settle_frames": settle,
           "frame_tail": "in order (value, ms_per_step)"}
    if r["ms_overlapped"] is not None:
        cfg["ms_per_step_tail_overlapped"] = round(r["ms_overlapped"], 4)
        cfg["value_tail_overlapped"] = round(r["total_px"] / (r["ms_overlapped"] * 1e-3) / 1e6, 2)
        cfg["frame_tail"] += "; *_tail_overlapped = the same frames with all-reduce + average + tone-map on a side stream, beside the next frame's shade"
    o1 = r.get("overlap_n1")
    if o1:
        cfg["throughput_mode"] = dict(o1, note="N = 1, frames NOT in order: frame i's tail on the context's high-priority side stream beside frame i + 1's cluster "
                                               "pass + shade (HDR target and histogram double-buffered, per-frame results identical to the in-order frames): "
                                               "tail = average + tone-map; post_shade = bloom chain + average + tone-map.  `value` / `ms_per_step` of this record "
                                               "remain the IN-ORDER figures")
        best = min((v for v in o1.values() if isinstance(v, dict) and v.get("ms_per_step")), key=lambda v: v["ms_per_step"], default=None)
        if best:
            cfg["ms_per_step_tail_overlapped"] = best["ms_per_step"]
            cfg["value_tail_overlapped"] = best["value"]
            cfg["frame_tail"] += "; *_tail_overlapped = the faster of throughput_mode's two variants (see there)"
    if r["notes"] or job.notes:
        cfg["notes"] = job.notes + r["notes"]
    return cfg


def host_graph_leg(job, r, frames):
    """The same workload driven by the C++ pass graph (libpbr_host.so: RenderScheduler -> FrameGraph -> the reference-shaped
    passes -> HipCommandList -> C ABI): ms per frame with every reference dispatch issued one by one and with fused passes, each
    with the reference's per-frame fence wait and in throughput mode (3 frames in flight).  Outside the timed region.  Multi-GPU:
    pbrh_create_tile in the workload's mode, halo exchange + histogram all-reduce over the renderer's own RCCL communicators."""
    from direct12pbrrenderer_amd.host import HostRenderer
    from direct12pbrrenderer_amd.api import comm_unique_id
    a, world, rank, dist = job.a, job.world, job.rank, job.dist
    spec = r["spec"]
    halo = r["mode"] == "halo"
    ok, err, hr = True, None, None
    try:
        tile = (spec.full_w, spec.full_h, r["cols"], r["rows"], rank, halo) if world > 1 else None
        hr = HostRenderer(job.ctx.device, spec.full_w, spec.full_h, ENV_SIZE, LUT_RES, tile=tile)
        hr.set_skybox(job.sky_np, ENV_SIZE)
        hr.set_lights(r["lights"])
        if world > 1 and not halo:
            raise RuntimeError("apron mode is not driven through the host graph here")
        hr.set_gbuffer(r["gb_np"])
        hr.set_initial_luminance(0.18)
    except Exception as e:   # noqa: BLE001
        ok, err = False, f"host graph not set up on rank {rank}: {e}"
    if not job.all_agree(ok):
        if hr:
            hr.close()
        return {"error": err or "host graph not set up on another rank"}
    out = {"path": "libpbr_host.so: RenderScheduler::ExecutePipeline -> FrameGraph::Execute -> pass classes -> HipCommandList -> C ABI (include/pbr_hip.h)",
           "frames_per_figure": frames}
    try:
        with job.wd.phase(a.deadline, "host graph leg"):
            if world > 1 and job.rehearsal:
                # all ranks on one GPU: RCCL refuses several ranks of one communicator on one device, so the renderer runs its halo pass
                # with the loopback transport (pack -> nothing moves -> unpack: the strips a rank receives stay empty) and gets the other
                # tiles' histogram counts through the host (gloo) once — every call the pass graph makes on a tile is made, the frames are
                # not the real ones at the tile borders.  A functional run of this leg, like the rest of a rehearsal
                hr.set_halo_loopback(True)
                hr.capture_histogram(True)
                hr.render()
                mine = torch.from_numpy(hr.captured_histogram().astype(np.int64))
                total = mine.clone()
                dist.all_reduce(total)
                hr.capture_histogram(False)
                hr.set_external_histogram((total - mine).numpy().astype(np.uint32))
                out["rehearsal"] = "loopback halo transport + histogram counts through gloo: functional run, not a measurement"
            elif world > 1:
                ids = [comm_unique_id() if rank == 0 else None]
                dist.broadcast_object_list(ids, src=0)
                hr.comm_init(world, rank, ids[0])
            hr.render()                       # one-shot IBL passes + first frame
            for fused in (0, 1):
                hr.set_fused(fused)
                for in_flight in (1, 3):
                    hr.set_frames_in_flight(in_flight)
                    hr.render_n(max(frames // 2, 5))
                    if dist:
                        dist.barrier()
                    ms = job.max_over_ranks(hr.render_n(frames))
                    key = ("fused" if fused else "dispatch_by_dispatch") + ("_fence_per_frame" if in_flight == 1 else "_throughput")
                    out[key + "_ms"] = round(ms, 4)
                out["dispatches_per_frame_" + ("fused" if fused else "dispatch_by_dispatch")] = hr.dispatch_count()
            # and with the frame's tail (all-reduce, average, tone-map) on the side stream beside the next frame's shade
            hr.set_tail_overlap(True)
            hr.render_n(max(frames // 2, 5))
            if dist:
                dist.barrier()
            out["fused_throughput_tail_overlapped_ms"] = round(job.max_over_ranks(hr.render_n(frames)), 4)
            hr.set_tail_overlap(False)
            if world == 1:   # ... and with everything behind the shade there (frames without a halo exchange)
                hr.set_tail_overlap(2)
                hr.render_n(max(frames // 2, 5))
                out["fused_throughput_post_shade_overlapped_ms"] = round(hr.render_n(frames), 4)
                hr.set_tail_overlap(0)
        best = out["fused_throughput_ms"]
        out["Mpixel_s_fused_throughput"] = round(r["total_px"] / (best * 1e-3) / 1e6, 1)
        out["vs_python_driven_frame"] = round(best / r["ms_per_step"], 4)
        out["note"] = ("fence_per_frame = D3D12Device::EndFrame's wait after every frame (the reference's loop); throughput = 3 frames in flight; "
                       "the host graph resolves the sky on stencil == 0 pixels every frame like the reference (SkyboxPass), the Python-driven frame does not")
    except Exception as e:   # noqa: BLE001
        out["error"] = f"rank {rank}: {e}"
    finally:
        hr.close()
    return out


def main():
    a = parse()
    if a.cfg5 and not a.frame:
        a.frame = f"{CFG5_FRAME[0]}x{CFG5_FRAME[1]}"
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(a))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # stdout carries ONE line, the JSON record of rank 0: whatever libraries print meanwhile (gloo's "[Gloo] Rank 0 is connected
    # to ..." goes to the C stdout) is sent to stderr by pointing fd 1 there; the record is written to the saved descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    # PBR_BENCH_REHEARSAL=1: every rank on cuda:0 with gloo collectives + host copies — exercises this file's
    # multi-rank path on a one-GPU box; not a measurement
    rehearsal = os.environ.get("PBR_BENCH_REHEARSAL", "0") == "1"
    global DRY
    DRY = bool(a.dry_run)
    if DRY:
        if world < 2:
            print("bench.py: --dry-run rehearses the multi-rank launch: use it with --gpus N > 1", file=sys.stderr)
            sys.exit(2)
        rehearsal = False
        a.no_kernel_timing = a.no_host_graph = a.no_cpu_baseline = a.no_shade_paths = True   # legs that are nothing but device work
    if world > 1 and not rehearsal and not DRY and torch.cuda.device_count() < int(os.environ.get("LOCAL_WORLD_SIZE", world)):
        # started as ranks by torch.distributed.run itself (the driver's launch form) on a box with fewer devices than ranks: the same
        # rehearsal mode spawn_ranks() would have chosen, instead of a crash in set_device
        if torch.cuda.device_count() < 1 or world > 5:
            print(f"bench.py: WORLD_SIZE={world} but {torch.cuda.device_count()} device(s) visible (a rehearsal on one GPU takes at most 5 ranks)", file=sys.stderr)
            sys.exit(2)
        rehearsal = True
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if DRY:
            dist.init_process_group("gloo")
        elif rehearsal:
            local_rank = 0
            torch.cuda.set_device(0)
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist = None
        torch.cuda.set_device(0)
    assert a.gpus == world, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    dev = local_rank if world > 1 else 0
    if DRY:
        from bench_dryrun import DryContext
        fault = None
        if a.dry_fail and a.dry_fail.split(":", 1)[0] == str(rank) and a.dry_fail.split(":", 1)[1] in ("miscount", "hang"):
            fault = a.dry_fail.split(":", 1)[1]
        ctx = DryContext(dist, rank, world, fault)
    else:
        ctx = PbrContext(dev)
    flag_dev = "cpu" if (rehearsal or DRY) else f"cuda:{dev}"

    # ---- the record so far, for a rank that has to give up in a later (optional) phase
    state = {"out": None}

    def emit(out):
        sys.stdout.flush()
        sys.stderr.flush()
        if rank == 0 and out is not None:
            os.write(json_fd, (json.dumps(out) + "\n").encode())

    def on_deadline(what):
        out = state["out"]
        if out is None:
            return 3
        out.setdefault("config", {}).setdefault("notes", []).append(f"gave up on '{what}' after {a.deadline:.0f} s; the record holds what was measured before")
        emit(out)
        return 4    # the (partial) record is on stdout, but a run whose collective hung must not look like a clean one

    wd = Watchdog()
    wd.on_fire = on_deadline
    job = Job(a, ctx, dist, rank, world, rehearsal, flag_dev, wd)
    job.lut, job.env, job.sh, job.sky_np = build_ibl(ctx, want_sky=True)

    # ---- collectives: the C ABI's own RCCL communicators first, torch.distributed (also RCCL) as the fallback
    if world > 1 and not rehearsal and a.transport == "capi":
        ok = True
        with wd.phase(a.deadline, "pbr_comm_init"):
            try:
                from direct12pbrrenderer_amd.api import comm_unique_id
                if DRY:
                    comm_unique_id = ctx.comm_unique_id   # noqa: F811 — 128 bytes from rank 0, broadcast and checked like the real id
                ids = [comm_unique_id() if rank == 0 else None]
                dist.broadcast_object_list(ids, src=0)
                ctx.comm_init(world, rank, ids[0])
            except Exception as e:   # noqa: BLE001 — any failure here means "use the other RCCL transport"
                ok = False
                job.notes.append(f"pbr_comm_init failed on rank {rank}: {e}")
            job.capi_comm = job.all_agree(ok)
        if not job.capi_comm:
            job.notes.append("C-ABI RCCL communicators unavailable: collectives through torch.distributed")

    # ---- workloads: which frame, which tile grid
    layout = parse_layout(a.layout) if a.layout else None

    def strong(fw, fh, grid):
        cols, rows = grid_for_world(world, grid)
        return ("strong", lambda r, halo: tile_of_frame(r, world, fw, fh, layout=(cols, rows), halo=halo), fw, fh, cols, rows)

    if a.frame:
        fw, fh = (int(v) for v in a.frame.lower().split("x"))
        primary = strong(fw, fh, layout or (CFG5_GRID.get(world) if (fw, fh) == CFG5_FRAME else None))
    else:
        cols, rows = grid_for_world(world, layout)
        tw, th = weak_tile(world, cols, rows, a.width, a.height)
        primary = ("weak", lambda r, halo: tile_for_rank(r, world, tw, th, layout=(cols, rows), halo=halo), cols * tw, rows * th, cols, rows)

    res = run_workload(job, "headline", primary[0], primary[1], primary[2], primary[3], primary[4], primary[5], a.steps, a.warmup, a.settle,
                       want_stage_timing=True)
    frame, spec, g, lights = res["frame_obj"], res["spec"], res["g"], res["lights"]
    out = {
        "metric": "shaded Mpixel/s at 4K G-buffer (full deferred frame: clustered shade + bloom + auto-exposure + ACES)",
        "value": round(res["value"], 2), "unit": "Mpixel/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(res["ms_per_step"], 4), "higher_is_better": True, "scaling": res["scaling"] if world > 1 else "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": workload_config(job, res, a.settle),
    }
    if rehearsal:
        out["rehearsal"] = "all ranks share cuda:0 (gloo + host copies): functional run, not a measurement"
    if DRY:
        out["dry_run"] = True
        out["data"] = "none (dry run: no device call, frames are no-ops; `value` counts them and measures nothing)"
    state["out"] = out     # complete as a contract line from here on; what follows adds blocks

    shaded_px = spec.sw * spec.sh
    if not a.no_kernel_timing:   # every rank runs the same stage sequence (halo stages are collective); rank 0 reports
        with wd.phase(a.deadline, "per-stage timing"):
            iters = max(5, min(a.steps, 20))
            ext_px = spec.ew * spec.eh
            int_px = spec.w * spec.h
            # bloom and the luminance histogram run fused in the frame (pbr_bloom_histogram); the two
            # un-fused stage calls are timed as well for reference
            stages = {
                "cluster": (frame.clustered, None, 0.0),
                "shade": (frame.shade, None, BYTES_PER_PX["shade"] * shaded_px),
                "bloom+histogram": (frame.bloom_histogram, None, BYTES_PER_PX["bloom"] * (int_px if spec.halo else ext_px) + BYTES_PER_PX["histogram"] * int_px),
                "histogram": (frame.histogram, None, BYTES_PER_PX["histogram"] * int_px),
                "average": (frame.average, frame.histogram, 0.0),
                "tonemap": (frame.tonemap, None, BYTES_PER_PX["tonemap"] * int_px),
                # pbr_average_tonemap: the two dispatches above as ONE launch — timed for the record only, the frame keeps the two
                # (DeferredFrame.fused_exposure is off: no gain at frame level, EXPERIMENTS.md round 4).  The call swaps the frame's
                # histogram / luminance cells, so it is made an even number of times (time_stage: 1 + iters calls; one more below)
                "average+tonemap": (frame.average_tonemap, frame.histogram, BYTES_PER_PX["tonemap"] * int_px),
            }
            if spec.halo:
                stages["halo prefilter"] = (frame.halo_prefilter, None, 0.0)
                stages["halo exchange"] = (frame.halo_exchange, None, 0.0)
                stages["halo pyramid+merge"] = (frame.halo_pyramid, None, 0.0)
            else:
                stages["bloom"] = (frame.bloom, None, BYTES_PER_PX["bloom"] * ext_px)
            frame.hist.zero_()
            kern = {}
            for name, (fn, pre, nbytes) in stages.items():
                ms = time_stage(fn, iters, pre)
                if name == "average+tonemap" and iters % 2 == 0:   # 1 + iters calls so far: restore the frame's own cells
                    frame.histogram()
                    fn()
                kern[name] = {"ms": round(ms, 4), "GB/s": round(nbytes / (ms * 1e-3) / 1e9, 1) if nbytes else None}
                frame.hist.zero_()
            hbm_meas = measure_hbm_read(ctx)
            if rehearsal:   # ranks sharing one card would measure each other's load: the normalisers mean nothing there
                valu_meas = {"skipped": "rehearsal: all ranks share one device"}
            else:
                try:
                    valu_meas = measure_valu(ctx)
                except Exception as e:   # noqa: BLE001 — reporting only
                    valu_meas = {"error": str(e)}
    if rank == 0 and not a.no_kernel_timing:
        # dominant kernel: the shade.  Its launch duration is the mean over the frames of the timed region; the
        # per-stage figures below come from separate isolated launches after it
        shade_events = res["shade_events"]
        shade_ms_in_frame = sum(e0.elapsed_time(e1) for e0, e1 in shade_events) / max(len(shade_events), 1)
        kern["shade(in frame)"] = {"ms": round(shade_ms_in_frame, 4), "GB/s": round(BYTES_PER_PX["shade"] * shaded_px / (shade_ms_in_frame * 1e-3) / 1e9, 1)}
        achieved = kern["shade(in frame)"]["GB/s"]
        frame_bytes = sum(BYTES_PER_PX.values()) * spec.w * spec.h
        frame_gbps = frame_bytes / (res["ms_per_step"] * 1e-3) / 1e9
        out["roofline"] = {"bound": "hbm", "kernel": "k_deferred_shade", "kernel_limited_by": "valu",
                           "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                           "traffic": pmc_traffic("k_deferred_shade", shaded_px),
                           "hbm_measured_GBps": round(hbm_meas, 1), "frac_of_measured": round(achieved / hbm_meas, 5),
                           # BASELINE's metric is "Mpixel/s ...; % HBM roofline" of the FRAME: all stages' algorithmic bytes over the timed step
                           "frame": {"bytes": int(frame_bytes), "bytes_per_pixel": sum(BYTES_PER_PX.values()), "ms": round(res["ms_per_step"], 4),
                                     "GBps": round(frame_gbps, 1), "frac": round(frame_gbps / HBM_PEAK_GBS, 4), "frac_of_measured": round(frame_gbps / hbm_meas, 4),
                                     "note": "SURVEY 8d frame total (shade 25 + bloom 75.75 + histogram 8 + tone-map 12 B/px) x this rank's interior pixels / "
                                             "ms_per_step of the timed region"},
                           "stage_ms": {k: v["ms"] for k, v in kern.items()},
                           "stage_GBps": {k: v["GB/s"] for k, v in kern.items()},
                           "stage_frac_of_measured": {k: round(v["GB/s"] / hbm_meas, 4) for k, v in kern.items() if v["GB/s"]},
                           "note": "dominant kernel = the shade, which with 256 clustered lights is bound by FP32 VALU ISSUE (kernel_limited_by: valu; the `valu` "
                                   "block — same bound / achieved / peak / unit / frac shape — prices it), so this block's achieved / peak / frac — the kernel's "
                                   "ALGORITHMIC bytes (SURVEY 8d: 25 B/px) over its HIP-event launch duration against HBM, which is what the schema's "
                                   "bound: hbm | mfma offers for a kernel without MFMA — are structurally low; `frame` is the whole step against HBM; "
                                   "hbm_measured_GBps = pbr_membench_read streaming 2 GiB on this device; "
                                   "stage_GBps of the fused bloom is EFFECTIVE (the 16 reference passes' bytes / the fused launches' time), not traffic; "
                                   "traffic (when present) is the committed rocprofv3 PMC figure for this workload and this version of shade.hip, "
                                   "IBL gathers served by L2/MALL included"}
        # what THIS box sustains, measured in this run (pbr_valubench at the shade's occupancy): VALU issue rate and shader clock per
        # instruction class — present whether or not a committed counter profile matches the kernel (the `valu` block below needs one)
        out["roofline"]["box"] = dict(valu_meas, hbm_read_GBps=round(hbm_meas, 1))
        if "packed_v_pk_fma_f32" in valu_meas:
            # the shade's launch in SHADER CYCLES (SIMD-cycles per pixel, at the clock the chip holds under packed-fp32 load — 46 of the walk's 52
            # instructions are packed): separates a slower box (lower clock, same cycles) from slower code (more cycles)
            clk_ = valu_meas["packed_v_pk_fma_f32"]["clock_GHz"] * 1e9
            out["roofline"]["box"]["shade_simd_cycles_per_pixel"] = round(shade_ms_in_frame * 1e-3 * clk_ * valu_meas["compute_units"] * 4 / shaded_px, 2)
        sq = kernel_entry(load_profile("pmc_sq_latest.json", shaded_px), "k_deferred_shade", ("SQ_INSTS_VALU",))
        if sq:
            # Supplementary compute roofline for the VALU-bound shade, against BOTH ceilings:
            #  * the guide's spec issue rate: a wave64 fp32 op every 2 cycles per SIMD at 2.4 GHz (157.3 TFLOP/s when every
            #    op is a packed FMA);
            #  * what tools/valu_rate3.hip measures on this part in shader-clock cycles (profiles/r02_valu_rate3.txt), at the
            #    kernel's own occupancy of 5 waves per SIMD: plain v_mul/v_add 3.0, v_fma 2.6, v_pk_{fma,mul,add}_f32 4.7,
            #    v_max/v_min 4.6, v_rcp/v_rsq 8.5 cycles per wave-instruction per SIMD, at the ~2.1 GHz the chip sustains
            #    under packed fp32 load.  issue_model = the launch's instruction mix (SQ_INSTS_VALU_* counters) priced at
            #    those costs: the time the VALU needs to ISSUE the kernel's instructions, as a share of the launch.
            nv = sq["SQ_INSTS_VALU"]
            rate = nv / (shade_ms_in_frame * 1e-3)
            # the plain-class issue rate and the sustained clocks of THIS box, measured in this run (pbr_valubench); the committed
            # round-2 constants only when the probe failed
            on_box = "plain_v_mul_f32" in valu_meas
            plain_peak = valu_meas["plain_v_mul_f32"]["Ginst_s"] * 1e9 if on_box else N_SIMD * MEASURED_CLOCK_HZ / MEASURED_CYCLES["plain"]
            valu = {"bound": "valu", "achieved": round(rate / 1e9, 1), "peak": round(N_SIMD * CLOCK_HZ / 2.0 / 1e9, 1), "unit": "Ginst/s (wave64 VALU instructions)",
                    "frac": round(rate / (N_SIMD * CLOCK_HZ / 2.0), 3),
                    "wave_insts_per_launch": nv, "achieved_Ginst_s": round(rate / 1e9, 1),
                    "spec_peak_Ginst_s": round(N_SIMD * CLOCK_HZ / 2.0 / 1e9, 1), "frac_of_spec": round(rate / (N_SIMD * CLOCK_HZ / 2.0), 3),
                    "measured_plain_peak_Ginst_s": round(plain_peak / 1e9, 1),
                    "frac_of_measured_plain": round(rate / plain_peak, 3),
                    "measured_on_this_box": "roofline.box" if on_box else False,
                    "source": "SQ_INSTS_VALU: profiles/pmc_sq_latest.json + pmc_shade_issue_latest.json (rocprofv3, same shade.hip); peaks and clocks: "
                              + ("pbr_valubench in this run, at the shade's occupancy" if on_box else "profiles/r02_valu_rate3.txt (the on-box probe failed)")}
            if on_box:
                # the shade's launch in SHADER CYCLES: what separates a slower box (lower sustained clock) from slower code.  The clock is
                # the one the chip holds under packed-fp32 load at the shade's occupancy (46 of its 52 loop instructions are packed)
                clk = valu_meas["packed_v_pk_fma_f32"]["clock_GHz"] * 1e9
                valu["shade_clock_GHz_assumed"] = round(clk / 1e9, 3)
                valu["shade_simd_cycles_per_pixel"] = round(shade_ms_in_frame * 1e-3 * clk * valu_meas["compute_units"] * 4 / shaded_px, 2)
                valu["shade_chip_cycles_per_pixel"] = round(shade_ms_in_frame * 1e-3 * clk / shaded_px, 5)
                valu["shade_cycles_per_wave_inst_per_simd"] = round(shade_ms_in_frame * 1e-3 * clk * valu_meas["compute_units"] * 4 / nv, 2)
            mix = kernel_entry(load_profile("pmc_shade_issue_latest.json", shaded_px), "k_deferred_shade",
                               ("SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_TRANS_F32"))
            if mix:
                fma, mul, add, trans = (mix["SQ_INSTS_VALU_FMA_F32"], mix["SQ_INSTS_VALU_MUL_F32"], mix["SQ_INSTS_VALU_ADD_F32"],
                                        mix["SQ_INSTS_VALU_TRANS_F32"])
                other = max(nv - fma - mul - add - trans, 0.0)
                # the counters do not tell packed from plain: price FMA/MUL/ADD once as all plain and once as all packed — at the issue
                # rates this box sustained per class in this run (each at the clock the chip holds under that class), else the round-2 constants
                if on_box:
                    r_ = {k: valu_meas[n]["Ginst_s"] * 1e9 for k, n in (("plain", "plain_v_mul_f32"), ("fma", "v_fma_f32"), ("packed", "packed_v_pk_fma_f32"), ("trans", "trans_v_rcp_f32"))}
                else:
                    r_ = {k: N_SIMD * MEASURED_CLOCK_HZ / v for k, v in MEASURED_CYCLES.items()}
                lo = (fma / r_["fma"] + (mul + add) / r_["plain"] + trans / r_["trans"] + other / r_["plain"]) * 1e3
                hi = ((fma + mul + add) / r_["packed"] + trans / r_["trans"] + other / r_["plain"]) * 1e3
                valu["issue_model_ms"] = [round(lo, 4), round(hi, 4)]
                valu["issue_model_frac_of_launch"] = [round(lo / shade_ms_in_frame, 3), round(hi / shade_ms_in_frame, 3)]
                # flop upper bound: every FMA/MUL/ADD wave-instruction counted as PACKED (2 results per lane)
                flop = 64.0 * 2.0 * (2.0 * fma + mul + add)
                tf = flop / (shade_ms_in_frame * 1e-3) / 1e12
                valu["TFLOPs_upper_bound"] = round(tf, 1)
                valu["frac_of_fp32_peak_upper_bound"] = round(tf / FP32_VALU_PEAK_TFLOPS, 3)
            out["roofline"]["valu"] = valu

    if world == 1 and not a.no_kernel_timing and not a.no_shade_paths:
        try:
            out["shade_ms_by_path"] = shade_paths_ab(job, res)
        except Exception as e:   # noqa: BLE001 — reporting only
            out["shade_ms_by_path"] = {"error": f"{type(e).__name__}: {e}"}

    if world == 1 and not a.no_configs and not DRY:
        try:
            out["configs"] = configs_block(job)
        except Exception as e:   # noqa: BLE001 — reporting only
            out["configs"] = {"error": f"{type(e).__name__}: {e}"}

    # ---- the drop-in path: the same workload driven by the C++ pass graph
    # (optional legs from here on: an exception on this rank is recorded, not raised — the contract line above is complete, and a
    # rank that falls out of step with the others is released by the deadline of the next collective phase)
    if not a.no_host_graph:
        try:
            hg = host_graph_leg(job, res, max(a.steps, 20))
        except Exception as e:   # noqa: BLE001
            hg = {"error": f"rank {rank}: {type(e).__name__}: {e}"}
        if rank == 0:
            out["host_graph"] = hg

    # ---- BASELINE configs[4] in the same launch: the 8K frame cut over the ranks (strong scaling) + its single-GPU time
    if world > 1 and not a.frame and not a.no_cfg5:
        del frame
        res["frame_obj"] = None
        dev_empty_cache()
        # (a rehearsal with a reduced tile budget exercises this leg on a reduced frame: it is a functional run either way)
        cfg5_frame = (1920, 1088) if (rehearsal and a.width < 3840) else CFG5_FRAME
        s5 = strong(cfg5_frame[0], cfg5_frame[1], CFG5_GRID.get(world))
        try:
            r5 = run_workload(job, "cfg5", s5[0], s5[1], s5[2], s5[3], s5[4], s5[5], a.steps, a.warmup, min(a.settle, 100), want_stage_timing=False)
            c5 = workload_config(job, r5, min(a.settle, 100))
            c5.update({"ms_per_step": round(r5["ms_per_step"], 4), "value": round(r5["value"], 2), "unit": "Mpixel/s", "scaling": "strong", "steps": a.steps})
            hg5 = None
            if not a.no_host_graph:
                try:
                    hg5 = host_graph_leg(job, r5, max(a.steps, 20))
                except Exception as e:   # noqa: BLE001
                    hg5 = {"error": f"rank {rank}: {type(e).__name__}: {e}"}
            r5["frame_obj"] = None
            dev_empty_cache()
            # the denominator: the same 8K frame on ONE GPU (rank 0, the others wait), so that the x-factor comes from one record
            single_ms = None
            with wd.phase(max(a.deadline, 600.0), "cfg5: single-GPU 8K frame on rank 0"):
                if rank == 0:
                    fw, fh = cfg5_frame
                    from direct12pbrrenderer_amd.pipeline import TileSpec
                    fr1 = DeferredFrame(ctx, TileSpec(0, 0, fw, fh, fw, fh, 0), r5["g"], r5["lights"], job.lut, LUT_RES, job.env, ENV_SIZE, ENV_MIPS)
                    if DRY:
                        fr1.upload_gbuffer({k: np.zeros((fh, fw), dtype=d) for k, d in (("A", np.uint32), ("B", np.uint32), ("C", np.uint32), ("depth", np.float32), ("stencil", np.uint8))})
                    else:
                        fr1.upload_gbuffer(synth.gbuffer_tile(0, 0, fw, fh, fw, fh))
                    fr1.set_prev_luminance(0.18)
                    for _ in range(30 + a.warmup):
                        fr1.render()
                    dev_sync()
                    t0 = time.perf_counter()
                    for _ in range(a.steps):
                        fr1.render()
                    dev_sync()
                    single_ms = (time.perf_counter() - t0) / a.steps * 1e3
                    del fr1
                if dist:
                    dist.barrier()
            if rank == 0:
                c5["single_gpu_ms_per_step"] = round(single_ms, 4)
                c5["single_gpu_value"] = round(r5["total_px"] / (single_ms * 1e-3) / 1e6, 2)
                c5["speedup_vs_single_gpu"] = round(single_ms / r5["ms_per_step"], 3)
                if r5["ms_overlapped"] is not None:
                    c5["speedup_vs_single_gpu_tail_overlapped"] = round(single_ms / r5["ms_overlapped"], 3)
                if hg5 is not None:
                    c5["host_graph"] = hg5
                out["config"]["cfg5"] = c5
        except SystemExit as e:
            out["config"]["cfg5"] = {"error": str(e)}
        except Exception as e:   # noqa: BLE001
            out["config"]["cfg5"] = {"error": f"rank {rank}: {type(e).__name__}: {e}"}

    if rank == 0 and not a.no_cpu_baseline and world == 1:
        try:
            lut_np = job.lut.cpu().view(torch.int16).numpy().view(np.float16)
            env_np = job.env.cpu().view(torch.int16).numpy().view(np.float16)
            probe = shade_parity_probe(ctx, g, lights, job.lut, job.env, lut_np, env_np, a.width, a.height)
            out["cpu_baseline"], parity = cpu_baseline(g, lights, lut_np, env_np, a.width, a.height, parity_probe=probe)
            if parity is not None:
                out["parity"] = parity
        except Exception as e:   # the baseline is reporting only; never fail the GPU measurement on it
            out["cpu_baseline"] = {"value": None, "unit": "Mpixel/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}
    with wd.phase(a.deadline, "teardown"):
        if dist:
            dist.barrier()
            dist.destroy_process_group()
        ctx.close()
    # the record is the last thing this process writes (teardown messages of the libraries, if any, come before it)
    emit(out)
    os.close(json_fd)


if __name__ == "__main__":
    main()
