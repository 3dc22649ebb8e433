#!/usr/bin/env python3
"""Times the BASELINE.json configs other than the bench.py headline (cfg4) on one MI355X:
cfg1 LUT 256^2 / 512^2, cfg2 1080p shade with 1 point light (+ full frame), cfg3 512^2 prefilter (5 mips, 1 024 spp)
+ SH9, cfg5 one rank's share of the 8K frame (the busiest 1920x2160 tile of the 2 rows x 4 cols layout) in apron mode
(tile + 256-px apron, everything measured) and in halo mode (tile + 4 px; the neighbours' level-1 strips cannot arrive on a
one-GPU box, so the exchange itself is not in the figure — its plane is filled once before the clock).  One JSON line per config.
cfg1 - cfg3 also get their CPU leg: the oracle (kind "port") on all host threads, median of 5, on the
whole config where that takes seconds and on a stated sample where it does not.  `--no-cpu` skips those legs."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from direct12pbrrenderer_amd import scene, synth  # noqa: E402
from direct12pbrrenderer_amd.api import PbrContext  # noqa: E402
from direct12pbrrenderer_amd.pipeline import DeferredFrame, HaloTransport, TileSpec, parse_layout, tile_of_frame  # noqa: E402
from direct12pbrrenderer_amd.structs import ENV_MIPS  # noqa: E402

ctx = PbrContext(0)
out = []
CPU = "--no-cpu" not in sys.argv


def cpu_median(fn, reps=5):
    """median wall time (s) of `fn` over `reps` runs after one warm-up — the oracle uses every host thread (OpenMP)"""
    import time
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2]


def kept_samples(roughness, n=1024):
    """How many of the n Hammersley / GGX samples of env_map_gen.hlsl have N.L > 0 at this roughness (V = N): the others
    contribute nothing and the table-driven kernel drops them."""
    i = np.arange(n, dtype=np.uint32)
    bits = i.copy()
    bits = (bits << 16) | (bits >> 16)
    bits = ((bits & 0x55555555) << 1) | ((bits & 0xAAAAAAAA) >> 1)
    bits = ((bits & 0x33333333) << 2) | ((bits & 0xCCCCCCCC) >> 2)
    bits = ((bits & 0x0F0F0F0F) << 4) | ((bits & 0xF0F0F0F0) >> 4)
    bits = ((bits & 0x00FF00FF) << 8) | ((bits & 0xFF00FF00) >> 8)
    xi_y = bits.astype(np.float32) * np.float32(2.3283064365386963e-10)
    a = np.float32(roughness) * np.float32(roughness)
    hz = np.sqrt((np.float32(1.0) - xi_y) / (np.float32(1.0) + (a * a - np.float32(1.0)) * xi_y))
    return int((np.float32(2.0) * hz * hz - np.float32(1.0) > 0).sum())



def emit(**kw):
    print(json.dumps(kw), flush=True)
    out.append(kw)


# ---- cfg1: split-sum BRDF LUT
for res in (256, 512):
    buf = ctx.empty((res, res, 2), torch.float16)
    ms = bench.time_stage(lambda: ctx.brdf_lut(res, out=buf), 10)
    rec = dict(config="cfg1", what=f"BRDF LUT {res}x{res}, 1024 spp", ms=round(ms, 4), Msamples_per_s=round(res * res * 1024 / ms / 1e3, 1),
               Mtexels_per_s=round(res * res / ms / 1e3, 2))
    if CPU:
        from oracle import binding as orc
        dt = cpu_median(lambda: orc.brdf_lut(res))
        rec["cpu_baseline"] = {"value": round(res * res * 1024 / dt / 1e6, 1), "unit": "Msamples/s", "ms": round(dt * 1e3, 2), "cores": orc.num_threads(), "kind": "port",
                               "sample": f"the whole {res}x{res} plane (oracle/pbr_oracle.cpp orc_brdf_lut, OpenMP), median of 5"}
    emit(**rec)

# ---- cfg3: prefilter + SH9 on the 512^2 cube
sky_mips = 10
sky = ctx.upload(synth.env_cube(512, sky_mips))
ctx.cube_gen_mips(sky, 512, sky_mips)
envbuf = ctx.prefilter_env(sky, 512, sky_mips, 512, ENV_MIPS)
ms = bench.time_stage(lambda: ctx.prefilter_env(sky, 512, sky_mips, 512, ENV_MIPS, out=envbuf), 5)
texels = 6 * sum((512 >> m) ** 2 for m in range(5))
# what the kernel EVALUATES: mip 0 (roughness 0) is one fetch per texel; mips 1..4 loop over the samples with N.L > 0
evaluated = 6 * 512 * 512 + sum(6 * (512 >> m) ** 2 * kept_samples(m / 4.0) for m in range(1, 5))
rec = dict(config="cfg3", what="GGX prefilter 512^2 cube, 5 mips, 1024 spp", ms=round(ms, 3),
           Gsamples_per_s_reference_equivalent=round(texels * 1024 / ms / 1e6, 2), Gsamples_per_s_evaluated=round(evaluated / ms / 1e6, 2),
           evaluated_samples=evaluated, reference_samples=texels * 1024, Mtexels_per_s=round(texels / ms / 1e3, 2),
           note="reference_equivalent = output texels x 1024 (what env_map_gen.hlsl loops over); evaluated = fetches the kernel performs (mip 0: 1 per texel; "
                "mips 1-4: the samples with N.L > 0)")
if CPU:
    from oracle import binding as orc
    sky_np = sky.cpu().numpy()
    rng = np.random.default_rng(7)
    n_s = 2048
    picks = {m: rng.integers(0, 6 * (512 >> m) ** 2, n_s).astype(np.uint32) for m in range(5)}
    import time
    orc.prefilter_env_texels(sky_np, 512, sky_mips, 512, ENV_MIPS, 1, picks[1][:256])   # warm-up (OpenMP team, page faults)
    # each mip's sample scaled to the mip's texel count (the shader's loop is 1024 iterations per texel on every mip)
    est = 0.0
    for m in range(5):
        t0 = time.perf_counter()
        orc.prefilter_env_texels(sky_np, 512, sky_mips, 512, ENV_MIPS, m, picks[m])
        est += (time.perf_counter() - t0) / n_s * 6 * (512 >> m) ** 2
    rec["cpu_baseline"] = {"value": round(texels * 1024 / est / 1e9, 3), "unit": "Gsamples/s (reference-equivalent)", "ms_estimated_whole_config": round(est * 1e3, 1),
                           "cores": orc.num_threads(), "kind": "port",
                           "sample": f"{n_s} random texels of each of the 5 mips (orc_prefilter_env_texels, OpenMP), each mip's time scaled to its texel count"}
emit(**rec)
# the same on a source chain whose texels are half values — what the reference's BC6H_UF16 sky assets decode to: pbr_prefilter_env
# then samples mips >= 1 from its (exact) half-precision copy instead of the fp32 chain
sky_h = sky.half().float()
ms_h = bench.time_stage(lambda: ctx.prefilter_env(sky_h, 512, sky_mips, 512, ENV_MIPS, out=envbuf), 5)
emit(config="cfg3", what="GGX prefilter 512^2 cube, 5 mips, 1024 spp, source chain half-representable (as BC6H_UF16 assets decode)", ms=round(ms_h, 3),
     Gsamples_per_s_reference_equivalent=round(texels * 1024 / ms_h / 1e6, 2), Gsamples_per_s_evaluated=round(evaluated / ms_h / 1e6, 2))
shbuf = ctx.empty((28,), torch.float32)
ms = bench.time_stage(lambda: ctx.sh9_project(sky, 512, sky_mips, out=shbuf), 10)
rec = dict(config="cfg3", what="SH9 projection of the 512^2 cube (quadrature, 25.2 MB in)", ms=round(ms, 4), GBps=round(6 * 512 * 512 * 16 / ms / 1e6, 1),
           frac_of_8TBps=round(6 * 512 * 512 * 16 / ms / 1e6 / 8000.0, 3))
if CPU:
    dt = cpu_median(lambda: orc.sh9_project(sky_np, 512))
    rec["cpu_baseline"] = {"value": round(6 * 512 * 512 * 16 / dt / 1e9, 2), "unit": "GB/s", "ms": round(dt * 1e3, 2), "cores": orc.num_threads(), "kind": "port",
                           "sample": "the whole 512^2 cube (orc_sh9_project quadrature, OpenMP), median of 5"}
emit(**rec)
ms = bench.time_stage(lambda: ctx.env_pad(envbuf, 512, ENV_MIPS), 10)
emit(config="cfg3", what="env_pad (padded copy of the prefiltered chain)", ms=round(ms, 4))

# ---- cfg2 / cfg5: frames
lut, env, sh = bench.build_ibl(ctx)


class _NoExchange(HaloTransport):
    """One-GPU stand-in: the level-1 plane was filled once (zeros outside the interior); nothing travels."""
    def __init__(self):
        self.kind = "none"

    def exchange(self, fr):
        return


def frame_times(name, spec, n_lights, what, cell=1, all_specs=None, rank=0, overlap=False):
    cam = scene.Camera.reference_default(spec.full_w, spec.full_h)
    g = scene.make_global(cam, spec.full_w, spec.full_h, sh_pack=sh, delta_time=1.0 / 60.0)
    lights = synth.reference_scene_light() if n_lights == 1 else synth.lights_in_view_box(n_lights, cam)
    fr = DeferredFrame(ctx, spec, g, lights, lut, 512, env, 512, ENV_MIPS, all_specs=all_specs, rank=rank,
                       halo_transport=_NoExchange() if spec.halo else None, overlap=overlap)
    fr.upload_gbuffer(synth.gbuffer_tile(spec.sx0, spec.sy0, spec.sw, spec.sh, spec.full_w, spec.full_h, cell=cell))
    fr.set_prev_luminance(0.18)
    for _ in range(30):     # clocks up: the small tiles are over before the device has ramped
        fr.render()
    shade = bench.time_stage(fr.shade, 20)
    bloom = bench.time_stage(fr.bloom_histogram, 20)
    fr.hist.zero_()
    full = bench.time_stage(fr.render, 20)
    shaded, inner = spec.sw * spec.sh, spec.w * spec.h
    rec = dict(config=name, what=what, shade_ms=round(shade, 4), bloom_histogram_ms=round(bloom, 4), frame_ms=round(full, 4),
               shade_Mpixel_per_s=round(shaded / shade / 1e3, 1),
               shade_GBps_algorithmic=round(25.0 * shaded / shade / 1e6, 1), frame_Mpixel_per_s_interior=round(inner / full / 1e3, 1),
               shaded_pixels=shaded, bloom_pixels=spec.ew * spec.eh, interior_pixels=inner)
    if CPU and name == "cfg2":   # the 1080p / 1-light shade on the host cores (oracle, whole frame)
        from oracle import binding as orc
        from direct12pbrrenderer_amd.structs import Tile
        gb = synth.gbuffer_tile(0, 0, spec.w, spec.h, spec.w, spec.h)
        lut_np = lut.cpu().view(torch.int16).numpy().view(np.float16)
        env_np = env.cpu().view(torch.int16).numpy().view(np.float16)
        def shade_cpu():
            cl = orc.cluster_build(g)
            orc.cluster_cull(g, lights, cl)
            orc.deferred_shade(g, Tile(0, 0, spec.w, spec.h, spec.w, spec.h), gb, lut_np, env_np, 512, ENV_MIPS, cl, lights)
        dt = cpu_median(shade_cpu)
        rec["cpu_baseline"] = {"value": round(spec.w * spec.h / dt / 1e6, 2), "unit": "Mpixel/s (shade only)", "ms": round(dt * 1e3, 1), "cores": orc.num_threads(), "kind": "port",
                               "sample": "the whole 1920x1080 frame: cluster build + cull + deferred shade (oracle, OpenMP), inputs synthesised outside the clock, median of 5"}
    emit(**rec)


frame_times("cfg2", TileSpec(0, 0, 1920, 1080, 1920, 1080, 0), 1, "1920x1080 G-buffer, 1 point light + IBL")
frame_times("cfg4", TileSpec(0, 0, 3840, 2160, 3840, 2160, 0), 256, "3840x2160 G-buffer, 256 clustered lights + IBL (bench.py headline)")
frame_times("cfg4-coherent", TileSpec(0, 0, 3840, 2160, 3840, 2160, 0), 256,
            "3840x2160, 256 clustered lights + IBL, spatially coherent G-buffer (16x16-pixel surface patches) — NOT the BASELINE workload, for reference", cell=16)
lay = parse_layout("2x4")
for halo in (False, True):
    specs5 = [tile_of_frame(r, 8, 7680, 4320, layout=lay, halo=halo) for r in range(8)]
    frame_times("cfg5-halo" if halo else "cfg5-apron", specs5[1], 256,
                "busiest of 8 ranks of the 7680x4320 frame (2 rows x 4 cols): 1920x2160 tile, " +
                ("shaded +4 px, bloom on tile + 256-px halo of level 1 (exchange itself not included: one GPU)" if halo else "shaded and bloomed with a 256-px apron"),
                all_specs=specs5, rank=1)
specs5 = [tile_of_frame(r, 8, 7680, 4320, layout=lay, halo=True) for r in range(8)]
frame_times("cfg5-halo-ring-first", specs5[1], 256,
            "the same tile, border ring shaded first on the high-priority side stream (2 shade + 2 prefilter launches instead of 1 + 1) so that the exchange "
            "can overlap the core's shade (exchange itself not included: one GPU) — the single-GPU cost of the split", all_specs=specs5, rank=1, overlap=True)
frame_times("cfg5-1gpu", TileSpec(0, 0, 7680, 4320, 7680, 4320, 0), 256, "the whole 7680x4320 frame on one GPU (the strong-scaling denominator)")
# bench.py's DEFAULT multi-GPU workload (weak scaling: N x 8.3 Mpixel, 16:9): a rank's share of the N = 2 and N = 8 frames, halo mode
from direct12pbrrenderer_amd.pipeline import grid_for_world, tile_for_rank  # noqa: E402
for n_ranks, busiest in ((2, 0), (8, 1)):
    cols, rows = grid_for_world(n_ranks)
    tw, th = bench.weak_tile(n_ranks, cols, rows, 3840, 2160)
    specs_w = [tile_for_rank(r, n_ranks, tw, th, layout=(cols, rows), halo=True) for r in range(n_ranks)]
    frame_times(f"weak-N{n_ranks}-halo", specs_w[busiest], 256,
                f"bench.py --gpus {n_ranks} (weak scaling): rank {busiest}'s {tw}x{th} tile of the {specs_w[0].full_w}x{specs_w[0].full_h} frame, halo mode "
                "(exchange and all-reduce not included: one GPU); compare with cfg4 = the N = 1 frame", all_specs=specs_w, rank=busiest)

# ---- SURVEY 8f "next" rows at the headline size: G-buffer encode (48 B in + 12 B out per pixel) and a full-screen sky
W4, H4 = 3840, 2160
band = synth.material_tile(0, 0, W4, 216, W4, H4)                      # tile the planes from one band (host time)
m = [ctx.upload(np.ascontiguousarray(np.tile(p, (10, 1, 1)))) for p in band]
A4, B4, C4 = (ctx.zeros((H4, W4), torch.int32) for _ in range(3))
ms = bench.time_stage(lambda: ctx.gbuffer_encode(m[0], m[1], m[2], W4, H4, W4, A4, B4, C4), 20)
emit(config="8f", what="G-buffer encode 3840x2160 (gbuffer.hlsl ps_main; 60 B/pixel)", ms=round(ms, 4),
     Mpixel_per_s=round(W4 * H4 / ms / 1e3, 1), GBps_algorithmic=round(60.0 * W4 * H4 / ms / 1e6, 1))
del m
cam4 = scene.Camera.reference_default(W4, H4)
g4 = scene.make_global(cam4, W4, H4, sh_pack=sh)
from direct12pbrrenderer_amd.structs import Tile  # noqa: E402
st0 = ctx.zeros((H4, W4), torch.uint8)
hdr4 = ctx.zeros((H4, W4, 4), torch.float16)
ms = bench.time_stage(lambda: ctx.skybox(g4, Tile(0, 0, W4, H4, W4, H4), sky, 512, sky_mips, st0, W4, hdr4, W4), 20)
emit(config="8f", what="skybox resolve 3840x2160, every pixel sky (512^2 fp32 cube; 9 B/pixel + cube reads)", ms=round(ms, 4),
     Mpixel_per_s=round(W4 * H4 / ms / 1e3, 1))
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "bench_configs.json"), "w"), indent=1)
