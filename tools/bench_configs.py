#!/usr/bin/env python3
"""Times the BASELINE.json configs other than the bench.py headline (cfg4) on one MI355X:
cfg1 LUT 256^2 / 512^2, cfg2 1080p shade with 1 point light (+ full frame), cfg3 512^2 prefilter (5 mips, 1 024 spp)
+ SH9, cfg5 one rank's share of the 8K frame (the busiest 1920x2160 tile of the 2 rows x 4 cols layout) in apron mode
(tile + 256-px apron, everything measured) and in halo mode (tile + 4 px; the neighbours' level-1 strips cannot arrive on a
one-GPU box, so the exchange itself is not in the figure — its plane is filled once before the clock).  One JSON line per config."""
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


def emit(**kw):
    print(json.dumps(kw), flush=True)
    out.append(kw)


# ---- cfg1: split-sum BRDF LUT
for res in (256, 512):
    buf = ctx.empty((res, res, 2), torch.float16)
    ms = bench.time_stage(lambda: ctx.brdf_lut(res, out=buf), 10)
    emit(config="cfg1", what=f"BRDF LUT {res}x{res}, 1024 spp", ms=round(ms, 4), Msamples_per_s=round(res * res * 1024 / ms / 1e3, 1),
         Mtexels_per_s=round(res * res / ms / 1e3, 2))

# ---- cfg3: prefilter + SH9 on the 512^2 cube
sky_mips = 10
sky = ctx.upload(synth.env_cube(512, sky_mips))
ctx.cube_gen_mips(sky, 512, sky_mips)
envbuf = ctx.prefilter_env(sky, 512, sky_mips, 512, ENV_MIPS)
ms = bench.time_stage(lambda: ctx.prefilter_env(sky, 512, sky_mips, 512, ENV_MIPS, out=envbuf), 5)
texels = 6 * sum((512 >> m) ** 2 for m in range(5))
emit(config="cfg3", what="GGX prefilter 512^2 cube, 5 mips, 1024 spp", ms=round(ms, 3), Gsamples_per_s=round(texels * 1024 / ms / 1e6, 2),
     Mtexels_per_s=round(texels / ms / 1e3, 2))
shbuf = ctx.empty((28,), torch.float32)
ms = bench.time_stage(lambda: ctx.sh9_project(sky, 512, sky_mips, out=shbuf), 10)
emit(config="cfg3", what="SH9 projection of the 512^2 cube (quadrature, 25.2 MB in)", ms=round(ms, 4), GBps=round(6 * 512 * 512 * 16 / ms / 1e6, 1))
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
    emit(config=name, what=what, shade_ms=round(shade, 4), bloom_histogram_ms=round(bloom, 4), frame_ms=round(full, 4),
         shade_Mpixel_per_s=round(shaded / shade / 1e3, 1),
         shade_GBps_algorithmic=round(25.0 * shaded / shade / 1e6, 1), frame_Mpixel_per_s_interior=round(inner / full / 1e3, 1),
         shaded_pixels=shaded, bloom_pixels=spec.ew * spec.eh, interior_pixels=inner)


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
