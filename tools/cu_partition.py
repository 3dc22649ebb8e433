"""Throughput mode with the compute units PARTITIONED between the frame's two streams (pbr_ctx_set_cu_masks): frame i's bloom chain +
average + tone-map on the side stream's CUs while frame i + 1's cluster pass + shade have the others to themselves.
python tools/cu_partition.py [side CU counts ...]   (default 0 24 32 40 48 64; 0 = no partition: the priority side stream)
Prints ms per frame in order (on the main stream's CUs alone) and in post-shade throughput mode, and checks that the overlapped
frames are the in-order frames.
pbr_ctx_set_cu_masks is an entry point of the KNOBS build only (round 6): run with PBR_HIP_LIB=$PWD/direct12pbrrenderer_amd/libpbr_hip_knobs.so."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time  # noqa: E402

import torch  # noqa: E402

import bench  # noqa: E402
from direct12pbrrenderer_amd import scene, synth  # noqa: E402
from direct12pbrrenderer_amd.api import PbrContext  # noqa: E402
from direct12pbrrenderer_amd.pipeline import DeferredFrame, tile_for_rank  # noqa: E402

ctx = PbrContext(0)
lut, env, sh = bench.build_ibl(ctx)
W, H = 3840, 2160
cam = scene.Camera.reference_default(W, H)
g = scene.make_global(cam, W, H, sh_pack=sh, delta_time=1.0 / 60.0)
lights = synth.lights_in_view_box(256, cam)
gb = synth.gbuffer_tile(0, 0, W, H, W, H)
torch.cuda.synchronize()
ctx.use_own_stream()


def make(overlap):
    fr = DeferredFrame(ctx, tile_for_rank(0, 1, W, H), g, lights, lut, 512, env, 512, 5)
    fr.upload_gbuffer(gb)
    fr.set_prev_luminance(0.18)
    torch.cuda.synchronize()
    if overlap:
        fr.enable_tail_overlap(from_bloom=True)
    return fr


def ms_per_frame(fr, n=200, settle=200):
    for _ in range(settle):
        fr.render()
    fr.finish(); ctx.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fr.render()
    fr.finish(); ctx.sync(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def signature(fr, frames=5):
    fr.set_prev_luminance(0.18); fr.hist.zero_(); torch.cuda.synchronize()
    for _ in range(frames):
        fr.render()
    fr.finish(); ctx.sync(); torch.cuda.synchronize()
    return float(fr.avg.cpu()[0]), int(fr.ldr.to(torch.int64).sum().item())


layout = "low"
args = sys.argv[1:]
if args and args[0] in ("low", "strided", "per_xcd"):
    layout, args = args[0], args[1:]
counts = [int(v) for v in args] or [0, 16, 24, 32, 40, 48, 64]
ref = None
for side in counts:
    ctx.partition_cus(side, layout=layout, allow_uneven=True)   # this tool measures the uneven shares too
    a, b = make(False), make(True)
    sa, sb = signature(a), signature(b)
    ref = ref or sa
    ok = sa == ref and sb == ref
    t_in, t_ov = ms_per_frame(a), ms_per_frame(b)
    print(f"[{layout}] side CUs {side:3d}: in order {t_in:.4f} ms/frame, post-shade throughput mode {t_ov:.4f} ms/frame "
          f"({8294400 / t_ov / 1e3:.0f} Mpixel/s); same frames as the unpartitioned in-order render: {ok}", flush=True)
    del a, b
    torch.cuda.empty_cache()
ctx.partition_cus(0)
ctx.close()
