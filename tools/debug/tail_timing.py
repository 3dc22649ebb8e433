"""Experiment: where a tile of the wide bloom tail kernel spends its shader-clock cycles (needs the PBR_BLOOM_TIMING build of
bloom.hip: tools/debug/bin/libpbr_hip_timing.so, loaded through PBR_HIP_LIB).  Prints mean cycle deltas between the stamps
0 tile start | 1 loads issued + samples... | 2 all global loads arrived | 3 H phase done | 4 barrier passed | 5 V phase + tail done | 6 end"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from direct12pbrrenderer_amd import scene, synth, _lib
from direct12pbrrenderer_amd.api import PbrContext
from direct12pbrrenderer_amd.pipeline import DeferredFrame, tile_for_rank
import bench
ctx = PbrContext(0)
lut, env, sh = bench.build_ibl(ctx)
W, H = 3840, 2160
spec = tile_for_rank(0, 1, W, H)
cam = scene.Camera.reference_default(W, H)
g = scene.make_global(cam, W, H, sh_pack=sh)
fr = DeferredFrame(ctx, spec, g, synth.lights_in_view_box(256, cam), lut, 512, env, 512, 5)
fr.upload_gbuffer(synth.gbuffer_tile(0, 0, W, H, W, H))
fr.set_prev_luminance(0.18)
for _ in range(20):
    fr.render()
torch.cuda.synchronize()
lib = ctypes.CDLL(_lib.LIB_PATH)
n = 8 * 1020
buf = (ctypes.c_ulonglong * n)()
assert lib.pbr_debug_tail_stamps(buf, n) == 0
a = np.array(buf, dtype=np.uint64).reshape(-1, 8).astype(np.int64)
a = a[(a[:, 0] > 0) & (a[:, 6] > a[:, 0])]          # blocks that stamped
names = ["0 tile start", "1 loads issued, samples computed", "2 all global loads arrived", "3 H pass done", "4 barrier passed",
         "5 V pass + merge + histogram done", "6 end-of-tile barrier passed"]
d = np.diff(a[:, :7], axis=1)
print("blocks", len(a), "total cycles per tile: mean", (a[:, 6] - a[:, 0]).mean().round(0), "median", np.median(a[:, 6] - a[:, 0]))
for i in range(6):
    print(f"  {names[i]:36s} -> {names[i + 1][:1]}: mean {d[:, i].mean():8.0f}  median {np.median(d[:, i]):8.0f}  p90 {np.percentile(d[:, i], 90):8.0f}")
