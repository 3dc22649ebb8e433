"""Shade time over render-target sizes and light counts — the table a change of the shade's launch schedule is judged on:
    python tools/shade_tile_ms.py [label] [WxH[@FWxFH] ...]
Default sizes: 1440x960 (the reference's own target, App.h:77-78), 1920x1080 (BASELINE cfg2), 1928x2168 (a cfg5 rank's tile + 4 px,
inside the 7680x4320 frame), 3840x2160 (cfg4), 7680x4320 (cfg5 on one GPU); each with 1 light (the reference scene's light_1) and with
256 clustered lights.  Back-to-back launches (30 per batch, best of 3 batches, HIP events on the kernels' stream); one JSON line per
cell + a fitted  t = intercept + pixels / rate  per light count.  Knobs build: PBR_SHADE_SCHED=grid selects the round-5 static grid."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from direct12pbrrenderer_amd import scene, synth  # noqa: E402
from direct12pbrrenderer_amd.api import PbrContext  # noqa: E402
from direct12pbrrenderer_amd.pipeline import DeferredFrame, TileSpec  # noqa: E402

label = sys.argv[1] if len(sys.argv) > 1 else ""
sizes = sys.argv[2:] or ["1440x960", "1920x1080", "1928x2168@7680x4320", "3840x2160", "7680x4320"]
ctx = PbrContext(0)
lut, env, sh = bench.build_ibl(ctx)
rows = {1: [], 256: []}
for sz in sizes:
    wh, _, full = sz.partition("@")
    W, H = (int(v) for v in wh.split("x"))
    FW, FH = (int(v) for v in full.split("x")) if full else (W, H)
    x0 = 1920 if full else 0   # an inner tile of the cut frame
    cam = scene.Camera.reference_default(FW, FH)
    g = scene.make_global(cam, FW, FH, sh_pack=sh)
    spec = TileSpec(x0, 0, W, H, FW, FH, 0)
    gb = synth.gbuffer_tile(spec.sx0, spec.sy0, spec.sw, spec.sh, FW, FH)
    for n in (1, 256):
        lights = synth.reference_scene_light() if n == 1 else synth.lights_in_view_box(n, cam)
        fr = DeferredFrame(ctx, spec, g, lights, lut, 512, env, 512, 5)
        fr.upload_gbuffer(gb)
        fr.clustered()
        for _ in range(20):
            fr.shade()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30):
                fr.shade()
            e1.record()
            e1.synchronize()
            best = min(best, e0.elapsed_time(e1) / 30)
        rows[n].append((W * H, best))
        print(json.dumps({"label": label, "size": sz, "lights": n, "shade_ms": round(best, 4), "gpixel_s": round(W * H / best / 1e6, 2)}), flush=True)
        del fr
    del gb
    torch.cuda.empty_cache()
for n, r in rows.items():
    if len(r) >= 2:
        px, ms = np.array([a for a, _ in r], dtype=np.float64), np.array([b for _, b in r], dtype=np.float64)
        slope, icpt = np.polyfit(px, ms, 1)
        print(json.dumps({"label": label, "lights": n, "fit_intercept_us": round(icpt * 1e3, 1), "fit_gpixel_s": round(1e-6 / slope, 2)}), flush=True)
