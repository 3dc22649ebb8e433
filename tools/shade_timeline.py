"""Block / work-item timeline of the deferred shade, from 100 MHz wall-clock stamps the kernel writes in an experiment build:
    bash tools/build_variant.sh timing shade -DPBR_SHADE_TIMING -DPBR_DEBUG_KNOBS
    PBR_HIP_LIB=tools/ab/libpbr_timing.so [PBR_SHADE_SCHED=grid] python tools/shade_timeline.py [label] [WxH[@FWxFH] ...]
Per (size, lights): the launch's span (first item start -> last item end), when the resident slots were filled, how busy they were over
the launch (items running, in tenths of the span), what an item spends staging its lists, and the tail (time from the moment fewer than
half of the peak number of items are running to the end).  The stamps cost a few global stores per item: spans are 1-2 % above the
product kernel's."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from direct12pbrrenderer_amd import _lib, scene, synth  # noqa: E402
from direct12pbrrenderer_amd.api import PbrContext  # noqa: E402
from direct12pbrrenderer_amd.pipeline import DeferredFrame, TileSpec  # noqa: E402

label = sys.argv[1] if len(sys.argv) > 1 else ""
sizes = sys.argv[2:] or ["1920x1080", "1928x2168@7680x4320", "3840x2160"]
raw = C.CDLL(_lib.LIB_PATH)
raw.pbr_debug_shade_stamps.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
NB, NI = 8192, 160000
ctx = PbrContext(0)
lut, env, sh = bench.build_ibl(ctx)
for sz in sizes:
    wh, _, full = sz.partition("@")
    W, H = (int(v) for v in wh.split("x"))
    FW, FH = (int(v) for v in full.split("x")) if full else (W, H)
    cam = scene.Camera.reference_default(FW, FH)
    g = scene.make_global(cam, FW, FH, sh_pack=sh)
    spec = TileSpec(1920 if full else 0, 0, W, H, FW, FH, 0)
    gb = synth.gbuffer_tile(spec.sx0, spec.sy0, spec.sw, spec.sh, FW, FH)
    for n in (1, 256):
        lights = synth.reference_scene_light() if n == 1 else synth.lights_in_view_box(n, cam)
        fr = DeferredFrame(ctx, spec, g, lights, lut, 512, env, 512, 5)
        fr.upload_gbuffer(gb)
        fr.clustered()
        for _ in range(20):
            fr.shade()
        torch.cuda.synchronize()
        assert raw.pbr_debug_shade_stamps_reset() == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fr.shade()
        e1.record()
        e1.synchronize()
        blocks = np.zeros((NB, 4), dtype=np.uint64)
        items = np.zeros((NI, 4), dtype=np.uint64)
        assert raw.pbr_debug_shade_stamps(blocks.ctypes.data, NB, items.ctypes.data, NI) == 0
        it = items[items[:, 2] > 0].astype(np.int64)
        t0 = it[:, 0].min()
        st, lg, en = (it[:, 0] - t0) * 0.01, (it[:, 1] - t0) * 0.01, (it[:, 2] - t0) * 0.01   # microseconds
        span = en.max()
        rows = it[:, 3] >> 48
        # items running over time
        ev = np.concatenate([np.stack([st, np.ones_like(st)], 1), np.stack([en, -np.ones_like(en)], 1)])
        ev = ev[np.argsort(ev[:, 0], kind="stable")]
        run = np.cumsum(ev[:, 1])
        peak = run.max()
        edges = np.linspace(0, span, 11)
        tenths = []
        for a, b in zip(edges[:-1], edges[1:]):   # time-weighted mean of the running count in [a, b)
            tt = np.clip(ev[:, 0], a, b)
            dt = np.diff(np.concatenate([tt, [b]]))
            tenths.append(float((run * dt).sum() / (b - a)))
        below = np.nonzero(run >= 0.5 * peak)[0]
        tail = span - ev[below[-1], 0] if len(below) else 0.0
        first_gen = np.sort(st)[: int(peak)]
        rec = {"label": label, "size": sz, "lights": n, "event_ms": round(e0.elapsed_time(e1), 4), "span_us": round(float(span), 1),
               "items": int(len(it)), "rows_per_item": sorted(set(int(r) for r in rows)), "peak_running": int(peak),
               "slots_filled_at_us": round(float(first_gen[-1]), 1), "running_by_tenth_of_span": [round(v) for v in tenths],
               "item_us_mean": round(float((en - st).mean()), 2), "item_us_p99": round(float(np.percentile(en - st, 99)), 2),
               "staging_us_mean": round(float((lg - st).mean()), 2), "tail_below_half_peak_us": round(float(tail), 1),
               "busy_item_us_over_span_x_peak": round(float((en - st).sum() / (span * peak)), 3)}
        if os.environ.get("PBR_TIMELINE_MAP"):   # where the long blocks are: mean block time (us) over a coarse grid of the frame's long zone
            cols = (W + 255) // 256
            dur = (it[:, 2] - it[:, 0]) * 0.01
            rws = it[:, 3] >> 48
            bid = it[:, 3] & 0xFFFFFFFF
            big = rws == rws.max()
            by, cx = bid[big] // cols, bid[big] % cols
            nby = int(by.max()) + 1
            grid = np.zeros((8, cols)); cnt = np.zeros((8, cols))
            np.add.at(grid, (by * 8 // nby, cx), dur[big]); np.add.at(cnt, (by * 8 // nby, cx), 1)
            rec["mean_block_us_by_eighth_of_height_and_column"] = [[round(float(v), 1) for v in row] for row in grid / np.maximum(cnt, 1)]
            rec["start_us_by_eighth_of_height"] = [round(float(st[big][(by * 8 // nby) == k].mean()), 1) for k in range(8)]
        bl = blocks[blocks[:, 0] > 0].astype(np.int64)
        if len(bl):   # persistent blocks: prologue, life, items per block, XCD of each block
            rec.update({"blocks": int(len(bl)), "prologue_us_mean": round(float(((bl[:, 1] - bl[:, 0]) * 0.01).mean()), 2),
                        "block_first_start_to_last_start_us": round(float((bl[:, 0].max() - bl[:, 0].min()) * 0.01), 1),
                        "block_end_spread_us": round(float((bl[:, 2].max() - bl[:, 2].min()) * 0.01), 1),
                        "items_per_block_min_max": [int((bl[:, 3] >> 32).min()), int((bl[:, 3] >> 32).max())],
                        "blocks_per_xcd": np.bincount((bl[:, 3] & 15).astype(np.int64), minlength=8).tolist(),
                        "xcd_equals_block_mod_8": bool(((bl[:, 3] & 15) == (np.nonzero(blocks[:, 0] > 0)[0] & 7)).all())})
        print(json.dumps(rec), flush=True)
        del fr
    del gb
    torch.cuda.empty_cache()
