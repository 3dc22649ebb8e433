"""Shade time of one W x H tile (256 lights, bench IBL) for sweeps over the block-schedule knobs PBR_SHADE_BIGFRAC /
PBR_SHADE_ROWS_SMALL (read once per process): python tools/shade_tile_ms.py W H [label]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from direct12pbrrenderer_amd import scene, synth  # noqa: E402
from direct12pbrrenderer_amd.api import PbrContext  # noqa: E402
from direct12pbrrenderer_amd.pipeline import DeferredFrame, TileSpec  # noqa: E402

W, H = int(sys.argv[1]), int(sys.argv[2])
ctx = PbrContext(0)
lut, env, sh = bench.build_ibl(ctx)
FW, FH = 7680, 4320
cam = scene.Camera.reference_default(FW, FH)
g = scene.make_global(cam, FW, FH, sh_pack=sh)
spec = TileSpec(1920, 0, W, H, FW, FH, 0)
fr = DeferredFrame(ctx, spec, g, synth.lights_in_view_box(256, cam), lut, 512, env, 512, 5)
fr.upload_gbuffer(synth.gbuffer_tile(spec.sx0, spec.sy0, spec.sw, spec.sh, FW, FH))
fr.clustered()
for _ in range(100):
    fr.shade()
ms = min(bench.time_stage(fr.shade, 40) for _ in range(3))
print(f"{sys.argv[3] if len(sys.argv) > 3 else ''}: {W}x{H} shade {ms:.4f} ms = {W * H / ms / 1e3:.0f} Mpixel/s", flush=True)
