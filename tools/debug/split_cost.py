import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from direct12pbrrenderer_amd import scene, synth
from direct12pbrrenderer_amd.api import PbrContext
from direct12pbrrenderer_amd.pipeline import DeferredFrame, HaloTransport, parse_layout, tile_of_frame, tile_for_rank
ctx = PbrContext(0)
lut, env, sh = bench.build_ibl(ctx)
class No(HaloTransport):
    def __init__(self): self.kind = "none"
    def exchange(self, fr): return
for name, specs, rank in (("cfg5 tile", [tile_of_frame(r, 8, 7680, 4320, layout=parse_layout("2x4"), halo=True) for r in range(8)], 1),
                          ("weak tile", [tile_for_rank(r, 8, 2720, 3056, layout=(4, 2), halo=True) for r in range(8)], 1)):
    spec = specs[rank]
    cam = scene.Camera.reference_default(spec.full_w, spec.full_h)
    g = scene.make_global(cam, spec.full_w, spec.full_h, sh_pack=sh, delta_time=1.0 / 60.0)
    fr = DeferredFrame(ctx, spec, g, synth.lights_in_view_box(256, cam), lut, 512, env, 512, 5, all_specs=specs, rank=rank, halo_transport=No(), overlap=True)
    fr.upload_gbuffer(synth.gbuffer_tile(spec.sx0, spec.sy0, spec.sw, spec.sh, spec.full_w, spec.full_h))
    fr.set_prev_luminance(0.18)
    for _ in range(40): fr.render()
    ring, core, l1r, l1c = fr.split
    t = {}
    t["shade whole"] = bench.time_stage(fr.shade, 30)
    t["shade ring (1 launch)"] = bench.time_stage(lambda: fr.shade_rects(ring), 30)
    t["shade core"] = bench.time_stage(lambda: fr.shade_rects([core]), 30)
    for i, r in enumerate(ring): t[f"  ring rect {r}"] = bench.time_stage(lambda r=r: fr.shade_rects([r]), 30)
    t["prefilter whole"] = bench.time_stage(fr.halo_prefilter, 30)
    t["prefilter ring"] = bench.time_stage(lambda: fr.prefilter_l1_rects(l1r), 30)
    t["prefilter core"] = bench.time_stage(lambda: fr.prefilter_l1_rects([l1c]), 30)
    t["frame overlapped-order"] = bench.time_stage(fr.render, 30)
    fr.split = None
    t["frame plain"] = bench.time_stage(fr.render, 30)
    print(name, spec.sw, spec.sh, {k: round(v, 4) for k, v in t.items()}, flush=True)
