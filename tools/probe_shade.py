import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from direct12pbrrenderer_amd import scene, synth
from direct12pbrrenderer_amd.api import PbrContext
from direct12pbrrenderer_amd.pipeline import DeferredFrame, tile_for_rank
import bench
ctx = PbrContext(0)
lut, env, sh = bench.build_ibl(ctx)
W, H = 3840, 2160
spec = tile_for_rank(0, 1, W, H)
cam = scene.Camera.reference_default(W, H)
g = scene.make_global(cam, W, H, sh_pack=sh)
gb = synth.gbuffer_tile(0, 0, W, H, W, H)
for n in (0, 1, 64, 256):
    lights = synth.lights_in_view_box(n, cam) if n else synth.lights_in_view_box(1, cam)[:0]
    fr = DeferredFrame(ctx, spec, g, lights, lut, 512, env, 512, 5)
    fr.upload_gbuffer(gb)
    fr.clustered() if n else ctx.cluster_build(g, fr.clusters)
    cl = np.frombuffer(fr.clusters.cpu().numpy().tobytes(), dtype=bench.__dict__.get('CLUSTER_DTYPE', None) or __import__('direct12pbrrenderer_amd.structs', fromlist=['x']).CLUSTER_DTYPE)
    ms = bench.time_stage(fr.shade, 20)
    print(f"lights {n}: shade {ms:.4f} ms, mean NumLights/cluster {cl['NumLights'].mean():.2f}, max {cl['NumLights'].max()}", flush=True)
