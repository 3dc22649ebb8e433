"""How much of the shade's light walk is lost to lanes waiting for the longest list of their wave, and what regrouping could
return: CPU only (oracle cluster cull + the synthetic G-buffer of the bench), a 256-row band of the 4K / 256-light frame.
    python tools/shade_divergence.py
A lane walks ceil(n / 2) pairs of its pixel's cluster list; a wave = 64 consecutive pixels of a row (k_deferred_shade) runs
max over its lanes.  Printed: the mean per pixel, the mean per wave as shipped, and per wave after sorting the 256 pixels of a
block row by list length (what a permuted lane -> pixel assignment inside a block could reach)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from direct12pbrrenderer_amd import scene, synth  # noqa: E402
from direct12pbrrenderer_amd.structs import CLUSTER_DTYPE  # noqa: E402
from oracle import binding as orc  # noqa: E402  (measurement tool: the oracle's cluster cull supplies the lists)

W, H, ROWS = 3840, 2160, 256
y0 = (H - ROWS) // 2
cam = scene.Camera.reference_default(W, H)
g = scene.make_global(cam, W, H)
lights = synth.lights_in_view_box(bench.N_LIGHTS, cam)
cl = orc.cluster_build(g)
orc.cluster_cull(g, lights, cl)
gb = synth.gbuffer_tile(0, y0, W, ROWS, W, H)
CX, CY, CZ = bench.CLUSTER_X, bench.CLUSTER_Y, bench.CLUSTER_Z
near, far = float(g.Near), float(g.Far)
z = near * far / (far - gb["depth"].astype(np.float64) * (far - near))
sz = np.clip((CZ * np.log(np.clip(z, near, far) / near) / np.log(far / near)).astype(np.int64), 0, CZ - 1)
sx = np.clip(np.floor((np.arange(W) + 0.5) / W * CX).astype(np.int64), 0, CX - 1)[None, :]
sy = np.clip(np.floor((1 - (np.arange(y0, y0 + ROWS) + 0.5) / H) * CY).astype(np.int64), 0, CY - 1)[:, None]
n = np.minimum(np.asarray(cl).view(CLUSTER_DTYPE)["NumLights"].reshape(-1), 32)[sz + sx * CZ + sy * CX * CZ]
trips = np.maximum((n + 1) // 2, 1) * (gb["stencil"] != 0)
ideal = trips.sum() / 64
shipped = trips.reshape(ROWS, W // 64, 64).max(axis=2)
sorted256 = np.sort(trips.reshape(ROWS, W // 256, 256), axis=2).reshape(ROWS, W // 256, 4, 64).max(axis=3)
print(f"lights per pixel {n.mean():.2f}; pairs per pixel {trips.mean():.2f}; distinct depth slices per wave {np.mean([len(np.unique(r)) for r in sz.reshape(-1, 64)[:20000]]):.2f}")
print(f"pairs per wave as shipped (64 x 1 pixels): {shipped.mean():.2f} = {shipped.sum() / ideal - 1:+.1%} over the per-pixel mean")
print(f"pairs per wave with the 256 pixels of a block row sorted by list length: {sorted256.mean():.2f} = {sorted256.sum() / ideal - 1:+.1%}")
