"""How much of the shade's light walk is lost (a) to lanes waiting for the longest list of their wave and (b) to lights that
face away from the pixel, and what regrouping could return: CPU only (oracle cluster cull + the synthetic G-buffer of the
bench), a band of the 4K / 256-light frame.
    python tools/shade_divergence.py [rows]
A lane walks ceil(n / 2) pairs of its pixel's cluster list; a wave = 64 consecutive pixels of a row (k_deferred_shade) runs
the max over its lanes.  A light with N.(Lpos - P) <= 0 multiplies its whole term by max(N.L, 0) = 0 (brdf.hlsli:51,
deferred_shading.hlsl:186): walking it accumulates exact zeros.  Printed: listed and front-facing lights per pixel; pairs per
wave as shipped, with every lane skipping its own back-facing lights, and after sorting groups of G pixels (a block's rows) by
the number of pairs left — what a permuted lane -> pixel assignment inside a block could reach, per group size."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from direct12pbrrenderer_amd import scene, synth  # noqa: E402
from direct12pbrrenderer_amd.structs import CLUSTER_DTYPE  # noqa: E402
from oracle import binding as orc  # noqa: E402  (measurement tool: the oracle's cluster cull supplies the lists)

W, H = 3840, 2160
ROWS = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ROUGH_MIN = int(os.environ.get("ROUGH_MIN", "48"))
y0 = (H - ROWS) // 2
cam = scene.Camera.reference_default(W, H)
g = scene.make_global(cam, W, H)
lights = synth.lights_in_view_box(bench.N_LIGHTS, cam)
cl = orc.cluster_build(g)
orc.cluster_cull(g, lights, cl)
gb = synth.gbuffer_tile(0, y0, W, ROWS, W, H, rough_min=ROUGH_MIN)
CX, CY, CZ = bench.CLUSTER_X, bench.CLUSTER_Y, bench.CLUSTER_Z
near, far = float(g.Near), float(g.Far)
z = near * far / (far - gb["depth"].astype(np.float64) * (far - near))
sz = np.clip((CZ * np.log(np.clip(z, near, far) / near) / np.log(far / near)).astype(np.int64), 0, CZ - 1)
sx = np.clip(np.floor((np.arange(W) + 0.5) / W * CX).astype(np.int64), 0, CX - 1)[None, :]
sy = np.clip(np.floor((1 - (np.arange(y0, y0 + ROWS) + 0.5) / H) * CY).astype(np.int64), 0, CY - 1)[:, None]
clv = np.asarray(cl).view(CLUSTER_DTYPE).reshape(-1)
cidx = sz + sx * CZ + sy * CX * CZ
n = np.minimum(clv["NumLights"], 32)[cidx]
covered = gb["stencil"] != 0

# world position and normal of every pixel (deferred_shading.hlsl:79-83,91-121; global.hlsli:101-115), float64
u = (np.arange(W) + 0.5) / W
v = (np.arange(y0, y0 + ROWS) + 0.5) / H
nh = 2.0 * near * np.tan(float(g.Fov) / 2.0)
nw = nh * float(g.Ratio)
cvv = np.stack(np.broadcast_arrays(((2 * u - 1) * 0.5 * nw)[None, :], ((1 - 2 * v) * 0.5 * nh)[:, None], np.full((1, 1), near)), axis=-1)
inv_view = np.array(list(g.InvView), dtype=np.float64).reshape(4, 4)[:3, :3]
pos = np.array(list(g.CameraPos), dtype=np.float64) + (cvv @ inv_view.T) * (z / near)[..., None]
bu, bv = (gb["B"] & 255) / 255.0, ((gb["B"] >> 8) & 255) / 255.0
d = np.stack([2 * bu - 1, 2 * bv - 1, np.zeros_like(bu)], axis=-1)
d[..., 2] = 1 - np.abs(d[..., 0]) - np.abs(d[..., 1])
fold = d[..., 2] < 0
sgn = lambda a: np.where(a < 0, -1.0, 1.0)   # noqa: E731  (Q22: sign(0) = +1)
fx, fy = sgn(d[..., 0]) * (1 - np.abs(d[..., 1])), sgn(d[..., 1]) * (1 - np.abs(d[..., 0]))
d[..., 0], d[..., 1] = np.where(fold, fx, d[..., 0]), np.where(fold, fy, d[..., 1])
nrm = d / np.linalg.norm(d, axis=-1, keepdims=True)

# front-facing entries of every pixel's list: slot j counts when j < n and N.(L_j - P) > 0
lpos = lights["Position"].astype(np.float64)
idx = np.clip(clv["LightIndex"][cidx], 0, len(lights) - 1)                       # [ROWS, W, 32]
front = np.zeros((ROWS, W), dtype=np.int64)
for j in range(32):
    dn = np.einsum("ijk,ijk->ij", lpos[idx[..., j]] - pos, nrm)
    front += (dn > 0) & (j < n)
trips = np.maximum((n + 1) // 2, 1) * covered
ftrips = np.maximum((front + 1) // 2, 1) * covered     # (the kernel's walk is a do-while: one pair at least)


def per_wave(t, group):
    """mean pairs per wave when groups of `group` consecutive pixels of a block (256 wide, group / 256 rows) are sorted"""
    rows = max(group // 256, 1)
    if ROWS % rows:
        return float("nan")
    a = t.reshape(ROWS // rows, rows, W // 256, 256).transpose(0, 2, 1, 3).reshape(ROWS // rows, W // 256, rows * 256)
    if group < 256:
        a = a.reshape(-1, 256 // group, group)
    return np.sort(a, axis=-1).reshape(-1, 64).max(axis=1).mean()


ideal, fideal = trips.sum() / covered.sum(), ftrips.sum() / covered.sum()
print(f"{ROWS} rows of the 4K bench frame, roughness >= {ROUGH_MIN}/255")
print(f"lights per pixel: listed {n.mean():.2f}, front-facing {front.mean():.2f}; pairs per pixel: listed {ideal:.2f}, front-facing {fideal:.2f}")
print(f"pixels with every listed light in front: {np.mean((front == n) & (n > 0)):.1%}; with none: {np.mean((front == 0) & (n > 0)):.1%}")
print(f"pairs per wave as shipped (64 x 1 pixels, whole lists):            {per_wave(trips, 64):.2f}")
print(f"pairs per wave, every lane skips its own back-facing lights:      {per_wave(ftrips, 64):.2f}")
for grp in (256, 512, 1024, 2048):
    print(f"pairs per wave, {grp:4d} pixels ({max(grp // 256, 1)} block rows) sorted by listed pairs: {per_wave(trips, grp):.2f}   by front-facing pairs: {per_wave(ftrips, grp):.2f}")
hist = np.bincount(ftrips[covered].ravel(), minlength=17)
print("front-facing pairs per pixel, share of the pixels: " + " ".join(f"{k}:{c / covered.sum():.3f}" for k, c in enumerate(hist) if c))


def per_wave_cols(t, rows, cols=64):
    """the kernel's own group: the cols x rows pixels one wave of k_deferred_shade_ff owns, sorted, 64 at a time"""
    a = t.reshape(ROWS // rows, rows, W // cols, cols).transpose(0, 2, 1, 3).reshape(-1, rows * cols)
    return np.sort(a, axis=-1).reshape(-1, 64).max(axis=1).mean()


for rows in (1, 2, 4, 8, 16):
    if ROWS % rows == 0:
        print(f"pairs per wave-iteration, a wave's own 64 x {rows:2d} pixels sorted by front-facing pairs (k_deferred_shade_ff): {per_wave_cols(ftrips, rows):.2f}")
