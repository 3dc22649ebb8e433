"""Synthetic inputs for the shading path.

Everything comes from a counter-based hash of the GLOBAL pixel index, so any tile of a frame
can be generated on its own rank and is identical to the same region of the whole frame.
numpy only — inputs are generated on the host and uploaded before any timed region.
"""
import numpy as np

from .scene import Camera, make_lights
from .structs import cube_mip_offset

SEED_GBUFFER = 0x5EED0001
SEED_STENCIL = 0x5EED0002
SEED_LIGHTS = 0x5EED0003
SEED_ENV = 0x5EED0004

_M32 = np.uint64(0xFFFFFFFF)


def pcg_hash(v):
    """PCG-RXS-M-XS 32-bit output hash (Jarzynski & Olano 2020) on uint32 arrays."""
    v = np.asarray(v, dtype=np.uint64) & _M32
    state = (v * np.uint64(747796405) + np.uint64(2891336453)) & _M32
    shift = (state >> np.uint64(28)) + np.uint64(4)
    word = (((state >> shift) ^ state) * np.uint64(277803737)) & _M32
    return (((word >> np.uint64(22)) ^ word) & _M32).astype(np.uint32)


def hash_stream(index, seed, k):
    """k-th 32-bit random word of element `index` under `seed`."""
    s = (np.uint64(seed) + np.uint64(k) * np.uint64(0x9E3779B9)) & _M32
    return pcg_hash(pcg_hash(index).astype(np.uint64) ^ s)


def _unit(h):
    """uint32 -> float64 in [0,1)."""
    return h.astype(np.float64) * (1.0 / 4294967296.0)


def gbuffer_tile(x0, y0, w, h, full_w, full_h, near=0.1, far=1000.0, rough_min=48, coverage_mask=False, cell=1, workers=None):
    """gbuffer_band on row bands in a thread pool (numpy releases the GIL inside its loops): the same arrays, several
    times faster for 4K / 8K frames.  workers: None = min(8, cores); 1 = in the calling thread."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    n = min(8, os.cpu_count() or 1) if workers is None else int(workers)
    if n <= 1 or h < 256:
        return gbuffer_band(x0, y0, w, h, full_w, full_h, near, far, rough_min, coverage_mask, cell)
    rows = -(-h // (4 * n))
    bands = [(y, min(rows, h - y)) for y in range(0, h, rows)]
    with ThreadPoolExecutor(n) as ex:
        parts = list(ex.map(lambda b: gbuffer_band(x0, y0 + b[0], w, b[1], full_w, full_h, near, far, rough_min, coverage_mask, cell), bands))
    return {k: np.concatenate([p[k] for p in parts], axis=0) for k in parts[0]}


def gbuffer_band(x0, y0, w, h, full_w, full_h, near=0.1, far=1000.0, rough_min=48, coverage_mask=False, cell=1):
    """Returns dict of A,B,C (uint32 [h,w]), depth (float32 [h,w]), stencil (uint8 [h,w]).

    albedo rgb u8 uniform; emission 255 with p = 1/128; octahedral normal (u8,u8) uniform (every pair
    decodes to a valid direction); roughness u8 uniform in [rough_min, 255]; metallic in {0,255};
    AO uniform (unused by the shade); depth: view-space z log-uniform in [1, 60] mapped to NDC with
    the inverse of ViewSpaceDepth (deferred_shading.hlsl:74-77).

    cell: 1 = the BASELINE workload (every pixel an independent surface sample: the worst case for the IBL gathers and
    for clustered light lists); > 1 = piecewise-constant surfaces of cell x cell pixels (what rendered geometry looks
    like to the caches).

    rough_min: SURVEY 8d asks for roughness uniform on [0,255] AND for radiance that stays below the
    fp16 maximum; with intensity-10 lights the GGX peak 1/(pi a^4) overflows half for roughness below
    ~0.17, so throughput frames clamp the range to [48,255] (tests that want the full range pass 0).
    """
    ys, xs = np.meshgrid(np.arange(y0, y0 + h, dtype=np.uint64), np.arange(x0, x0 + w, dtype=np.uint64), indexing="ij")
    if cell > 1:   # spatially coherent variant: every cell x cell block of pixels shares one surface sample
        idx = (((ys // np.uint64(cell)) * np.uint64(cell)) * np.uint64(full_w) + (xs // np.uint64(cell)) * np.uint64(cell)) & _M32
    else:
        idx = (ys * np.uint64(full_w) + xs) & _M32
    h0 = hash_stream(idx, SEED_GBUFFER, 0)
    h1 = hash_stream(idx, SEED_GBUFFER, 1)
    h2 = hash_stream(idx, SEED_GBUFFER, 2)
    h3 = hash_stream(idx, SEED_GBUFFER, 3)
    emission = np.where((h1 & np.uint32(127)) == 0, np.uint32(255), np.uint32(0))
    A = (h0 & np.uint32(0x00FFFFFF)) | (emission << np.uint32(24))
    B = ((h1 >> np.uint32(8)) & np.uint32(0xFFFF)) | np.uint32(0x00FF0000)          # rg normal, b = 1, a = 0
    span = 256 - int(rough_min)
    rough = (np.uint32(rough_min) + ((h2 & np.uint32(0xFFFF)).astype(np.uint64) * np.uint64(span) >> np.uint64(16)).astype(np.uint32))
    metal = np.where((h2 >> np.uint32(16)) & np.uint32(1), np.uint32(255), np.uint32(0))
    ao = (h2 >> np.uint32(24)) & np.uint32(255)
    Cc = rough | (metal << np.uint32(8)) | (ao << np.uint32(16))
    z_vs = np.exp(_unit(h3) * np.log(60.0))                                           # log-uniform [1,60)
    depth = ((far - near * far / z_vs) / (far - near)).astype(np.float32)
    if coverage_mask:
        cell = ((ys >> np.uint64(4)) * np.uint64((full_w + 15) // 16) + (xs >> np.uint64(4))) & _M32
        stencil = np.where(_unit(hash_stream(cell, SEED_STENCIL, 0)) < 0.10, 0, 1).astype(np.uint8)
    else:
        stencil = np.ones((h, w), dtype=np.uint8)
    return {"A": A.astype(np.uint32), "B": B.astype(np.uint32), "C": Cc.astype(np.uint32), "depth": depth, "stencil": stencil}


SEED_MATERIAL = 0x5EED0020


def material_tile(x0, y0, w, h, full_w, full_h, seed=SEED_MATERIAL):
    """Per-pixel material attributes as the rasterizer + texture fetches hand them to gbuffer.hlsl::ps_main:
    three float32 [h,w,4] planes — m0 = (albedo.rgb in gamma space, emission), m1 = (un-normalised world
    normal, roughness), m2 = (metallic, ambient occlusion, 0, 0).  Values uniform in [0,1] (normal components
    in [-1,1], length in [0.2, 1.7]); one pixel in 64 carries an out-of-range value (< 0 or > 1) in emission /
    roughness / AO to exercise UNORM saturation."""
    ys, xs = np.meshgrid(np.arange(y0, y0 + h, dtype=np.uint64), np.arange(x0, x0 + w, dtype=np.uint64), indexing="ij")
    idx = (ys * np.uint64(full_w) + xs) & _M32
    u = [_unit(hash_stream(idx, seed, k)).astype(np.float32) for k in range(12)]
    m0 = np.stack([u[0], u[1], u[2], u[3]], axis=-1)
    n = np.stack([2 * u[4] - 1, 2 * u[5] - 1, 2 * u[6] - 1], axis=-1)
    ln = np.maximum(np.linalg.norm(n, axis=-1, keepdims=True), 0.2)
    n = np.where(np.linalg.norm(n, axis=-1, keepdims=True) < 0.2, np.float32([0.0, 0.2, 0.0]), n).astype(np.float32)
    m1 = np.concatenate([n, u[7][..., None]], axis=-1)
    m2 = np.stack([u[8], u[9], np.zeros_like(u[8]), np.zeros_like(u[8])], axis=-1)
    wild = (hash_stream(idx, seed, 12) & np.uint32(63)) == 0
    m0[..., 3] = np.where(wild, 3.0 * u[3] - 1.0, m0[..., 3])
    m1[..., 3] = np.where(wild, 3.0 * u[7] - 1.0, m1[..., 3])
    m2[..., 1] = np.where(wild, 3.0 * u[9] - 1.0, m2[..., 1])
    del ln
    return (np.ascontiguousarray(m0, dtype=np.float32), np.ascontiguousarray(m1, dtype=np.float32),
            np.ascontiguousarray(m2, dtype=np.float32))


def lights_in_view_box(n, camera: Camera, seed=SEED_LIGHTS, radius=2.0, intensity=10.0):
    """n lights uniform in the view-space box x[-25,25] y[-8,8] z[1,60], moved to world space."""
    i = np.arange(n, dtype=np.uint64)
    u = [_unit(hash_stream(i, seed, k)) for k in range(6)]
    pv = np.stack([-25.0 + 50.0 * u[0], -8.0 + 16.0 * u[1], 1.0 + 59.0 * u[2], np.ones(n)], axis=1)
    pw = (camera.world_matrix().astype(np.float64) @ pv.T).T[:, :3]
    col = np.stack([u[3], u[4], u[5]], axis=1)
    return make_lights(pw.astype(np.float32), col.astype(np.float32), radius, intensity)


def reference_scene_light():
    """light_1 of Asset/Scene/main.json: position (-4.2,1,3.5), colour (.9,.1,.3), radius 2, intensity 10."""
    return make_lights([[-4.2, 1.0, 3.5]], [[0.9, 0.1, 0.3]], 2.0, 10.0)


def cube_directions(size):
    """Unit direction of every texel centre, [6,size,size,3] (env_map_gen.hlsl:20-44 face convention)."""
    c = (2.0 * (np.arange(size, dtype=np.float64) + 0.5) / size) - 1.0
    v, u = np.meshgrid(c, c, indexing="ij")
    one = np.ones_like(u)
    faces = [(one, -v, -u), (-one, -v, u), (u, one, v), (u, -one, -v), (u, -v, one), (-u, -v, -one)]
    d = np.stack([np.stack(f, axis=-1) for f in faces], axis=0)
    return d / np.linalg.norm(d, axis=-1, keepdims=True)


def env_cube(size, mips=None, seed=SEED_ENV):
    """fp32 RGBA sky: gradient (0.3,0.5,0.9)*(0.5+0.5 d.y) + sun 50*exp(-200(1-d.s)), +-5 % hash noise.

    Returns a flat float32 array in the pbr_cube_f32 layout with room for `mips` levels; only mip 0
    is filled (use pbr_cube_gen_mips / the oracle's box mips for the rest).
    """
    if mips is None:
        mips = int(np.log2(size)) + 1
    d = cube_directions(size)
    s = np.array([1.0, 1.0, 1.0]) / np.sqrt(3.0)
    grad = 0.5 + 0.5 * d[..., 1]
    sun = 50.0 * np.exp(-200.0 * (1.0 - d @ s))
    base = np.array([0.3, 0.5, 0.9])
    rgb = base[None, None, None, :] * grad[..., None] + sun[..., None]
    idx = np.arange(6 * size * size, dtype=np.uint64).reshape(6, size, size)
    noise = 1.0 + 0.05 * (2.0 * _unit(hash_stream(idx, seed, 0)) - 1.0)
    rgb = rgb * noise[..., None]
    out = np.zeros(4 * cube_mip_offset(size, mips), dtype=np.float32)
    m0 = out[: 4 * 6 * size * size].reshape(6, size, size, 4)
    m0[..., :3] = rgb.astype(np.float32)
    m0[..., 3] = 1.0
    return out


def hdr_noise_image(w, h, seed=0x5EED0010, impulse=True):
    """half4 HDR test image: log-uniform noise in [2^-6, 2^3] per channel (+ a few bright impulses)."""
    idx = np.arange(w * h, dtype=np.uint64).reshape(h, w)
    ch = [np.exp2(-6.0 + 9.0 * _unit(hash_stream(idx, seed, k))) for k in range(3)]
    img = np.stack(ch + [np.ones((h, w))], axis=-1)
    if impulse:
        img[h // 3, w // 4, :3] = [400.0, 250.0, 90.0]
        img[(2 * h) // 3, (3 * w) // 4, :3] = [30.0, 60.0, 120.0]
        img[0, 0, :3] = [55.0, 5.0, 5.0]
        img[h - 1, w - 1, :3] = [5.0, 5.0, 70.0]
    return img.astype(np.float16)
