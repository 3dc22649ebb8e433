"""ctypes / numpy mirrors of the POD structs in include/pbr_hip.h.

Each struct cites the reference type it mirrors (paths relative to /root/reference).
"""
import ctypes as C

import numpy as np

CLUSTER_X, CLUSTER_Y, CLUSTER_Z = 24, 16, 8          # DeferredRendering/Shader/clustered.hlsli:10-12
MAX_LIGHTS_PER_CLUSTER = 32                           # clustered.hlsli:9
MAX_SCENE_LIGHTS = 1024                               # Engine/Include/Renderer/Pipeline/DeferredPipeline.h:329
NUM_CLUSTERS = CLUSTER_X * CLUSTER_Y * CLUSTER_Z
HISTOGRAM_BINS = 256                                  # DeferredPipeline.h:409
ENV_MIPS = 5                                          # global.hlsli:9
BLOOM_MIPS = 5                                        # DeferredPipeline.h:212
# AutoExposurePass constants, DeferredPipeline.h:404-407
MIN_LOG_LUMINANCE = -10.0
MAX_LOG_LUMINANCE = 2.0
LOG_LUMINANCE_RANGE = MAX_LOG_LUMINANCE - MIN_LOG_LUMINANCE
INV_LOG_LUMINANCE_RANGE = float(np.float32(1.0) / np.float32(LOG_LUMINANCE_RANGE))
# BloomPass prefilter constants, DeferredPipeline.cpp:419-420
BLOOM_THRESHOLD = 1.0
BLOOM_KNEE = 0.5


class ShPack(C.Structure):
    """SH2CoefficientsPack, Engine/Include/Utils/SH.h:20-29."""
    _fields_ = [(n, C.c_float * 4) for n in ("sha_r", "shb_r", "sha_g", "shb_g", "sha_b", "shb_b", "shc")]


class Global(C.Structure):
    """ConstantBufferGlobal, Engine/Include/Renderer/Pipeline/IPipeline.h:38-62."""
    _fields_ = [
        ("SkyBoxSH", ShPack),
        ("InvView", C.c_float * 16),
        ("View", C.c_float * 16),
        ("Projection", C.c_float * 16),
        ("InvProjection", C.c_float * 16),
        ("CameraPos", C.c_float * 3),
        ("Ratio", C.c_float),
        ("Resolution", C.c_float * 2),
        ("Near", C.c_float),
        ("Far", C.c_float),
        ("Fov", C.c_float),
        ("DeltaTime", C.c_float),
        ("Time", C.c_float),
    ]


assert C.sizeof(Global) == 412


class Tile(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("x0", "y0", "w", "h", "full_w", "full_h")]


class GBuffer(C.Structure):
    """G-buffer planes (gbuffer.hlsl:10-26,144-146; formats DeferredPipeline.h:107-110)."""
    _fields_ = [
        ("A", C.c_void_p),
        ("B", C.c_void_p),
        ("C", C.c_void_p),
        ("depth", C.c_void_p),
        ("stencil", C.c_void_p),
        ("pitch", C.c_uint32),
    ]


class HaloPeer(C.Structure):
    """pbr_halo_peer: rectangles {x, y, w, h} of the level-1 plane exchanged with rank `rank`."""
    _fields_ = [("rank", C.c_int32), ("send", C.c_uint32 * 4), ("recv", C.c_uint32 * 4)]


class CubeF32(C.Structure):
    _fields_ = [("data", C.c_void_p), ("size", C.c_uint32), ("mips", C.c_uint32)]


# PointLight, DeferredPipeline.h:341-347 (44 B)
LIGHT_DTYPE = np.dtype([
    ("Position", np.float32, 3), ("Color", np.float32, 3), ("Intensity", np.float32),
    ("Radius", np.float32), ("C0", np.float32), ("C1", np.float32), ("C2", np.float32),
])
assert LIGHT_DTYPE.itemsize == 44

# Cluster, DeferredPipeline.h:333-339 (156 B)
CLUSTER_DTYPE = np.dtype([
    ("MinBound", np.float32, 3), ("MaxBound", np.float32, 3), ("NumLights", np.int32),
    ("LightIndex", np.int32, MAX_LIGHTS_PER_CLUSTER),
])
assert CLUSTER_DTYPE.itemsize == 156


def cube_mip_offset(size: int, mip: int) -> int:
    """Texel offset of mip `mip` in a cube chain (mips concatenated, 6 faces per mip)."""
    return sum(6 * (size >> m) ** 2 for m in range(mip))


def cube_texels(size: int, mips: int) -> int:
    return cube_mip_offset(size, mips)


def env_padded_mip_offset(size: int, mip: int) -> int:
    """Texel offset of mip `mip` in the footprint layout of pbr_env_pad (4 texels per bilinear footprint origin,
    (s+1)^2 origins per face)."""
    return sum(6 * ((size >> m) + 1) ** 2 * 4 for m in range(mip))


def env_padded_texels(size: int, mips: int) -> int:
    return env_padded_mip_offset(size, mips)


def bloom_level_offset(w: int, h: int, level: int) -> int:
    return sum((w >> l) * (h >> l) for l in range(level))


def bloom_chain_texels(w: int, h: int) -> int:
    return bloom_level_offset(w, h, BLOOM_MIPS)
