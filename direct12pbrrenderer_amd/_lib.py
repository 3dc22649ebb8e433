"""Loader for the C-ABI shared library (include/pbr_hip.h).

The HIP library is the product: there is no CPU fallback.  If libpbr_hip.so is missing or a
symbol cannot be resolved this module raises — it never substitutes another implementation.
"""
import ctypes as C
import os

from .structs import CubeF32, GBuffer, Global, HaloPeer, Tile

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PBR_HIP_LIB", os.path.join(_HERE, "libpbr_hip.so"))   # override = experiments only

_u32, _f32, _int, _vp, _sz = C.c_uint32, C.c_float, C.c_int, C.c_void_p, C.c_size_t

# name -> (restype, argtypes); exactly the entry points include/pbr_hip.h declares
SIGNATURES = {
    "pbr_version": (C.c_char_p, []),
    "pbr_cube_texels": (_sz, [_u32, _u32]),
    "pbr_cube_mip_offset": (_sz, [_u32, _u32]),
    "pbr_env_padded_texels": (_sz, [_u32, _u32]),
    "pbr_env_padded_mip_offset": (_sz, [_u32, _u32]),
    "pbr_bloom_chain_texels": (_sz, [_u32, _u32]),
    "pbr_bloom_level_offset": (_sz, [_u32, _u32, _u32]),
    "pbr_ctx_create": (_int, [_int, C.POINTER(_vp)]),
    "pbr_ctx_destroy": (None, [_vp]),
    "pbr_ctx_set_stream": (_int, [_vp, _vp]),
    "pbr_ctx_use_own_stream": (_int, [_vp]),
    "pbr_ctx_get_stream": (_vp, [_vp]),
    "pbr_ctx_side_begin": (_int, [_vp]),
    "pbr_ctx_side_end": (_int, [_vp]),
    "pbr_ctx_side_join": (_int, [_vp]),
    "pbr_last_error": (C.c_char_p, [_vp]),
    "pbr_sync": (_int, [_vp]),
    "pbr_brdf_lut": (_int, [_vp, _u32, _vp]),
    "pbr_rgbe_decode": (_int, [_vp, _vp, C.c_size_t, _vp]),
    "pbr_cube_gen_mips": (_int, [_vp, _vp, _u32, _u32]),
    "pbr_prefilter_env": (_int, [_vp, C.POINTER(CubeF32), _u32, _u32, _vp]),
    "pbr_prefilter_env_mip": (_int, [_vp, C.POINTER(CubeF32), _u32, _u32, _f32, _vp]),
    "pbr_env_pad": (_int, [_vp, _vp, _u32, _u32, _vp]),
    "pbr_sh9_project": (_int, [_vp, C.POINTER(CubeF32), _vp]),
    "pbr_cluster_build": (_int, [_vp, C.POINTER(Global), _vp]),
    "pbr_cluster_cull": (_int, [_vp, C.POINTER(Global), _vp, _int, _vp]),
    "pbr_clustered": (_int, [_vp, C.POINTER(Global), _vp, _int, _vp]),
    "pbr_deferred_shade": (_int, [_vp, C.POINTER(Global), C.POINTER(Tile), C.POINTER(GBuffer),
                                  _vp, _u32, _vp, _u32, _u32, _vp, _vp, _int, _vp, _u32]),
    "pbr_deferred_shade_rects": (_int, [_vp, C.POINTER(Global), C.POINTER(Tile), C.POINTER(GBuffer),
                                        _vp, _u32, _vp, _u32, _u32, _vp, _vp, _int, _vp, _u32, _vp, _u32]),
    "pbr_deferred_shade_f32": (_int, [_vp, C.POINTER(Global), C.POINTER(Tile), C.POINTER(GBuffer),
                                      _vp, _u32, _vp, _u32, _u32, _vp, _vp, _int, _vp, _u32]),
    "pbr_skybox": (_int, [_vp, C.POINTER(Global), C.POINTER(Tile), C.POINTER(CubeF32), _vp, _u32, _vp, _u32]),
    "pbr_gbuffer_encode": (_int, [_vp, _vp, _vp, _vp, _u32, _u32, _u32, _vp, _vp, _vp]),
    "pbr_bloom_prefilter": (_int, [_vp, _vp, _u32, _u32, _u32, _vp, _f32, _f32]),
    "pbr_blur_h": (_int, [_vp, _vp, _u32, _u32, _vp, _u32, _u32]),
    "pbr_blur_v": (_int, [_vp, _vp, _u32, _u32, _vp, _u32, _u32]),
    "pbr_bloom_upsample_add": (_int, [_vp, _vp, _u32, _u32, _vp, _u32, _u32, _vp]),
    "pbr_bloom_up_level": (_int, [_vp, _vp, _vp, _u32, _u32, _vp, _u32, _u32]),
    "pbr_bloom_merge": (_int, [_vp, _vp, _u32, _vp, _u32, _u32]),
    "pbr_bloom": (_int, [_vp, _vp, _u32, _u32, _u32, _vp, _vp, _f32, _f32]),
    "pbr_bloom_prefilter_rect": (_int, [_vp, _vp, _u32, _u32, _u32, _vp, _u32, _u32, _u32, C.POINTER(_u32 * 4), _f32, _f32]),
    "pbr_bloom_prefilter_rects": (_int, [_vp, _vp, _u32, _u32, _u32, _vp, _u32, _u32, _u32, _vp, _u32, _f32, _f32]),
    "pbr_bloom_tiled": (_int, [_vp, _vp, _u32, C.POINTER(_u32 * 4), _u32, _u32, _vp, _vp, C.POINTER(_u32 * 4), _f32, _f32, _vp]),
    "pbr_bloom_histogram": (_int, [_vp, _vp, _u32, _u32, _u32, _vp, _vp, _f32, _f32, C.POINTER(_u32 * 4), _f32, _f32, _vp]),
    "pbr_lum_histogram": (_int, [_vp, _vp, _u32, _u32, _u32, _f32, _f32, _vp]),
    "pbr_lum_average": (_int, [_vp, _vp, _u32, _f32, _f32, _f32, _vp]),
    "pbr_tonemap": (_int, [_vp, _vp, _u32, _u32, _u32, _vp, _vp, _u32]),
    "pbr_average_tonemap": (_int, [_vp, _vp, _u32, _f32, _f32, _f32, _vp, _vp, _vp, _vp, _u32, _u32, _u32, _vp, _u32]),
    "pbr_comm_unique_id": (_int, [_vp]),
    "pbr_comm_init": (_int, [_vp, _int, _int, _vp]),
    "pbr_allreduce_hist": (_int, [_vp, _vp]),
    "pbr_halo_staging_bytes": (_sz, [C.POINTER(HaloPeer), _u32]),
    "pbr_halo_exchange": (_int, [_vp, _vp, _u32, _u32, C.POINTER(HaloPeer), _u32, _vp, _sz]),
    "pbr_halo_pack": (_int, [_vp, _vp, _u32, _u32, C.POINTER(HaloPeer), _u32, _vp, _sz, _int]),
    "pbr_runtime_error": (C.c_char_p, []),
    "pbr_runtime_mismatch_dirs": (_int, [C.c_char_p, C.c_char_p]),
    "pbr_membench_read": (_int, [_vp, _vp, _sz, _vp, _u32]),
    "pbr_valubench": (_int, [_vp, _u32, _u32, _u32, _vp]),
    "pbr_ctx_set_bloom_shader_order": (_int, [_vp, _int]),
}

# measurement entry points of the knobs build (PBR_HIP_LIB=.../libpbr_hip_knobs.so); the product library does not export them
KNOBS_ONLY = {
    "pbr_ctx_set_cu_masks": (_int, [_vp, _vp, _vp, _u32]),
}

_lib = None


def load():
    """dlopen libpbr_hip.so and bind every symbol; raises if the library or a symbol is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build it with `make -C direct12pbrrenderer_amd/csrc` "
            "(or __graft_entry__.build()).  There is no CPU fallback for the HIP path.")
    # torch first: it brings its own copy of the HIP runtime, and a process that has already loaded the system one
    # (through this library) cannot create streams on torch's device afterwards (hipStreamCreate fails)
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError if the export is missing
        fn.restype = res
        fn.argtypes = args
    for name, (res, args) in KNOBS_ONLY.items():   # present in libpbr_hip_knobs.so only (include/pbr_hip.h, last section)
        fn = getattr(lib, name, None)
        if fn is not None:
            fn.restype = res
            fn.argtypes = args
    _lib = lib
    return lib
