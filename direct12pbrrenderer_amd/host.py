"""ctypes binding of libpbr_host.so (direct12pbrrenderer_amd/host/pbr_host.h): the C++ pass graph — RenderScheduler ->
FrameGraph -> the reference-shaped pass classes -> HipCommandList -> the C ABI of include/pbr_hip.h.

This is the drop-in side of the boundary: a host program written against the reference's pass API runs the HIP
kernels through it.  Python only loads the library and hands over host arrays; nothing here computes."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpbr_host.so")
_u32, _vp, _int = C.c_uint32, C.c_void_p, C.c_int

SIGNATURES = {
    "pbrh_create": (_vp, [_int, _u32, _u32, _u32, _u32, C.c_char_p, C.c_size_t]),
    "pbrh_create_tile": (_vp, [_int, _u32, _u32, _u32, _u32, _u32, _int, _u32, _u32, C.c_char_p, C.c_size_t]),
    "pbrh_tile_layout": (_int, [_u32, _u32, _u32, _u32, _u32, _int, _vp, _vp, _int]),
    "pbrh_destroy": (None, [_vp]),
    "pbrh_last_error": (C.c_char_p, [_vp]),
    "pbrh_set_skybox": (_int, [_vp, _vp, _u32]),
    "pbrh_load_skybox": (_int, [_vp, C.c_char_p]),
    "pbrh_set_lights": (_int, [_vp, _vp, _int]),
    "pbrh_load_scene_lights": (_int, [_vp, C.c_char_p]),
    "pbrh_parse_scene_lights": (_int, [C.c_char_p, C.c_size_t, _vp, _int, C.c_char_p, C.c_size_t]),
    "pbrh_light_buffer": (_int, [_u32, _u32, _vp, _vp, _int, _vp, _int]),
    "pbrh_set_gbuffer": (_int, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "pbrh_set_materials": (_int, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "pbrh_set_initial_luminance": (_int, [_vp, C.c_float]),
    "pbrh_set_tile": (_int, [_vp] + [_u32] * 8),
    "pbrh_comm_init": (_int, [_vp, _int, _int, _vp]),
    "pbrh_set_halo_loopback": (_int, [_vp, _int]),
    "pbrh_halo_copy_from": (_int, [_vp, _vp]),
    "pbrh_set_frames_in_flight": (_int, [_vp, _int]),
    "pbrh_set_tail_overlap": (_int, [_vp, _int]),
    "pbrh_set_external_histogram": (_int, [_vp, _vp]),
    "pbrh_capture_histogram": (_int, [_vp, _int]),
    "pbrh_captured_histogram": (_int, [_vp, _vp]),
    "pbrh_set_fused": (_int, [_vp, _int]),
    "pbrh_render_n": (_int, [_vp, _int, C.c_float, C.POINTER(C.c_double)]),
    "pbrh_render": (_int, [_vp, C.c_float]),
    "pbrh_execution_order": (_int, [_vp, C.c_char_p, C.c_size_t]),
    "pbrh_dispatch_count": (_int, [_vp]),
    "pbrh_event_log": (_int, [_vp, C.c_char_p, C.c_size_t]),
    "pbrh_read": (C.c_long, [_vp, C.c_char_p, _vp, C.c_size_t]),
    "pbrh_get_global": (_int, [_vp, _vp]),
    "pbrh_cull_lights": (_int, [_u32, _u32, _vp, _vp, _int, _vp, _int]),
    "pbrh_scene_light_bounds": (_int, [_u32, _u32, _vp, C.c_char_p, C.c_size_t, _vp, _int, _vp, _vp, C.c_char_p, C.c_size_t]),
    "pbrh_dry_run_execution_order": (_int, [_u32, _u32, C.c_char_p, C.c_size_t]),
    "pbrh_probe_binding": (_int, [C.c_char_p, _int, C.c_char_p, _int]),
    "pbrh_parse_hdr": (_int, [_vp, C.c_size_t, _vp, _vp, _vp, C.c_size_t, C.c_char_p, C.c_size_t]),
}

_lib = None


def load():
    """dlopen libpbr_host.so (after torch: one HIP runtime per process, see _lib.load) and type every entry point."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} not found: build it with `make -C direct12pbrrenderer_amd/host` (or __graft_entry__.build())")
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class HostError(RuntimeError):
    pass


def pack_lights(lights):
    """structured light array (structs.LIGHT_DTYPE) -> the 8-float records pbrh_set_lights takes: position, colour, radius, intensity."""
    n = len(lights)
    return np.ascontiguousarray(np.concatenate([lights["Position"], lights["Color"], lights["Radius"].reshape(n, 1),
                                                lights["Intensity"].reshape(n, 1)], axis=1).astype(np.float32))


class HostRenderer:
    """One DeferredRenderPipeline + FrameGraph + RenderScheduler (pbrh_renderer).  tile = (full_w, full_h, cols, rows, rank,
    halo) renders one device's tile of a larger frame; otherwise a whole width x height frame."""

    def __init__(self, device, width=None, height=None, env_size=512, lut_res=512, tile=None):
        self.lib = load()
        err = C.create_string_buffer(512)
        if tile is not None:
            fw, fh, cols, rows, rank, halo = tile
            self.h = self.lib.pbrh_create_tile(int(device), fw, fh, cols, rows, rank, 1 if halo else 0, env_size, lut_res, err, 512)
        else:
            self.h = self.lib.pbrh_create(int(device), width, height, env_size, lut_res, err, 512)
        if not self.h:
            raise HostError(f"pbrh_create failed: {err.value.decode()}")

    def _check(self, st):
        if st != 0:
            raise HostError(self.lib.pbrh_last_error(self.h).decode())

    def close(self):
        if getattr(self, "h", None):
            self.lib.pbrh_destroy(self.h)
            self.h = None

    def set_skybox(self, cube_mip0, size):
        a = np.ascontiguousarray(cube_mip0[:4 * 6 * size * size], dtype=np.float32)
        self._check(self.lib.pbrh_set_skybox(self.h, a.ctypes.data, size))

    def set_lights(self, lights):
        p = pack_lights(lights)
        self._check(self.lib.pbrh_set_lights(self.h, p.ctypes.data, len(p)))

    def load_scene_lights(self, path):
        """the mSceneLight records of a reference scene file (Asset/Scene/main.json) become the renderer's lights"""
        self._check(self.lib.pbrh_load_scene_lights(self.h, os.fsencode(path)))

    def set_gbuffer(self, gb):
        planes = [np.ascontiguousarray(gb[k]) for k in ("A", "B", "C", "depth", "stencil")]
        self._check(self.lib.pbrh_set_gbuffer(self.h, *[p.ctypes.data for p in planes]))

    def set_initial_luminance(self, v):
        self._check(self.lib.pbrh_set_initial_luminance(self.h, float(v)))

    def set_fused(self, on):
        self._check(self.lib.pbrh_set_fused(self.h, 1 if on else 0))

    def set_frames_in_flight(self, k):
        self._check(self.lib.pbrh_set_frames_in_flight(self.h, int(k)))

    def set_tail_overlap(self, on):
        """0 / False: off; 1 / True: from the average-luminance dispatch; 2: from the bloom pass (frames without a halo exchange)"""
        self._check(self.lib.pbrh_set_tail_overlap(self.h, int(on)))

    def comm_init(self, world, rank, unique_id):
        buf = C.create_string_buffer(unique_id, 128) if unique_id is not None else None
        self._check(self.lib.pbrh_comm_init(self.h, world, rank, C.cast(buf, C.c_void_p) if buf else None))

    def set_halo_loopback(self, on):
        """halo mode without a communicator: the exchange packs / unpacks the staging area and moves nothing (tests and the one-GPU
        rehearsal of bench.py, which carry the strips — or nothing — themselves)"""
        self._check(self.lib.pbrh_set_halo_loopback(self.h, 1 if on else 0))

    def capture_histogram(self, on):
        self._check(self.lib.pbrh_capture_histogram(self.h, 1 if on else 0))

    def captured_histogram(self):
        """the tile's own 256 luminance counts of the last frame rendered with capture_histogram(True)"""
        h = np.zeros(256, dtype=np.uint32)
        self._check(self.lib.pbrh_captured_histogram(self.h, h.ctypes.data))
        return h

    def set_external_histogram(self, counts256):
        """counts of the OTHER tiles, added before the average (what pbr_allreduce_hist does over RCCL); None: none"""
        if counts256 is None:
            self._check(self.lib.pbrh_set_external_histogram(self.h, None))
        else:
            c = np.ascontiguousarray(counts256, dtype=np.uint32)
            assert c.size == 256
            self._check(self.lib.pbrh_set_external_histogram(self.h, c.ctypes.data))

    def render(self, dt=1.0 / 60.0):
        self._check(self.lib.pbrh_render(self.h, dt))

    def render_n(self, n, dt=1.0 / 60.0):
        """n frames; returns the average wall time per frame in ms (the queue is drained before the clock stops)."""
        ms = C.c_double(0.0)
        self._check(self.lib.pbrh_render_n(self.h, int(n), dt, C.byref(ms)))
        return ms.value

    def dispatch_count(self):
        return int(self.lib.pbrh_dispatch_count(self.h))

    def read(self, name, shape, dtype):
        a = np.zeros(shape, dtype=dtype)
        n = self.lib.pbrh_read(self.h, name.encode(), a.ctypes.data, a.nbytes)
        if n != a.nbytes:
            raise HostError(f"pbrh_read({name}): {n} of {a.nbytes} bytes: {self.lib.pbrh_last_error(self.h).decode()}")
        return a
