"""Thin Python host wrapper over the C ABI (include/pbr_hip.h).

PyTorch is used only as plumbing: device allocations (torch tensors), the current HIP stream
and torch.distributed.  Every method below is one C-ABI call on device pointers; errors raise
RuntimeError with pbr_last_error(), mirroring the reference's throw-on-failure
(ThrowIfFailed, Engine/Include/Renderer/Device/Direct12/D3DUtils.h:12-41).
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .structs import (BLOOM_KNEE, BLOOM_THRESHOLD, CLUSTER_DTYPE, ENV_MIPS, HISTOGRAM_BINS,
                      INV_LOG_LUMINANCE_RANGE, LIGHT_DTYPE, LOG_LUMINANCE_RANGE, MIN_LOG_LUMINANCE,
                      NUM_CLUSTERS, CubeF32, GBuffer, Global, HaloPeer, Tile, bloom_chain_texels, cube_texels, env_padded_texels)


class PbrError(RuntimeError):
    pass


def _ptr(t):
    if t is None:
        return None
    if isinstance(t, torch.Tensor):
        if not t.is_cuda:
            raise PbrError("expected a device tensor")
        if not t.is_contiguous():
            raise PbrError("expected a contiguous tensor")
        return C.c_void_p(t.data_ptr())
    return C.c_void_p(int(t))


class PbrContext:
    """One context per device per host thread (pbr_ctx is not thread-safe)."""

    def __init__(self, device=0, use_torch_stream=True):
        self.lib = _lib.load()
        self.device = int(device)
        h = C.c_void_p()
        st = self.lib.pbr_ctx_create(self.device, C.byref(h))
        if st != 0:
            raise PbrError(f"pbr_ctx_create({device}) failed: status {st}")
        self.h = h
        self.torch_device = torch.device("cuda", self.device)
        if use_torch_stream:
            self.bind_torch_stream()

    def bind_torch_stream(self):
        """Enqueue on torch's current stream so torch allocations/events order with our kernels."""
        s = torch.cuda.current_stream(self.torch_device)
        self._check(self.lib.pbr_ctx_set_stream(self.h, C.c_void_p(s.cuda_stream)))

    def close(self):
        if getattr(self, "h", None):
            self.lib.pbr_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, st):
        if st != 0:
            msg = self.lib.pbr_last_error(self.h)
            raise PbrError(f"status {st}: {msg.decode() if msg else '?'}")

    def side_begin(self):
        """Calls until side_end() enqueue on the context's high-priority side stream (after all earlier work)."""
        self._check(self.lib.pbr_ctx_side_begin(self.h))

    def side_end(self):
        self._check(self.lib.pbr_ctx_side_end(self.h))

    def side_join(self):
        """The context's stream waits for the side stream."""
        self._check(self.lib.pbr_ctx_side_join(self.h))

    def sync(self):
        self._check(self.lib.pbr_sync(self.h))

    def use_own_stream(self):
        """Enqueue on the context's private stream (not torch's): the caller orders torch work with ctx.sync() / torch.cuda.synchronize()."""
        self._check(self.lib.pbr_ctx_use_own_stream(self.h))

    def set_bloom_shader_order(self, on):
        """pbr_ctx_set_bloom_shader_order: the large 2x-up bloom levels in the shader's operation order (bit-exact at any size) instead
        of the polyphase form (<= 1 fp16 ULP per stage)"""
        self._check(self.lib.pbr_ctx_set_bloom_shader_order(self.h, 1 if on else 0))

    def partition_cus(self, side_cus, total_cus=None, layout="low", allow_uneven=False):
        """pbr_ctx_set_cu_masks with `side_cus` compute units, spread evenly over the device, for the side stream and the rest for the
        context's private stream (0: no partition, every CU for both).  The context must be on its private stream (use_own_stream).
        Masked streams synchronise with the legacy null stream (hipExtStreamCreateWithCUMask takes no flags): default-stream torch work
        (.zero_(), event records) or a hipMemset inside a partitioned frame makes the two partitions take turns.
        layout="low" needs side_cus to be a multiple of 32 (the same number of CUs from every XCD): anything else runs at the pace of
        the poorest XCD and is refused — the uneven layouts exist as "strided" / "per_xcd", for the measurement of exactly that;
        allow_uneven=True (tools/cu_partition.py, which measures that effect with layout "low" too) lifts the refusal."""
        if not hasattr(self.lib, "pbr_ctx_set_cu_masks"):
            raise PbrError("partition_cus: pbr_ctx_set_cu_masks is an entry point of the knobs build only (PBR_HIP_LIB=.../libpbr_hip_knobs.so): "
                           "the CU partition is a measurement aid, not part of the product library (round 6)")
        if side_cus and layout == "low" and int(side_cus) % 32 and not allow_uneven:
            raise PbrError(f"partition_cus: layout 'low' takes a multiple of 32 compute units (got {side_cus}): an uneven share per XCD runs at the poorest XCD's pace")
        if not side_cus:
            self._check(self.lib.pbr_ctx_set_cu_masks(self.h, None, None, 0))
            return
        n = int(total_cus or torch.cuda.get_device_properties(self.torch_device).multi_processor_count)
        words = (n + 31) // 32
        side = np.zeros(words, np.uint32)
        main = np.zeros(words, np.uint32)
        # On MI355X the bits of a CU mask run XCD by XCD in groups of four (measured: tools/cu_partition.py), and a kernel's workgroups
        # are dealt to the XCDs in equal shares whatever their CU counts — so both partitions must hold the SAME number of CUs of every
        # XCD, or the XCD with the fewest sets the pace: the side stream gets the low side_cus bits, side_cus a multiple of 32
        # (layout="strided": every (n / side_cus)-th bit instead, for the measurement of exactly that effect).
        if layout == "low":          # the low side_cus bits: even per XCD only for multiples of 32
            picked = set(range(side_cus))
        elif layout == "per_xcd":    # side_cus / 8 CUs of EVERY XCD (bit = 32 * (j // 4) + 4 * xcd + j % 4 for the j-th CU of an XCD)
            k = side_cus // 8
            picked = set(32 * (j // 4) + 4 * x + (j % 4) for x in range(8) for j in range(k))
        else:                        # "strided": every (n / side_cus)-th bit — uneven per XCD, for the measurement of exactly that effect
            picked = set(int(round(i * n / side_cus)) % n for i in range(side_cus))
        for cu in range(n):
            (side if cu in picked else main)[cu // 32] |= np.uint32(1 << (cu % 32))
        self._check(self.lib.pbr_ctx_set_cu_masks(self.h, main.ctypes.data, side.ctypes.data, words))

    # ---- allocation helpers (torch = device memory plumbing) ---------------------------------
    def empty(self, shape, dtype):
        return torch.empty(shape, dtype=dtype, device=self.torch_device)

    def zeros(self, shape, dtype):
        return torch.zeros(shape, dtype=dtype, device=self.torch_device)

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        if arr.dtype in (LIGHT_DTYPE, CLUSTER_DTYPE):
            arr = arr.view(np.uint8)
        if arr.dtype == np.uint32:
            return torch.from_numpy(arr.view(np.int32)).to(self.torch_device)
        if arr.dtype == np.uint16:
            return torch.from_numpy(arr.view(np.int16)).to(self.torch_device)
        return torch.from_numpy(arr).to(self.torch_device)

    # ---- one-shot IBL --------------------------------------------------------------------------
    def brdf_lut(self, res, out=None):
        out = out if out is not None else self.empty((res, res, 2), torch.float16)
        self._check(self.lib.pbr_brdf_lut(self.h, res, _ptr(out)))
        return out

    def cube_gen_mips(self, cube, size, mips):
        self._check(self.lib.pbr_cube_gen_mips(self.h, _ptr(cube), size, mips))
        return cube

    def prefilter_env(self, sky, sky_size, sky_mips, size=512, mips=ENV_MIPS, out=None):
        out = out if out is not None else self.empty((cube_texels(size, mips), 4), torch.float16)
        c = CubeF32(sky.data_ptr(), sky_size, sky_mips)
        self._check(self.lib.pbr_prefilter_env(self.h, C.byref(c), size, mips, _ptr(out)))
        return out

    def prefilter_env_dispatches(self, sky, sky_size, sky_mips, size=512, mips=ENV_MIPS, out=None):
        """The chain as PreFilterEnvMapPass::Execute builds it: ONE pbr_prefilter_env_mip per mip (the shader's sequential
        sum, roughness = mip / (mips - 1), DeferredPipeline.cpp:99) — what the C++ pass graph dispatches."""
        from .structs import cube_mip_offset
        out = out if out is not None else self.empty((cube_texels(size, mips), 4), torch.float16)
        c = CubeF32(sky.data_ptr(), sky_size, sky_mips)
        for m in range(mips):
            dst = out.data_ptr() + 8 * cube_mip_offset(size, m)
            self._check(self.lib.pbr_prefilter_env_mip(self.h, C.byref(c), size, m, float(m) / float(max(mips - 1, 1)), C.c_void_p(dst)))
        return out

    def env_pad(self, env, size, mips=ENV_MIPS, out=None):
        """Padded copy of a prefiltered env chain — the layout deferred_shade samples (one-shot)."""
        out = out if out is not None else self.empty((env_padded_texels(size, mips), 4), torch.float16)
        self._check(self.lib.pbr_env_pad(self.h, _ptr(env), size, mips, _ptr(out)))
        return out

    def sh9_project(self, sky, sky_size, sky_mips=1, out=None):
        out = out if out is not None else self.empty((28,), torch.float32)
        c = CubeF32(sky.data_ptr(), sky_size, sky_mips)
        self._check(self.lib.pbr_sh9_project(self.h, C.byref(c), _ptr(out)))
        return out

    # ---- per-frame -------------------------------------------------------------------------------
    def alloc_clusters(self):
        return self.zeros((NUM_CLUSTERS * CLUSTER_DTYPE.itemsize,), torch.uint8)

    def cluster_build(self, g: Global, clusters):
        self._check(self.lib.pbr_cluster_build(self.h, C.byref(g), _ptr(clusters)))

    def cluster_cull(self, g: Global, lights, n, clusters):
        self._check(self.lib.pbr_cluster_cull(self.h, C.byref(g), _ptr(lights), int(n), _ptr(clusters)))

    def clustered(self, g: Global, lights, n, clusters):
        """cluster_build + cluster_cull in one launch (ClusteredPass::Execute)."""
        self._check(self.lib.pbr_clustered(self.h, C.byref(g), _ptr(lights), int(n), _ptr(clusters)))

    def deferred_shade(self, g: Global, tile: Tile, gb, pitch, lut, lut_res, env, env_size, env_mips,
                       clusters, lights, num_lights, hdr, hdr_pitch):
        """gb: dict with device tensors A,B,C,depth,stencil; env: the PADDED chain from env_pad()."""
        s = GBuffer(gb["A"].data_ptr(), gb["B"].data_ptr(), gb["C"].data_ptr(), gb["depth"].data_ptr(),
                    gb["stencil"].data_ptr(), pitch)
        self._check(self.lib.pbr_deferred_shade(self.h, C.byref(g), C.byref(tile), C.byref(s), _ptr(lut), lut_res,
                                                _ptr(env), env_size, env_mips, _ptr(clusters), _ptr(lights),
                                                int(num_lights), _ptr(hdr), hdr_pitch))

    @staticmethod
    def _rects(rects):
        arr = ((C.c_uint32 * 4) * len(rects))()
        for i, r in enumerate(rects):
            arr[i] = (C.c_uint32 * 4)(*[int(v) for v in r])
        return arr

    def deferred_shade_rects(self, g: Global, tile: Tile, gb, pitch, lut, lut_res, env, env_size, env_mips,
                             clusters, lights, num_lights, hdr, hdr_pitch, rects):
        """deferred_shade on up to 5 rectangles (tile-local x, y, w, h) of the tile in one launch."""
        s = GBuffer(gb["A"].data_ptr(), gb["B"].data_ptr(), gb["C"].data_ptr(), gb["depth"].data_ptr(),
                    gb["stencil"].data_ptr(), pitch)
        arr = self._rects(rects)
        self._check(self.lib.pbr_deferred_shade_rects(self.h, C.byref(g), C.byref(tile), C.byref(s), _ptr(lut), lut_res,
                                                      _ptr(env), env_size, env_mips, _ptr(clusters), _ptr(lights),
                                                      int(num_lights), _ptr(hdr), hdr_pitch, C.cast(arr, C.c_void_p), len(rects)))

    def deferred_shade_f32(self, g: Global, tile: Tile, gb, pitch, lut, lut_res, env, env_size, env_mips,
                           clusters, lights, num_lights, hdr_f32, hdr_pitch):
        """Parity probe: deferred_shade with a float32 [h, w, 4] output (the colour before the fp16 store)."""
        s = GBuffer(gb["A"].data_ptr(), gb["B"].data_ptr(), gb["C"].data_ptr(), gb["depth"].data_ptr(),
                    gb["stencil"].data_ptr(), pitch)
        self._check(self.lib.pbr_deferred_shade_f32(self.h, C.byref(g), C.byref(tile), C.byref(s), _ptr(lut), lut_res,
                                                    _ptr(env), env_size, env_mips, _ptr(clusters), _ptr(lights),
                                                    int(num_lights), _ptr(hdr_f32), hdr_pitch))

    def rgbe_decode(self, rgbe, out):
        """rgbe: uint8 device tensor [..., 4] (Radiance texels); out: float32 [..., 4]."""
        self._check(self.lib.pbr_rgbe_decode(self.h, _ptr(rgbe), rgbe.numel() // 4, _ptr(out)))

    def skybox(self, g: Global, tile: Tile, sky, sky_size, sky_mips, stencil, pitch, hdr, hdr_pitch):
        """skybox.hlsl: sky colour into hdr where stencil == 0 (run before deferred_shade)."""
        c = CubeF32(sky.data_ptr(), sky_size, sky_mips)
        self._check(self.lib.pbr_skybox(self.h, C.byref(g), C.byref(tile), C.byref(c), _ptr(stencil), pitch,
                                        _ptr(hdr), hdr_pitch))

    def gbuffer_encode(self, m0, m1, m2, w, h, pitch, A, B, Cc):
        """gbuffer.hlsl::ps_main on per-pixel material planes (float4 each) -> RGBA8 G-buffer planes."""
        self._check(self.lib.pbr_gbuffer_encode(self.h, _ptr(m0), _ptr(m1), _ptr(m2), w, h, pitch,
                                                _ptr(A), _ptr(B), _ptr(Cc)))

    def bloom_prefilter(self, hdr, w, h, pitch, out, threshold=BLOOM_THRESHOLD, knee=BLOOM_KNEE):
        self._check(self.lib.pbr_bloom_prefilter(self.h, _ptr(hdr), w, h, pitch, _ptr(out), threshold, knee))

    def blur_h(self, src, iw, ih, out, ow, oh):
        self._check(self.lib.pbr_blur_h(self.h, _ptr(src), iw, ih, _ptr(out), ow, oh))

    def blur_v(self, src, iw, ih, out, ow, oh):
        self._check(self.lib.pbr_blur_v(self.h, _ptr(src), iw, ih, _ptr(out), ow, oh))

    def bloom_up_level(self, upper, lower, lw, lh, out, ow, oh):
        """out = V(H(upper) + H(lower at out's size)) — one fused upsample level (upper may be None)"""
        self._check(self.lib.pbr_bloom_up_level(self.h, _ptr(upper) if upper is not None else None, _ptr(lower), lw, lh, _ptr(out), ow, oh))

    def bloom_upsample_add(self, upper, uw, uh, lower, lw, lh, out):
        self._check(self.lib.pbr_bloom_upsample_add(self.h, _ptr(upper), uw, uh, _ptr(lower), lw, lh, _ptr(out)))

    def bloom_merge(self, hdr, pitch, src, w, h):
        self._check(self.lib.pbr_bloom_merge(self.h, _ptr(hdr), pitch, _ptr(src), w, h))

    def alloc_bloom_chain(self, w, h):
        return self.zeros((bloom_chain_texels(w, h), 4), torch.float16)

    def bloom(self, hdr, w, h, pitch, chain_a, chain_b, threshold=BLOOM_THRESHOLD, knee=BLOOM_KNEE):
        self._check(self.lib.pbr_bloom(self.h, _ptr(hdr), w, h, pitch, _ptr(chain_a), _ptr(chain_b), threshold, knee))

    def bloom_histogram(self, hdr, w, h, pitch, chain_a, chain_b, rect, hist, threshold=BLOOM_THRESHOLD, knee=BLOOM_KNEE,
                        min_log=MIN_LOG_LUMINANCE, inv_range=INV_LOG_LUMINANCE_RANGE):
        """bloom + luminance histogram of rect=(x,y,w,h) in one pass (adds into hist)."""
        r = (C.c_uint32 * 4)(*[int(v) for v in rect])
        self._check(self.lib.pbr_bloom_histogram(self.h, _ptr(hdr), w, h, pitch, _ptr(chain_a), _ptr(chain_b), threshold, knee,
                                                 C.byref(r), min_log, inv_range, _ptr(hist)))

    def bloom_prefilter_rect(self, hdr, w, h, pitch, out, out_pitch, out_x, out_y, rect, threshold=BLOOM_THRESHOLD, knee=BLOOM_KNEE):
        """bloom_prefilter on the half-res outputs rect=(x,y,w,h) of the image, stored at (out_x + x, out_y + y) of `out`."""
        r = (C.c_uint32 * 4)(*[int(v) for v in rect])
        self._check(self.lib.pbr_bloom_prefilter_rect(self.h, _ptr(hdr), w, h, pitch, _ptr(out), out_pitch, out_x, out_y,
                                                      C.byref(r), threshold, knee))

    def bloom_prefilter_rects(self, hdr, w, h, pitch, out, out_pitch, out_x, out_y, rects, threshold=BLOOM_THRESHOLD, knee=BLOOM_KNEE):
        arr = self._rects(rects)
        self._check(self.lib.pbr_bloom_prefilter_rects(self.h, _ptr(hdr), w, h, pitch, _ptr(out), out_pitch, out_x, out_y,
                                                       C.cast(arr, C.c_void_p), len(rects), threshold, knee))

    def bloom_tiled(self, hdr, hdr_pitch, hdr_rect, ew, eh, chain_a, chain_b, merge_rect, hist=None,
                    min_log=MIN_LOG_LUMINANCE, inv_range=INV_LOG_LUMINANCE_RANGE):
        """Bloom levels 1..4 on the extended tile (level 1 of chain_a already complete) + merge/histogram of merge_rect."""
        hr = (C.c_uint32 * 4)(*[int(v) for v in hdr_rect])
        mr = (C.c_uint32 * 4)(*[int(v) for v in merge_rect])
        self._check(self.lib.pbr_bloom_tiled(self.h, _ptr(hdr), hdr_pitch, C.byref(hr), ew, eh, _ptr(chain_a), _ptr(chain_b),
                                             C.byref(mr), min_log, inv_range, _ptr(hist)))

    def lum_histogram(self, hdr, w, h, pitch, hist, min_log=MIN_LOG_LUMINANCE, inv_range=INV_LOG_LUMINANCE_RANGE):
        self._check(self.lib.pbr_lum_histogram(self.h, _ptr(hdr), w, h, pitch, min_log, inv_range, _ptr(hist)))

    def lum_average(self, hist, pixel_count, dt, avg, min_log=MIN_LOG_LUMINANCE, log_range=LOG_LUMINANCE_RANGE):
        self._check(self.lib.pbr_lum_average(self.h, _ptr(hist), pixel_count, min_log, log_range, dt, _ptr(avg)))

    def tonemap(self, hdr, w, h, pitch, avg, out, out_pitch):
        self._check(self.lib.pbr_tonemap(self.h, _ptr(hdr), w, h, pitch, _ptr(avg), _ptr(out), out_pitch))

    def average_tonemap(self, hist, pixel_count, dt, avg_in, avg_out, hist_clear, hdr, w, h, pitch, out, out_pitch,
                        min_log=MIN_LOG_LUMINANCE, log_range=LOG_LUMINANCE_RANGE):
        """pbr_lum_average + pbr_tonemap as one launch: avg_out != avg_in, hist_clear != hist (see pbr_hip.h)"""
        self._check(self.lib.pbr_average_tonemap(self.h, _ptr(hist), pixel_count, min_log, log_range, dt, _ptr(avg_in), _ptr(avg_out),
                                                 _ptr(hist_clear) if hist_clear is not None else None, _ptr(hdr), w, h, pitch, _ptr(out), out_pitch))

    def membench_read(self, buf, sink, blocks):
        """One streaming-read pass over `buf` (measurement aid: the device's achievable HBM-read bandwidth)."""
        self._check(self.lib.pbr_membench_read(self.h, _ptr(buf), buf.numel() * buf.element_size(), _ptr(sink), blocks))

    def valubench(self, op, blocks, iters, stamps):
        """pbr_valubench: blocks x 4 waves issue iters x 8 instructions of class op (0 v_mul_f32, 1 v_fma_f32, 2 v_pk_fma_f32,
        3 v_rcp_f32); stamps: int64 [blocks * 4, 4] device tensor = {shader cycles start, end, 100 MHz ticks start, end} per wave"""
        # the kernel writes blocks * 4 * 4 uint64 through the raw pointer: a short or mistyped tensor would be a silent out-of-bounds write
        if not (isinstance(stamps, torch.Tensor) and stamps.is_cuda and stamps.dtype == torch.int64 and stamps.is_contiguous()
                and stamps.numel() >= int(blocks) * 16):
            raise PbrError(f"valubench: stamps must be a contiguous int64 device tensor of at least {int(blocks) * 16} elements")
        self._check(self.lib.pbr_valubench(self.h, int(op), int(blocks), int(iters), _ptr(stamps)))

    # ---- multi-GPU --------------------------------------------------------------------------------
    def comm_init(self, world, rank, unique_id: bytes):
        buf = C.create_string_buffer(unique_id, 128) if unique_id is not None else None
        self._check(self.lib.pbr_comm_init(self.h, world, rank, C.cast(buf, C.c_void_p) if buf else None))

    def allreduce_hist(self, hist):
        self._check(self.lib.pbr_allreduce_hist(self.h, _ptr(hist)))

    @staticmethod
    def halo_peers(plan):
        """plan: [(rank, send_rect, recv_rect)] with rect = (x, y, w, h) in plane texels or None -> pbr_halo_peer array."""
        arr = (HaloPeer * max(len(plan), 1))()
        for i, (rank, send, recv) in enumerate(plan):
            arr[i].rank = int(rank)
            arr[i].send = (C.c_uint32 * 4)(*(send or (0, 0, 0, 0)))
            arr[i].recv = (C.c_uint32 * 4)(*(recv or (0, 0, 0, 0)))
        return arr, len(plan)

    def halo_staging_bytes(self, peers, n):
        return int(self.lib.pbr_halo_staging_bytes(peers, n))

    def halo_exchange(self, plane, pitch, rows, peers, n, staging):
        """RCCL send/recv of the plan's rectangles on the ctx communicator (pbr_comm_init first)."""
        self._check(self.lib.pbr_halo_exchange(self.h, _ptr(plane), pitch, rows, peers, n, _ptr(staging), staging.numel() * staging.element_size()))

    def halo_pack(self, plane, pitch, rows, peers, n, staging, unpack=False):
        self._check(self.lib.pbr_halo_pack(self.h, _ptr(plane), pitch, rows, peers, n, _ptr(staging),
                                           staging.numel() * staging.element_size(), 1 if unpack else 0))


def comm_unique_id() -> bytes:
    lib = _lib.load()
    buf = C.create_string_buffer(128)
    st = lib.pbr_comm_unique_id(C.cast(buf, C.c_void_p))
    if st != 0:
        raise PbrError(f"pbr_comm_unique_id failed: status {st}")
    return buf.raw


HIST_BINS = HISTOGRAM_BINS
