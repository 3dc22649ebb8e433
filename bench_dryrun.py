"""`bench.py --gpus N --dry-run`: the multi-rank launch of bench.py with every device call replaced by a host stand-in.

What it is for: the N = 8 launch has never run on hardware (no 8-GPU node in development; the pool's process guard caps a one-GPU
rehearsal at 5 ranks).  The dry run starts the real rank processes (spawn_ranks / torch.distributed.run), does the real rendezvous
(gloo), the unique-id broadcast, the grid / tile / halo-plan arithmetic of pipeline.py (DeferredFrame itself runs: only its context is
swapped), the candidate fall-back order of run_workload, the verification frames (every pixel counted once by the all-reduce, every
level-1 texel of every rank's extended tile delivered with its sender's checksum), the timed loops with their barriers, the cfg5
sub-record with its single-GPU denominator, the watchdog phases and the JSON assembly — with 8 ranks on CPU.  No HIP call is made and
no kernel runs: `value` is the rate of no-op frames, which the record says (`"dry_run": true`).  The oracle is not involved.

DryContext has PbrContext's method names (direct12pbrrenderer_amd/api.py); tensors live on the CPU (torch.zeros maps zero pages lazily:
an 8K frame's buffers cost nothing until touched).  Kernels that the verification depends on do the minimum that keeps it meaningful:
 * the interior's luminance histogram counts its pixels into bin 1 (so the all-reduced histogram must count the whole frame once),
 * the level-1 prefilter of a rectangle stamps it with rank-dependent texels (so a strip that arrives from the wrong place, or not at
   all, fails the checksum / "unfilled" test of bench.verify_frame),
 * halo exchange and all-reduce move real bytes over gloo, through the plans the product code made.
"""
import ctypes as C
import hashlib

import numpy as np
import torch

from direct12pbrrenderer_amd import _lib
from direct12pbrrenderer_amd.api import PbrContext
from direct12pbrrenderer_amd.structs import CLUSTER_DTYPE, LIGHT_DTYPE, NUM_CLUSTERS, bloom_chain_texels, cube_texels, env_padded_texels

ENV_MIPS = 5


class DryContext:
    """Host stand-in for PbrContext (see the module docstring).  Fault injection (bench.py --dry-fail): a set-up failure of one candidate on
    one rank is raised by bench.build_frame itself; `fault` below makes this rank miscount or hang inside a frame."""

    torch_device = "cpu"

    def __init__(self, dist, rank, world, fault=None):
        """fault: None, "miscount" (this rank's first interior histogram counts one pixel too many: the verification frame of the first
        candidate fails on every rank) or "hang" (this rank never comes back from its first histogram all-reduce: the other ranks are
        released by their watchdog)."""
        self.dist, self.rank, self.world = dist, rank, world
        self.fault = fault
        self.lib = _lib.load()      # host-side helpers of the real library only (pbr_halo_staging_bytes); no context is created
        self.comm_id = None
        self.calls = {}

    def _n(self, name):
        self.calls[name] = self.calls.get(name, 0) + 1

    # ---- plumbing
    def close(self):
        pass

    def sync(self):
        pass

    side_begin = side_end = side_join = bind_torch_stream = use_own_stream = sync

    def empty(self, shape, dtype):
        return torch.zeros(shape, dtype=dtype)

    zeros = empty

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        if arr.dtype in (LIGHT_DTYPE, CLUSTER_DTYPE):
            arr = arr.view(np.uint8)
        if arr.dtype == np.uint32:
            arr = arr.view(np.int32)
        if arr.dtype == np.uint16:
            arr = arr.view(np.int16)
        return torch.from_numpy(arr)

    # ---- one-shot IBL: shapes only
    def brdf_lut(self, res, out=None):
        return self.zeros((res, res, 2), torch.float16)

    def cube_gen_mips(self, cube, size, mips):
        return cube

    def prefilter_env(self, sky, sky_size, sky_mips, size=512, mips=ENV_MIPS, out=None):
        return self.zeros((cube_texels(size, mips), 4), torch.float16)

    def env_pad(self, env, size, mips=ENV_MIPS, out=None):
        return self.zeros((env_padded_texels(size, mips), 4), torch.float16)

    def sh9_project(self, sky, sky_size, sky_mips=1, out=None):
        return self.zeros((28,), torch.float32)

    # ---- per-frame dispatches: no-ops, but for what the verification frame reads
    def alloc_clusters(self):
        return self.zeros((NUM_CLUSTERS * CLUSTER_DTYPE.itemsize,), torch.uint8)

    def alloc_bloom_chain(self, w, h):
        return self.zeros((bloom_chain_texels(w, h), 4), torch.float16)

    def clustered(self, *a):
        self._n("clustered")

    def deferred_shade(self, *a):
        self._n("deferred_shade")

    def deferred_shade_rects(self, *a):
        self._n("deferred_shade_rects")

    def skybox(self, *a):
        self._n("skybox")

    def bloom(self, *a, **k):
        self._n("bloom")

    def bloom_histogram(self, hdr, w, h, pitch, chain_a, chain_b, rect, hist, **k):
        self._n("bloom_histogram")
        hist[1] += int(rect[2]) * int(rect[3])

    def lum_histogram(self, hdr, w, h, pitch, hist, **k):
        self._n("lum_histogram")
        hist[1] += int(w) * int(h)

    def bloom_tiled(self, hdr, hdr_pitch, hdr_rect, ew, eh, chain_a, chain_b, merge_rect, hist=None, **k):
        self._n("bloom_tiled")
        if hist is not None:
            hist[1] += int(merge_rect[2]) * int(merge_rect[3])
            if self.fault == "miscount":
                hist[1] += 1
                self.fault = None

    def _stamp(self, out, out_pitch, out_x, out_y, rect):
        x, y, w, h = (int(v) for v in rect)
        plane = out.view(torch.int16).view(-1, out_pitch, 4)
        yy = torch.arange(y, y + h, dtype=torch.int32).view(h, 1, 1)
        xx = torch.arange(x, x + w, dtype=torch.int32).view(1, w, 1)
        # a texel pattern that depends on the rank and on the position inside the rank's image: never the 777.0 fill of verify_frame
        plane[out_y + y:out_y + y + h, out_x + x:out_x + x + w] = ((yy * 31 + xx * 7 + self.rank * 1009) % 12000 + 1).to(torch.int16).expand(h, w, 4)

    def bloom_prefilter_rect(self, hdr, w, h, pitch, out, out_pitch, out_x, out_y, rect, **k):
        self._n("bloom_prefilter_rect")
        self._stamp(out, out_pitch, out_x, out_y, rect)

    def bloom_prefilter_rects(self, hdr, w, h, pitch, out, out_pitch, out_x, out_y, rects, **k):
        self._n("bloom_prefilter_rects")
        for r in rects:
            self._stamp(out, out_pitch, out_x, out_y, r)

    def lum_average(self, hist, pixel_count, dt, avg, **k):
        self._n("lum_average")
        hist.zero_()          # hdr_average_histogram.hlsl clears the histogram (the next frame accumulates from zero)

    def tonemap(self, *a):
        self._n("tonemap")

    def average_tonemap(self, hist, *a, **k):
        self._n("average_tonemap")
        hist.zero_()

    # ---- multi-GPU: the real plans, bytes over gloo
    def comm_unique_id(self):
        return hashlib.sha512(b"pbr dry run unique id").digest() * 2     # 128 bytes

    def comm_init(self, world, rank, unique_id):
        """Every rank must hold rank 0's 128 bytes (the broadcast is what is being rehearsed)."""
        if unique_id is None or len(unique_id) != 128:
            raise RuntimeError("dry comm_init: no 128-byte unique id")
        ids = [None] * world
        self.dist.all_gather_object(ids, hashlib.sha1(unique_id).hexdigest())
        if len(set(ids)) != 1:
            raise RuntimeError("dry comm_init: the ranks hold different unique ids")
        self.comm_id = unique_id

    def allreduce_hist(self, hist):
        self._n("allreduce_hist")
        if self.fault == "hang":
            import time
            while True:       # a rank that never arrives: only the watchdog's deadline ends this process (and its peers')
                time.sleep(1.0)
        self.dist.all_reduce(hist)

    halo_peers = staticmethod(PbrContext.halo_peers)

    def halo_staging_bytes(self, peers, n):
        return int(self.lib.pbr_halo_staging_bytes(peers, n))

    @staticmethod
    def _peer_rects(peers, n):
        out = []
        for i in range(n):
            s, r = tuple(peers[i].send), tuple(peers[i].recv)
            out.append((int(peers[i].rank), s if s[2] * s[3] else None, r if r[2] * r[3] else None))
        return out

    def halo_pack(self, plane, pitch, rows, peers, n, staging, unpack=False):
        """pbr_halo_pack's staging layout: the send rectangles in plan order, then the receive rectangles in plan order."""
        self._n("halo_unpack" if unpack else "halo_pack")
        p = plane.view(torch.int16).view(rows, pitch, 4)
        st = staging.view(torch.int16).view(-1, 4)
        plan = self._peer_rects(peers, n)
        off = sum(s[2] * s[3] for _, s, _ in plan if s) if unpack else 0
        for _, s, r in plan:
            q = r if unpack else s
            if not q:
                continue
            x, y, w, h = q
            if unpack:
                p[y:y + h, x:x + w] = st[off:off + w * h].view(h, w, 4)
            else:
                st[off:off + w * h] = p[y:y + h, x:x + w].reshape(w * h, 4)
            off += w * h

    def halo_exchange(self, plane, pitch, rows, peers, n, staging):
        """pbr_halo_exchange: pack, one group of send / recv, unpack."""
        self._n("halo_exchange")
        dist = self.dist
        self.halo_pack(plane, pitch, rows, peers, n, staging, unpack=False)
        plan = self._peer_rects(peers, n)
        st = staging.view(torch.int16).view(-1, 4)
        ops, off = [], 0
        for peer, s, _ in plan:
            if s:
                ops.append(dist.P2POp(dist.isend, st[off:off + s[2] * s[3]], peer))
                off += s[2] * s[3]
        for peer, _, r in plan:
            if r:
                ops.append(dist.P2POp(dist.irecv, st[off:off + r[2] * r[3]], peer))
                off += r[2] * r[3]
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        self.halo_pack(plane, pitch, rows, peers, n, staging, unpack=True)
