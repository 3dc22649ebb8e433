"""Per-frame pass sequence of the deferred pipeline for one GPU's tile of a frame.

Order = the order FrameGraph derives from the passes' declared reads/writes
(Engine/Source/Renderer/FrameGraph.cpp:191-250 on Engine/Include/Renderer/Pipeline/DeferredPipeline.h):
Clustered -> DeferredShading -> Bloom -> AutoExposure -> ToneMapping; bloom is applied to the HDR
buffer BEFORE the luminance histogram and the tone-map.

Multi-GPU: a rank owns an interior tile and shades/blooms it plus an apron of
`apron` pixels on every side that has a neighbour, so the interior is what a single GPU would
produce; the only collective is the 256-bin histogram all-reduce between a16 and a17.
"""
from dataclasses import dataclass

import numpy as np
import torch

from .api import PbrContext
from .structs import (ENV_MIPS, HISTOGRAM_BINS, Global, Tile)

# bloom's cumulative support is ~220 full-res pixels (3 down + 3 up levels of a radius-4 kernel on
# a 2x pyramid); 256 also keeps every mip of the extended tile on the full frame's texel grid
# (multiple of 16 = 2^(BLOOM_MIPS-1)).
DEFAULT_APRON = 256
# halo mode: the bloom prefilter of a half-res texel reads full-res pixels 2x-3 .. 2x+2 (five taps one half-res
# texel apart, each a 2x2 quad: bloom_prefilter.hlsl:17-60), so 4 shaded pixels beyond the interior make the
# interior's level-1 texels exact; everything further out comes from the neighbours' level 1 (halo exchange).
HALO_SHADE_APRON = 4


@dataclass
class TileSpec:
    """Interior tile (x0,y0,w,h) of a full_w x full_h frame plus the two rectangles a rank works on:
    E = interior + `apron` on every side that has a neighbour (clipped to the frame): where the bloom pyramid runs;
    S = interior + `shade_apron` (clipped): what is shaded.  apron mode: S == E (shade_apron = None);
    halo mode: shade_apron = HALO_SHADE_APRON and level 1 of the pyramid outside the interior comes from the
    neighbours (SURVEY 8e option 2)."""
    x0: int
    y0: int
    w: int
    h: int
    full_w: int
    full_h: int
    apron: int = 0
    shade_apron: int = None

    # ---- E: the bloom rectangle
    @property
    def ex0(self):
        return max(self.x0 - self.apron, 0)

    @property
    def ey0(self):
        return max(self.y0 - self.apron, 0)

    @property
    def ex1(self):
        return min(self.x0 + self.w + self.apron, self.full_w)

    @property
    def ey1(self):
        return min(self.y0 + self.h + self.apron, self.full_h)

    @property
    def ew(self):
        return self.ex1 - self.ex0

    @property
    def eh(self):
        return self.ey1 - self.ey0

    @property
    def ix(self):   # interior offset inside E
        return self.x0 - self.ex0

    @property
    def iy(self):
        return self.y0 - self.ey0

    # ---- S: the shaded rectangle
    @property
    def halo(self):
        return self.shade_apron is not None and self.apron > 0

    @property
    def _sa(self):
        return self.shade_apron if self.halo else self.apron

    @property
    def sx0(self):
        return max(self.x0 - self._sa, 0)

    @property
    def sy0(self):
        return max(self.y0 - self._sa, 0)

    @property
    def sx1(self):
        return min(self.x0 + self.w + self._sa, self.full_w)

    @property
    def sy1(self):
        return min(self.y0 + self.h + self._sa, self.full_h)

    @property
    def sw(self):
        return self.sx1 - self.sx0

    @property
    def sh(self):
        return self.sy1 - self.sy0

    @property
    def six(self):   # interior offset inside S
        return self.x0 - self.sx0

    @property
    def siy(self):
        return self.y0 - self.sy0


def parse_layout(text):
    """'RxC' (rows x cols, the way BASELINE cfg5 says "tiled 2x4": 2 rows of 4 tiles) -> (cols, rows)."""
    r, c = text.lower().split("x")
    return int(c), int(r)


def grid_for_world(world, layout=None):
    """Tile grid (cols, rows) for `world` ranks.  layout: explicit (cols, rows); default = the most square
    factorisation with cols >= rows (1x1, 2x1, 3x1, 2x2, 3x2, 4x2 ...): the shorter the shared edges, the
    smaller the apron / halo."""
    if world < 1:
        raise ValueError("world size must be >= 1")
    if layout is not None:
        cols, rows = int(layout[0]), int(layout[1])
        if cols < 1 or rows < 1 or cols * rows != world:
            raise ValueError(f"layout {cols}x{rows} (cols x rows) does not hold {world} ranks")
        return cols, rows
    rows = max(r for r in range(1, int(world ** 0.5) + 1) if world % r == 0)
    return world // rows, rows


def _check_tiling(world, tile_w, tile_h, apron, full_w, full_h):
    if world > 1 and (tile_w % 16 or tile_h % 16 or apron % 16):
        # every bloom mip of the extended tile must sit on the full frame's texel grid (2^(BLOOM_MIPS-1) = 16)
        raise ValueError("tile size and apron must be multiples of 16 for multi-GPU tiling")
    if full_w > 65535 or full_h > 65535:
        raise ValueError("the assembled frame exceeds 65535 pixels on a side")


def tile_for_rank(rank, world, tile_w, tile_h, apron=DEFAULT_APRON, layout=None, halo=False):
    """Weak scaling: every rank owns one tile_w x tile_h tile of a (cols*tile_w) x (rows*tile_h) frame."""
    cols, rows = grid_for_world(world, layout)
    cx, cy = rank % cols, rank // cols
    _check_tiling(world, tile_w, tile_h, apron, cols * tile_w, rows * tile_h)
    multi = world > 1
    return TileSpec(cx * tile_w, cy * tile_h, tile_w, tile_h, cols * tile_w, rows * tile_h, apron if multi else 0,
                    HALO_SHADE_APRON if (halo and multi) else None)


def tile_of_frame(rank, world, frame_w, frame_h, apron=DEFAULT_APRON, layout=None, halo=False):
    """Strong scaling: a frame_w x frame_h frame cut into cols x rows equal tiles (BASELINE cfg5: 7680x4320,
    2 rows x 4 cols of 1920x2160)."""
    cols, rows = grid_for_world(world, layout)
    if frame_w % cols or frame_h % rows:
        raise ValueError(f"{frame_w}x{frame_h} does not split into {cols}x{rows} equal tiles")
    return tile_for_rank(rank, world, frame_w // cols, frame_h // rows, apron, (cols, rows), halo)


def halo_plan(rank, world, specs):
    """Level-1 (half-res) strips rank `rank` exchanges with every other rank, in GLOBAL half-res texel
    coordinates.  specs: the TileSpec of every rank.  Rank r needs level 1 on E_r/2; it computes it on its
    interior I_r/2 and receives (E_r/2 intersect I_n/2) from each rank n — which covers E_r/2 exactly, because the
    interiors partition the frame and E is clipped to it.  Returns [(peer, send_rect, recv_rect)] with rect =
    (x0, y0, x1, y1) or None; both sides derive the same rectangles from the same specs, so no sizes travel."""
    def half(x0, y0, x1, y1):
        return (x0 // 2, y0 // 2, x1 // 2, y1 // 2)

    def isect(a, b):
        r = (max(a[0], b[0]), max(a[1], b[1]), min(a[2], b[2]), min(a[3], b[3]))
        return r if r[0] < r[2] and r[1] < r[3] else None

    me = specs[rank]
    e_me = half(me.ex0, me.ey0, me.ex1, me.ey1)
    i_me = half(me.x0, me.y0, me.x0 + me.w, me.y0 + me.h)
    plan = []
    for n in range(world):
        if n == rank:
            continue
        o = specs[n]
        recv = isect(e_me, half(o.x0, o.y0, o.x0 + o.w, o.y0 + o.h))
        send = isect(half(o.ex0, o.ey0, o.ex1, o.ey1), i_me)
        if recv or send:
            plan.append((n, send, recv))
    return plan


class HaloTransport:
    """How the level-1 strips travel between ranks in halo mode.  `exchange(frame)` is enqueued between
    pbr_bloom_prefilter_rect and pbr_bloom_tiled.
      "capi"  : pbr_halo_exchange — RCCL send/recv on the context's own communicator (pbr_comm_init);
      "torch" : pbr_halo_pack + torch.distributed P2P (RCCL with the nccl backend) + unpack;
      "host"  : like "torch" but through host copies — gloo rehearsals with several ranks on one GPU."""

    def __init__(self, kind, dist=None):
        assert kind in ("capi", "torch", "host")
        self.kind, self.dist = kind, dist

    def begin(self, fr):
        """torch transport only: run the exchange on a torch side stream so that it overlaps what the caller enqueues until
        end() (the capi transport is enqueued on the context's own side stream instead: DeferredFrame.shade_and_bloom_overlapped)."""
        assert self.kind == "torch"
        if not fr.halo_n:
            return
        main = torch.cuda.current_stream(fr.ctx.torch_device)
        if getattr(self, "side", None) is None:
            self.side = torch.cuda.Stream(fr.ctx.torch_device)
        self.side.wait_stream(main)
        with torch.cuda.stream(self.side):
            fr.ctx.bind_torch_stream()
            self.exchange(fr)
        fr.ctx.bind_torch_stream()

    def end(self, fr):
        if fr.halo_n:
            torch.cuda.current_stream(fr.ctx.torch_device).wait_stream(self.side)

    def exchange(self, fr):
        ctx, a1, pitch, rows = fr.ctx, fr.level1, fr.spec.ew // 2, fr.spec.eh // 2
        if not fr.halo_n:
            return
        if self.kind == "capi":
            ctx.halo_exchange(a1, pitch, rows, fr.halo_peers, fr.halo_n, fr.halo_staging)
            return
        dist = self.dist
        ctx.halo_pack(a1, pitch, rows, fr.halo_peers, fr.halo_n, fr.halo_staging, unpack=False)
        ops, off, host = [], 0, self.kind == "host"
        st = fr.halo_staging
        if host:
            ctx.sync()
            st = st.cpu()
        for peer, send, _ in fr.halo_plan_local:
            if send:
                n = send[2] * send[3]
                ops.append(dist.P2POp(dist.isend, st[off:off + n], peer))
                off += n
        for peer, _, recv in fr.halo_plan_local:
            if recv:
                n = recv[2] * recv[3]
                ops.append(dist.P2POp(dist.irecv, st[off:off + n], peer))
                off += n
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        if host:
            fr.halo_staging.copy_(st)
        ctx.halo_pack(a1, pitch, rows, fr.halo_peers, fr.halo_n, fr.halo_staging, unpack=True)


class DeferredFrame:
    """Owns the device buffers of one rank and runs the per-frame passes through the C ABI."""

    def __init__(self, ctx: PbrContext, spec: TileSpec, g: Global, lights_np, lut, lut_res, env, env_size,
                 env_mips=ENV_MIPS, allreduce=None, sky=None, all_specs=None, rank=0, halo_transport=None, overlap=False, fused_exposure=False):
        """sky: optional (cube tensor fp32 RGBA with mips, size, mips) — resolved on stencil == 0 pixels
        before the shade like the reference's SkyboxPass; without it those pixels keep what the buffer holds.
        Halo mode (spec.halo): all_specs = the TileSpec of every rank, rank = this one, halo_transport = HaloTransport.
        overlap (halo mode): shade the tile's border RING first, start the exchange of its level-1 strips on a side stream
        and shade the CORE while they travel (falls back to the plain sequence when the tile is too small to split)."""
        self.ctx, self.spec, self.g = ctx, spec, g
        self.sky = sky
        self.n_lights = int(len(lights_np))
        self.lights = ctx.upload(lights_np) if self.n_lights else None
        # env: plain prefiltered chain (pbr_prefilter_env); the shade samples its padded copy (one-shot)
        self.lut, self.lut_res, self.env_size, self.env_mips = lut, lut_res, env_size, env_mips
        self.env = ctx.env_pad(env, env_size, env_mips)
        self.allreduce = allreduce
        ew, eh = spec.ew, spec.eh
        self.clusters = ctx.alloc_clusters()
        self.hdr = ctx.zeros((spec.sh, spec.sw, 4), torch.float16)   # covers S (== E in apron mode)
        self.chain_a = ctx.alloc_bloom_chain(ew, eh)
        self.chain_b = ctx.alloc_bloom_chain(ew, eh)
        self.hist = ctx.zeros((HISTOGRAM_BINS,), torch.int32)
        self._tail_overlap = False
        self.avg = ctx.zeros((1,), torch.float32)
        # fused_exposure: average + tone-map as ONE launch (pbr_average_tonemap), which reads one histogram / luminance cell and writes
        # the other — the frame alternates two of each; `hist` / `avg` always name the current ones.  Off by default: measured in round 4
        # (tools/frame_ab.py, profiles/r04_g_frame_ab.txt) the launch it saves does not show in the frame (0.4670 / 0.4672 / 0.4680 vs
        # 0.4659 / 0.4682 / 0.4674 ms), so the frame keeps the reference's two dispatches
        self.fused_exposure = fused_exposure
        self._hist_next = ctx.zeros((HISTOGRAM_BINS,), torch.int32)
        self._avg_next = ctx.zeros((1,), torch.float32)
        self.ldr = ctx.zeros((spec.h, spec.w), torch.int32)
        self.gb = None
        self.tile = Tile(spec.sx0, spec.sy0, spec.sw, spec.sh, spec.full_w, spec.full_h)
        self.halo_transport = halo_transport
        if spec.halo:
            assert all_specs is not None and halo_transport is not None, "halo mode needs every rank's TileSpec and a transport"
            from .structs import bloom_level_offset
            o = bloom_level_offset(ew, eh, 1)
            self.level1 = self.chain_a[o:o + (ew // 2) * (eh // 2)]   # level 1 of chain A: the plane the halo fills
            hx, hy = spec.ex0 // 2, spec.ey0 // 2

            def loc(r):   # global half-res rect (x0,y0,x1,y1) -> (x, y, w, h) in the level-1 plane of E
                return None if r is None else (r[0] - hx, r[1] - hy, r[2] - r[0], r[3] - r[1])
            self.halo_plan_local = [(n, loc(snd), loc(rcv)) for n, snd, rcv in halo_plan(rank, len(all_specs), all_specs)]
            self.halo_peers, self.halo_n = ctx.halo_peers(self.halo_plan_local)
            self.halo_staging = ctx.zeros((max(ctx.halo_staging_bytes(self.halo_peers, self.halo_n) // 8, 1), 4), torch.float16)
        self.split = self._ring_core_split() if (spec.halo and overlap) else None

    def _ring_core_split(self):
        """Rectangles of the overlapped halo frame.  The strips a neighbour needs are the interior's level-1 texels within
        128 half-res texels of the shared edge; their prefilter reads full-res pixels up to 256 + 3 px inside the edge.
        Returns (shade_ring, shade_core, l1_ring, l1_core): shade rects in S-local pixels, level-1 rects in half-res texels
        of the S image; ring U core = S resp. interior / 2, disjoint.  None if the tile is too small to be worth it."""
        s = self.spec
        L, R = s.x0 > 0, s.x0 + s.w < s.full_w
        T, B = s.y0 > 0, s.y0 + s.h < s.full_h
        reach = s.apron + 4                                  # 256 + 3, rounded to keep rectangle edges even
        left = s.six + reach if L else 0
        right = (s.sw - s.six - s.w) + reach if R else 0
        top = s.siy + reach if T else 0
        bot = (s.sh - s.siy - s.h) + reach if B else 0
        mid_w, mid_h = s.sw - left - right, s.sh - top - bot
        if mid_w < 256 or mid_h < 64:
            return None
        ring = [r for r in ((0, 0, s.sw, top), (0, s.sh - bot, s.sw, bot), (0, top, left, mid_h), (s.sw - right, top, right, mid_h))
                if r[2] > 0 and r[3] > 0]
        core = (left, top, mid_w, mid_h)
        hb = s.apron // 2
        ox, oy, iw, ih = s.six // 2, s.siy // 2, s.w // 2, s.h // 2
        l, r_, t, b = (hb if L else 0), (hb if R else 0), (hb if T else 0), (hb if B else 0)
        l1_ring = [q for q in ((ox, oy, iw, t), (ox, oy + ih - b, iw, b), (ox, oy + t, l, ih - t - b), (ox + iw - r_, oy + t, r_, ih - t - b))
                   if q[2] > 0 and q[3] > 0]
        l1_core = (ox + l, oy + t, iw - l - r_, ih - t - b)
        return ring, core, l1_ring, l1_core

    def upload_gbuffer(self, gb_np):
        """gb_np: dict of numpy planes covering the SHADED rectangle S (sh x sw; the extended rectangle in apron mode)."""
        assert gb_np["A"].shape == (self.spec.sh, self.spec.sw)
        self.gb = {k: self.ctx.upload(v) for k, v in gb_np.items()}

    def set_prev_luminance(self, v):
        self.avg.fill_(float(v))

    # interior views (pointer + pitch) of the HDR buffer
    def _hdr_interior_ptr(self):
        s = self.spec
        return self.hdr.data_ptr() + 8 * (s.siy * s.sw + s.six)

    def clustered(self):
        self.ctx.clustered(self.g, self.lights, self.n_lights, self.clusters)

    def skybox(self):
        s = self.spec
        cube, size, mips = self.sky
        self.ctx.skybox(self.g, self.tile, cube, size, mips, self.gb["stencil"], s.sw, self.hdr, s.sw)

    def shade(self):
        s = self.spec
        self.ctx.deferred_shade(self.g, self.tile, self.gb, s.sw, self.lut, self.lut_res, self.env, self.env_size,
                                self.env_mips, self.clusters, self.lights, self.n_lights, self.hdr, s.sw)

    def shade_rects(self, rects):
        """The shade on rectangles (x, y, w, h) of the shaded rectangle S, ONE launch (pbr_deferred_shade_rects)."""
        s = self.spec
        self.ctx.deferred_shade_rects(self.g, self.tile, self.gb, s.sw, self.lut, self.lut_res, self.env, self.env_size,
                                      self.env_mips, self.clusters, self.lights, self.n_lights, self.hdr, s.sw, rects)

    def prefilter_l1_rects(self, rects):
        s = self.spec
        self.ctx.bloom_prefilter_rects(self.hdr, s.sw, s.sh, s.sw, self.level1, s.ew // 2, (s.sx0 - s.ex0) // 2, (s.sy0 - s.ey0) // 2, rects)

    def shade_and_bloom_overlapped(self, histogram=True):
        """Halo frame with the exchange hidden behind the core's shade: ring -> strips -> exchange || core -> pyramid
        (two shade launches and two prefilter launches instead of one each)."""
        ring, core, l1_ring, l1_core = self.split
        ctx, tr = self.ctx, self.halo_transport
        if tr.kind == "torch":     # torch's P2P ops follow torch's current stream: plain fork / join around the exchange
            self.shade_rects(ring)
            self.prefilter_l1_rects(l1_ring)
            tr.begin(self)
            self.shade_rects([core])
            tr.end(self)
        else:                      # ring + its strips + the exchange on the context's high-priority side stream,
            ctx.side_begin()       # the core on the main stream at the same time
            self.shade_rects(ring)
            self.prefilter_l1_rects(l1_ring)
            if tr.kind == "host":
                ctx.side_end()     # the host stand-in synchronises the device anyway
                ctx.side_join()
                tr.exchange(self)
            else:
                tr.exchange(self)  # pbr_halo_exchange, enqueued on the side stream
                ctx.side_end()
            self.shade_rects([core])
            ctx.side_join()
        self.prefilter_l1_rects([l1_core])   # reads 3 px into the ring: after the join
        self.halo_pyramid(histogram)

    def bloom(self):
        s = self.spec
        if s.halo:
            self.bloom_halo(histogram=False)
        else:
            self.ctx.bloom(self.hdr, s.ew, s.eh, s.ew, self.chain_a, self.chain_b)

    def histogram(self):
        s = self.spec
        self.ctx.lum_histogram(self._hdr_interior_ptr(), s.w, s.h, s.sw, self.hist)

    # ---- halo mode: prefilter the interior, fetch the rest of level 1 from the neighbours, run levels 1..4 on E
    def halo_prefilter(self):
        s = self.spec
        self.ctx.bloom_prefilter_rect(self.hdr, s.sw, s.sh, s.sw, self.level1, s.ew // 2, (s.sx0 - s.ex0) // 2, (s.sy0 - s.ey0) // 2,
                                      (s.six // 2, s.siy // 2, s.w // 2, s.h // 2))

    def halo_exchange(self):
        self.halo_transport.exchange(self)

    def halo_pyramid(self, histogram=True):
        s = self.spec
        self.ctx.bloom_tiled(self.hdr, s.sw, (s.sx0 - s.ex0, s.sy0 - s.ey0, s.sw, s.sh), s.ew, s.eh, self.chain_a, self.chain_b,
                             (s.ix, s.iy, s.w, s.h), self.hist if histogram else None)

    def bloom_halo(self, histogram=True):
        self.halo_prefilter()
        self.halo_exchange()
        self.halo_pyramid(histogram)

    def bloom_histogram(self):
        """Bloom with the interior-tile luminance histogram accumulated in its final kernel."""
        s = self.spec
        if s.halo:
            self.bloom_halo(histogram=True)
        else:
            self.ctx.bloom_histogram(self.hdr, s.ew, s.eh, s.ew, self.chain_a, self.chain_b, (s.ix, s.iy, s.w, s.h), self.hist)

    def average(self):
        s = self.spec
        self.ctx.lum_average(self.hist, s.full_w * s.full_h, float(self.g.DeltaTime), self.avg)

    def tonemap(self):
        s = self.spec
        self.ctx.tonemap(self._hdr_interior_ptr(), s.w, s.h, s.sw, self.avg, self.ldr, s.w)

    def average_tonemap(self):
        """The frame's last two dispatches as one launch; afterwards `avg` is the new adapted luminance and `hist` the zeroed
        histogram the next frame accumulates into."""
        s = self.spec
        self.ctx.average_tonemap(self.hist, s.full_w * s.full_h, float(self.g.DeltaTime), self.avg, self._avg_next, self._hist_next,
                                 self._hdr_interior_ptr(), s.w, s.h, s.sw, self.ldr, s.w)
        self.avg, self._avg_next = self._avg_next, self.avg
        self.hist, self._hist_next = self._hist_next, self.hist

    def exposure_and_tonemap(self):
        if self.fused_exposure:
            self.average_tonemap()
        else:
            self.average()
            self.tonemap()

    def render(self, shade_events=None):
        """One frame: every per-frame dispatch of the reference, in the frame graph's order.
        shade_events: optional list; a (start, end) pair of torch events bracketing the shade launch is appended."""
        self.clustered()
        if self.sky is not None:
            self.skybox()
        if self.split is not None:
            e0 = e1 = None
            if shade_events is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            self.shade_and_bloom_overlapped()
            if shade_events is not None:   # brackets shade + bloom here: the two are interleaved
                e1.record()
                shade_events.append((e0, e1))
            if self.allreduce is not None:
                self.allreduce(self.hist)
            self.exposure_and_tonemap()
            return
        if shade_events is None:
            self.shade()
        else:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            self.shade()
            e1.record()
            shade_events.append((e0, e1))
        if self._tail_overlap == "bloom":
            # everything behind the shade — the bloom chain (8 launches, HBM / latency-bound), average, tone-map — on the context's
            # high-priority side stream, beside the NEXT frame's cluster pass and shade (FP32-issue-bound), which write the other
            # HDR / histogram buffer.  The join makes the main stream wait for the side work of the frame BEFORE (it read the HDR
            # buffer the next shade overwrites); the chains and the adapted luminance are touched by the side stream only, in order.
            self.ctx.side_join()
            self.ctx.side_begin()
            self.bloom_histogram()
            if self.allreduce is not None:
                self.allreduce(self.hist)
            self.average()
            self.tonemap()
            self.ctx.side_end()
            self.hdr, self._hdr_alt = self._hdr_alt, self.hdr
            self.hist, self._hist_alt = self._hist_alt, self.hist
            return
        self.bloom_histogram()
        if self._tail_overlap:
            # the frame's tail — histogram all-reduce, average, tone-map — on the context's side stream: the collective's latency
            # and the two small launches run beside the NEXT frame's cluster pass and shade, which write the other HDR /
            # histogram buffer.  The join orders the side work of the frame BEFORE last ahead of this point (long finished).
            self.ctx.side_join()
            self.ctx.side_begin()
            if self.allreduce is not None:
                self.allreduce(self.hist)
            self.average()
            self.tonemap()
            self.ctx.side_end()
            self.hdr, self._hdr_alt = self._hdr_alt, self.hdr
            self.hist, self._hist_alt = self._hist_alt, self.hist
            return
        if self.allreduce is not None:
            self.allreduce(self.hist)
        self.exposure_and_tonemap()

    def enable_tail_overlap(self, capi_allreduce=False, from_bloom=False):
        """Throughput mode: double-buffer the HDR target and the histogram so that a frame's tail — histogram all-reduce, average,
        tone-map; with from_bloom=True the bloom chain as well, i.e. everything behind the shade — runs on the context's side stream
        beside the NEXT frame's cluster pass and shade.  The LDR image of frame i is complete when frame i + 1's tail has been
        joined (or after ctx.sync(), which also waits for the side stream).  With an all-reduce (world > 1) it MUST be one that is
        enqueued on the context's current stream — PbrContext.allreduce_hist, or a stand-in the caller vouches for with
        capi_allreduce=True (tests: a 1-rank communicator); a torch.distributed all-reduce runs on torch's own stream and would
        race with the double-buffered histogram.  from_bloom is for frames without a halo exchange (one GPU, apron mode)."""
        if self.split is not None:
            raise ValueError("tail overlap needs the plain frame order")
        if self.allreduce is not None:
            own = getattr(self.allreduce, "__self__", None) is self.ctx and getattr(self.allreduce, "__name__", "") == "allreduce_hist"
            if not (own or capi_allreduce):
                raise ValueError("tail overlap needs the C ABI's all-reduce (PbrContext.allreduce_hist): it must be enqueued on the context's side stream")
        if from_bloom and self.spec.halo:
            raise ValueError("from_bloom: the halo exchange stays on the frame's stream")
        self._hdr_alt = torch.zeros_like(self.hdr)
        self._hist_alt = torch.zeros_like(self.hist)
        self._tail_overlap = "bloom" if from_bloom else True

    def finish(self):
        """Wait (on the frame's stream) for an overlapped tail still in flight."""
        if self._tail_overlap:
            self.ctx.side_join()

    # ---- read-back helpers for tests ---------------------------------------------------------------
    def hdr_interior(self):
        s = self.spec
        return self.hdr[s.siy:s.siy + s.h, s.six:s.six + s.w].cpu().numpy()

    def ldr_numpy(self):
        return self.ldr.cpu().numpy().view(np.uint32)
