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


@dataclass
class TileSpec:
    """Interior tile (x0,y0,w,h) of a full_w x full_h frame and its apron-extended rectangle."""
    x0: int
    y0: int
    w: int
    h: int
    full_w: int
    full_h: int
    apron: int = 0

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
    def ix(self):   # interior offset inside the extended rectangle
        return self.x0 - self.ex0

    @property
    def iy(self):
        return self.y0 - self.ey0


def grid_for_world(world, tile_w=16, tile_h=9):
    """Tile grid (cols, rows) for `world` ranks of equal tiles: the tiles are laid out in ONE row (landscape tiles)
    or one column (portrait tiles), so that the shared edges are the tiles' short sides and a tile has at most two
    neighbours.  For 3840x2160 tiles with a 256-px apron that is 6.7 % (2 ranks) / 13.3 % (>= 3 ranks) extra
    pixels on the busiest rank; a 2x2 / 4x2 arrangement (BASELINE cfg5 cuts its 8K frame that way) costs 19 % / 27 %."""
    if world < 1:
        raise ValueError("world size must be >= 1")
    return (world, 1) if tile_w >= tile_h else (1, world)


def tile_for_rank(rank, world, tile_w, tile_h, apron=DEFAULT_APRON):
    """Weak scaling: every rank owns one tile_w x tile_h tile of a (cols*tile_w) x (rows*tile_h) frame."""
    cols, rows = grid_for_world(world, tile_w, tile_h)
    cx, cy = rank % cols, rank // cols
    if world > 1 and (tile_w % 16 or tile_h % 16 or apron % 16):
        # every bloom mip of the extended tile must sit on the full frame's texel grid (2^(BLOOM_MIPS-1) = 16)
        raise ValueError("tile size and apron must be multiples of 16 for multi-GPU tiling")
    if cols * tile_w > 65535 or rows * tile_h > 65535:
        raise ValueError("the assembled frame exceeds 65535 pixels on a side")
    return TileSpec(cx * tile_w, cy * tile_h, tile_w, tile_h, cols * tile_w, rows * tile_h, apron if world > 1 else 0)


class DeferredFrame:
    """Owns the device buffers of one rank and runs the per-frame passes through the C ABI."""

    def __init__(self, ctx: PbrContext, spec: TileSpec, g: Global, lights_np, lut, lut_res, env, env_size,
                 env_mips=ENV_MIPS, allreduce=None, sky=None):
        """sky: optional (cube tensor fp32 RGBA with mips, size, mips) — resolved on stencil == 0 pixels
        before the shade like the reference's SkyboxPass; without it those pixels keep what the buffer holds."""
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
        self.hdr = ctx.zeros((eh, ew, 4), torch.float16)
        self.chain_a = ctx.alloc_bloom_chain(ew, eh)
        self.chain_b = ctx.alloc_bloom_chain(ew, eh)
        self.hist = ctx.zeros((HISTOGRAM_BINS,), torch.int32)
        self.avg = ctx.zeros((1,), torch.float32)
        self.ldr = ctx.zeros((spec.h, spec.w), torch.int32)
        self.gb = None
        self.tile = Tile(spec.ex0, spec.ey0, ew, eh, spec.full_w, spec.full_h)

    def upload_gbuffer(self, gb_np):
        """gb_np: dict of numpy planes covering the EXTENDED rectangle (eh x ew)."""
        assert gb_np["A"].shape == (self.spec.eh, self.spec.ew)
        self.gb = {k: self.ctx.upload(v) for k, v in gb_np.items()}

    def set_prev_luminance(self, v):
        self.avg.fill_(float(v))

    # interior views (pointer + pitch) of the extended HDR buffer
    def _hdr_interior_ptr(self):
        s = self.spec
        return self.hdr.data_ptr() + 8 * (s.iy * s.ew + s.ix)

    def clustered(self):
        self.ctx.clustered(self.g, self.lights, self.n_lights, self.clusters)

    def skybox(self):
        s = self.spec
        cube, size, mips = self.sky
        self.ctx.skybox(self.g, self.tile, cube, size, mips, self.gb["stencil"], s.ew, self.hdr, s.ew)

    def shade(self):
        s = self.spec
        self.ctx.deferred_shade(self.g, self.tile, self.gb, s.ew, self.lut, self.lut_res, self.env, self.env_size,
                                self.env_mips, self.clusters, self.lights, self.n_lights, self.hdr, s.ew)

    def bloom(self):
        s = self.spec
        self.ctx.bloom(self.hdr, s.ew, s.eh, s.ew, self.chain_a, self.chain_b)

    def histogram(self):
        s = self.spec
        self.ctx.lum_histogram(self._hdr_interior_ptr(), s.w, s.h, s.ew, self.hist)

    def bloom_histogram(self):
        """Bloom with the interior-tile luminance histogram accumulated in its final kernel."""
        s = self.spec
        self.ctx.bloom_histogram(self.hdr, s.ew, s.eh, s.ew, self.chain_a, self.chain_b, (s.ix, s.iy, s.w, s.h), self.hist)

    def average(self):
        s = self.spec
        self.ctx.lum_average(self.hist, s.full_w * s.full_h, float(self.g.DeltaTime), self.avg)

    def tonemap(self):
        s = self.spec
        self.ctx.tonemap(self._hdr_interior_ptr(), s.w, s.h, s.ew, self.avg, self.ldr, s.w)

    def render(self, shade_events=None):
        """One frame: every per-frame dispatch of the reference, in the frame graph's order.
        shade_events: optional list; a (start, end) pair of torch events bracketing the shade launch is appended."""
        self.clustered()
        if self.sky is not None:
            self.skybox()
        if shade_events is None:
            self.shade()
        else:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            self.shade()
            e1.record()
            shade_events.append((e0, e1))
        self.bloom_histogram()
        if self.allreduce is not None:
            self.allreduce(self.hist)
        self.average()
        self.tonemap()

    # ---- read-back helpers for tests ---------------------------------------------------------------
    def hdr_interior(self):
        s = self.spec
        return self.hdr[s.iy:s.iy + s.h, s.ix:s.ix + s.w].cpu().numpy()

    def ldr_numpy(self):
        return self.ldr.cpu().numpy().view(np.uint32)
