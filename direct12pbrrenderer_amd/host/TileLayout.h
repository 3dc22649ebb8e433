// TileLayout.h — how one device's share of a frame is cut out (nothing like it in the reference, which
// renders one frame on one device).  Pure host arithmetic, no device calls.
//
// A frame of FullW x FullH pixels is cut into Cols x Rows equal tiles; rank r owns tile (r % Cols, r / Cols): its
// INTERIOR.  Bloom's cumulative support is ~220 full-res pixels, so the bloom pyramid of a tile runs on the extended
// rectangle E = interior + Apron on every side that has a neighbour (clipped to the frame).  Two ways to fill E:
//   apron mode : the device shades all of E (S == E) — no data-path collective, 42 % extra pixels on an inner cfg5 tile;
//   halo  mode : the device shades S = interior + 4 px (what the bloom prefilter's five taps reach: full-res pixels
//                2x-3 .. 2x+2 per half-res texel, bloom_prefilter.hlsl:17-60), computes the level-1 (half-res) texels of
//                its interior, and RECEIVES the level-1 texels of E outside its interior from the devices that own them.
// All rectangles are in GLOBAL pixels of the full frame.  Tile sizes and the apron must be multiples of 16
// (2^(BloomPass::MipmapLevel - 1)) so that every mip of E sits on the full frame's texel grid.
#pragma once
#include <stdexcept>
#include <vector>

#include "../../include/pbr_hip.h"

namespace MRendererHip {

struct PixelRect {
    uint32_t x = 0, y = 0, w = 0, h = 0;
    uint32_t x1() const { return x + w; }
    uint32_t y1() const { return y + h; }
};

struct TileLayout {
    static constexpr uint32_t DefaultApron = 256;   // >= bloom's support, multiple of 16
    static constexpr uint32_t HaloShadeApron = 4;   // bloom_prefilter.hlsl reaches 3 px beyond the interior; 4 keeps edges even

    uint32_t FullW = 0, FullH = 0;
    uint32_t Cols = 1, Rows = 1, Rank = 0;
    uint32_t Apron = 0;
    bool Halo = false;
    PixelRect Interior, Shaded, Bloom;   // I, S, E

    bool Tiled() const { return Cols * Rows > 1; }
    uint32_t World() const { return Cols * Rows; }
    // S and the interior inside E / the interior inside S (local offsets)
    PixelRect ShadedInBloom() const { return PixelRect{Shaded.x - Bloom.x, Shaded.y - Bloom.y, Shaded.w, Shaded.h}; }
    PixelRect InteriorInBloom() const { return PixelRect{Interior.x - Bloom.x, Interior.y - Bloom.y, Interior.w, Interior.h}; }
    PixelRect InteriorInShaded() const { return PixelRect{Interior.x - Shaded.x, Interior.y - Shaded.y, Interior.w, Interior.h}; }

    static TileLayout Single(uint32_t w, uint32_t h) {
        TileLayout t;
        t.FullW = w;
        t.FullH = h;
        t.Interior = t.Shaded = t.Bloom = PixelRect{0, 0, w, h};
        return t;
    }

    // rank's tile of a full_w x full_h frame cut cols x rows (BASELINE cfg5: 7680x4320, cols 4, rows 2)
    static TileLayout OfFrame(uint32_t full_w, uint32_t full_h, uint32_t cols, uint32_t rows, uint32_t rank, bool halo,
                              uint32_t apron = DefaultApron) {
        if (cols < 1 || rows < 1 || rank >= cols * rows) throw std::invalid_argument("TileLayout: bad grid / rank");
        if (full_w % cols || full_h % rows) throw std::invalid_argument("TileLayout: the frame does not split into equal tiles");
        if (full_w > 65535 || full_h > 65535) throw std::invalid_argument("TileLayout: the frame exceeds 65535 pixels on a side");
        const uint32_t tw = full_w / cols, th = full_h / rows;
        const bool multi = cols * rows > 1;
        if (multi && (tw % 16 || th % 16 || apron % 16)) throw std::invalid_argument("TileLayout: tile size and apron must be multiples of 16");
        TileLayout t;
        t.FullW = full_w;
        t.FullH = full_h;
        t.Cols = cols;
        t.Rows = rows;
        t.Rank = rank;
        t.Apron = multi ? apron : 0;
        t.Halo = halo && multi && apron > 0;
        t.Interior = PixelRect{(rank % cols) * tw, (rank / cols) * th, tw, th};
        t.Bloom = Grow(t.Interior, t.Apron, full_w, full_h);
        t.Shaded = t.Halo ? Grow(t.Interior, HaloShadeApron, full_w, full_h) : t.Bloom;
        return t;
    }

    // Level-1 strips this rank exchanges with every other rank, as pbr_halo_peer records whose rectangles are LOCAL to
    // the level-1 plane of E (pitch Bloom.w / 2).  Rank r needs level 1 on E_r / 2; it computes it on I_r / 2 and
    // receives (E_r / 2 intersect I_n / 2) from every rank n — which covers E_r / 2 exactly, because the interiors
    // partition the frame and E is clipped to it.  Both sides derive the same rectangles from the same grid, so no
    // sizes travel.
    std::vector<pbr_halo_peer> HaloPlan() const {
        std::vector<pbr_halo_peer> plan;
        if (!Halo) return plan;
        const Half me_e = Halve(Bloom), me_i = Halve(Interior);
        for (uint32_t n = 0; n < World(); n++) {
            if (n == Rank) continue;
            const TileLayout o = OfFrame(FullW, FullH, Cols, Rows, n, true, Apron);
            const Half recv = Intersect(me_e, Halve(o.Interior)), send = Intersect(Halve(o.Bloom), me_i);
            if (recv.Empty() && send.Empty()) continue;
            pbr_halo_peer p{};
            p.rank = (int32_t)n;
            Local(send, me_e, p.send);
            Local(recv, me_e, p.recv);
            plan.push_back(p);
        }
        return plan;
    }

private:
    struct Half {
        uint32_t x0, y0, x1, y1;
        bool Empty() const { return x0 >= x1 || y0 >= y1; }
    };
    static Half Halve(const PixelRect& r) { return Half{r.x / 2, r.y / 2, r.x1() / 2, r.y1() / 2}; }
    static Half Intersect(const Half& a, const Half& b) {
        return Half{a.x0 > b.x0 ? a.x0 : b.x0, a.y0 > b.y0 ? a.y0 : b.y0, a.x1 < b.x1 ? a.x1 : b.x1, a.y1 < b.y1 ? a.y1 : b.y1};
    }
    static void Local(const Half& r, const Half& origin, uint32_t out[4]) {
        if (r.Empty()) { out[0] = out[1] = out[2] = out[3] = 0; return; }
        out[0] = r.x0 - origin.x0;
        out[1] = r.y0 - origin.y0;
        out[2] = r.x1 - r.x0;
        out[3] = r.y1 - r.y0;
    }
    static PixelRect Grow(const PixelRect& r, uint32_t a, uint32_t full_w, uint32_t full_h) {
        const uint32_t x0 = r.x > a ? r.x - a : 0, y0 = r.y > a ? r.y - a : 0;
        const uint32_t x1 = r.x1() + a < full_w ? r.x1() + a : full_w, y1 = r.y1() + a < full_h ? r.y1() + a : full_h;
        return PixelRect{x0, y0, x1 - x0, y1 - y0};
    }
};

}  // namespace MRendererHip
