// bloom.hip — bloom chain on gfx950: soft-knee prefilter, separable 9-tap Gaussian pyramid,
// upsample-add and merge (bloom_prefilter.hlsl, blur.hlsli, blur_horizontal/vertical.hlsl,
// bloom_upsample_add.hlsl, bloom_merge.hlsl; schedule BloomPass::Execute,
// DeferredPipeline.cpp:400-570).
//
// This translation unit is compiled with -ffp-contract=off and follows the oracle's operation
// order (fused multiply-adds only where the oracle writes fmaf: sampler lerps and the blur's
// multiply-accumulate), so every stage is bit-identical to the CPU oracle on the same input.
//
// Two sets of kernels, bit-identical to each other:
//  * the STAGED kernels below, one per reference dispatch (pbr_bloom_prefilter, pbr_blur_h, pbr_blur_v,
//    pbr_bloom_upsample_add, pbr_bloom_merge; any image size) — what the host pass graph issues one by one;
//  * the FUSED kernels further down (k_bloom_prefilter_2x, k_blur_hv), which pbr_bloom / pbr_bloom_histogram use on
//    an exact 2x pyramid: shared prefilter samples, H + V pass of a level in one kernel, merge + histogram in the last.
//
// Layout / mapping of the staged kernels for MI355X:
//  * H passes keep the reference's 256-texel row groups (one 64-lane wave = 64 consecutive
//    texels = one 512-byte half4 segment), the bilinear-sampled row is cached in LDS as float4
//    (264 entries, ds_read_b128, conflict-free) exactly like blur.hlsli's Cache[].
//  * V passes do NOT use the reference's 1x256 column groups (one texel per 8 KiB-strided row =
//    64 cache lines per wave); they use 64x16 tiles: lanes run along x, the (16+8) sampled rows
//    of the tile are staged through LDS, so global accesses stay 512-byte coalesced.
#include <type_traits>
#include "pbr_internal.hpp"
#include "pbr_device.hpp"

using namespace pbr;

__constant__ float c_gauss[9] = {0.0148f, 0.0459f, 0.1050f, 0.1941f, 0.2803f, 0.1941f, 0.1050f, 0.0459f, 0.0148f};   // blur.hlsli:17

__device__ __forceinline__ float4 to4(F4 v) { return make_float4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ F4 from4(float4 v) { return f4(v.x, v.y, v.z, v.w); }

// ---------------------------------------------------------------- bloom_prefilter.hlsl:17-60
// grid (ceil(ow/64), ceil(oh/4)), block (64,4): one thread per half-res texel
// OutRect: the outputs [x0,x1) x [y0,y1) (half-res texels of this image) are computed and stored at
// out[(y + oy) * out_pitch + (x + ox)] — the whole image into a dense plane is {0,0,ow,oh}, pitch ow, offset 0;
// a tile's interior into the level-1 plane of its extended rectangle is the multi-GPU halo path.
struct OutRect { int x0, y0, x1, y1, ox, oy, pitch; };
// several output rectangles in one launch of the shared-sample kernel (1-D grid; the overlapped multi-GPU frame
// prefilters the four bands of its border ring at once): rectangle r owns blocks first[r] .. first[r+1]-1
constexpr int PF_MAX_RECTS = 5;
struct OutRects {
    int n, ox, oy, pitch;
    int x0[PF_MAX_RECTS], y0[PF_MAX_RECTS], x1[PF_MAX_RECTS], y1[PF_MAX_RECTS], tiles_x[PF_MAX_RECTS], first[PF_MAX_RECTS + 1];
};
__global__ __launch_bounds__(256) void k_bloom_prefilter(const pbr_half* __restrict__ hdr, int w, int h, int pitch,
                                                           pbr_half* __restrict__ out, OutRect rc,
                                                           float tx, float ty, float threshold, float knee) {
    const int x = rc.x0 + blockIdx.x * 64 + threadIdx.x;
    const int y = rc.y0 + blockIdx.y * 4 + threadIdx.y;
    if (x >= rc.x1 || y >= rc.y1) return;
    const float u = (float)x * tx, v = (float)y * ty;   // no +0.5 (Q9)
    const float ox[5] = {0.0f, -1.0f, -1.0f, 1.0f, 1.0f};
    const float oy[5] = {0.0f, -1.0f, 1.0f, -1.0f, 1.0f};
    float tr = 0.0f, tg = 0.0f, tb = 0.0f, tw = 0.0f;
#pragma unroll
    for (int i = 0; i < 5; i++) {
        const F4 c = sample_2d_h4(hdr, w, h, pitch, u + ox[i] * tx, v + oy[i] * ty);
        const float brightness = fmaxf(c.x, fmaxf(c.y, c.z));
        float soft = fminf(fmaxf(brightness - threshold + threshold * knee, 0.0f), 2.0f * threshold * knee);
        soft /= 4.0f * threshold * knee + 0.00001f;
        const float contribution = fmaxf(soft, brightness - threshold) / fmaxf(brightness, 0.00001f);
        const float cr = c.x * contribution, cg = c.y * contribution, cb = c.z * contribution;
        const float wgt = 1.0f / (luminance(cr, cg, cb) + 1.0f);
        tr += cr * wgt; tg += cg * wgt; tb += cb * wgt;
        tw += wgt;
    }
    if (tw > 0.0f) { tr /= tw; tg /= tw; tb /= tw; }
    store_h4(out + 4 * ((size_t)(y + rc.oy) * rc.pitch + (x + rc.ox)), f4(tr, tg, tb, 1.0f));
}

// ---------------------------------------------------------------- blur.hlsli:24-55
// One 256-thread group = 256 consecutive output texels of a row, the bilinear-sampled row cached in LDS
// (264 float4 entries incl. the 4+4 halo slots) exactly like the shader's Cache[].  A block walks HB_ROWS
// rows of its column group and software-pipelines them: the four raw taps of row r+1 are in flight while
// row r is filtered out of a double-buffered cache (one barrier per row).  The stand-alone one-row block
// was latency-bound (one dependent HBM round trip per 256 outputs, SQ_WAIT_ANY ~50 %).
constexpr int HB_MAX_ROWS = 8;   // rows per block: chosen per launch so that small pyramid levels still fill the chip
struct RawTap {   // the four texels of one bilinear sample, still in half precision, + the y weight
    H4 c00, c10, c01, c11;
    float fy;
};
struct ColumnCoord { int x0, x1; float fx; };   // x side of a sample: row-invariant
__device__ __forceinline__ ColumnCoord column_coord(float u, int iw) {
    const BilinearCoord c = bilinear_coord(u, iw);
    return ColumnCoord{clampi(c.i0, 0, iw - 1), clampi(c.i1, 0, iw - 1), c.f};
}
__device__ __forceinline__ RawTap load_tap(const pbr_half* __restrict__ in, int iw, int ih, const ColumnCoord& cc, float v) {
    const BilinearCoord cy = bilinear_coord(v, ih);
    const int y0 = clampi(cy.i0, 0, ih - 1), y1 = clampi(cy.i1, 0, ih - 1);
    const H4* r0 = reinterpret_cast<const H4*>(in) + (size_t)y0 * iw;
    const H4* r1 = reinterpret_cast<const H4*>(in) + (size_t)y1 * iw;
    return RawTap{r0[cc.x0], r0[cc.x1], r1[cc.x0], r1[cc.x1], cy.f};
}
__device__ __forceinline__ F4 h4f(H4 h) { return f4((float)h.x, (float)h.y, (float)h.z, (float)h.w); }
__device__ __forceinline__ float4 finish_tap(const RawTap& r, float fx) {
    return to4(bilerp(h4f(r.c00), h4f(r.c10), h4f(r.c01), h4f(r.c11), fx, r.fy));
}
__device__ __forceinline__ F4 gauss9(const float4* c) {
    F4 v = f4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
    for (int i = 0; i < 9; i++) v = fma4(from4(c[i]), c_gauss[i], v);   // value += pixel * weight (fused mad)
    return v;
}

// grid (ceil(ow/256), ceil(oh/rows_per_block)), block 256.  DUAL: bloom_upsample_add.hlsl:13-25 (lower first, then upper)
template <bool DUAL>
__global__ __launch_bounds__(256) void k_blur_h(const pbr_half* __restrict__ in, int iw, int ih,
                                                 const pbr_half* __restrict__ in2, int iw2, int ih2,
                                                 pbr_half* __restrict__ out, int ow, int oh, float tx, float ty, int rows_per_block) {
    constexpr int NS = DUAL ? 2 : 1;
    __shared__ float4 cache[2][NS][264];
    const int t = threadIdx.x;
    const int gx0 = blockIdx.x * 256;
    const int y_begin = blockIdx.y * rows_per_block, y_end = min(y_begin + rows_per_block, oh);
    // sample positions in x (blur.hlsli:26-43): every thread its own texel; threads 0-3 / 252-255 also one halo tap
    const float uvx = ((float)(gx0 + t) + 0.5f) * tx;
    const bool halo = (t < 4) | (t >= 252);
    const float uvx_h = t < 4 ? fmaxf(uvx - 4.0f * tx, 0.0f) : fminf(uvx + 4.0f * tx, 1.0f);
    const int slot_h = t < 4 ? t : t + 8;
    const ColumnCoord cm = column_coord(uvx, iw), ch = column_coord(uvx_h, iw);
    ColumnCoord cm2 = cm, ch2 = ch;
    if (DUAL) { cm2 = column_coord(uvx, iw2); ch2 = column_coord(uvx_h, iw2); }

    RawTap m[NS], hh[NS];
    auto load_row = [&](int y) {
        const float uvy = ((float)y + 0.5f) * ty;
        m[0] = load_tap(in, iw, ih, cm, uvy);
        if (halo) hh[0] = load_tap(in, iw, ih, ch, uvy);
        if (DUAL) {
            m[NS - 1] = load_tap(in2, iw2, ih2, cm2, uvy);
            if (halo) hh[NS - 1] = load_tap(in2, iw2, ih2, ch2, uvy);
        }
    };
    load_row(y_begin);
    for (int y = y_begin; y < y_end; y++) {
        const int buf = (y - y_begin) & 1;
        cache[buf][0][t + 4] = finish_tap(m[0], cm.fx);
        if (halo) cache[buf][0][slot_h] = finish_tap(hh[0], ch.fx);
        if (DUAL) {
            cache[buf][NS - 1][t + 4] = finish_tap(m[NS - 1], cm2.fx);
            if (halo) cache[buf][NS - 1][slot_h] = finish_tap(hh[NS - 1], ch2.fx);
        }
        __syncthreads();
        if (y + 1 < y_end) load_row(y + 1);   // in flight while this row is filtered
        const int x = gx0 + t;
        if (x < ow) {
            F4 v = gauss9(cache[buf][0] + t);
            if (DUAL) v = v + gauss9(cache[buf][NS - 1] + t);
            store_h4(out + 4 * ((size_t)y * ow + x), v);
        }
    }
}

// ---------------------------------------------------------------- blur.hlsli:58-89
// 64 x 16 output tiles; TR divides 256 so a tile never straddles one of the reference's
// 256-row groups, which decides whether a halo row uses the group-edge position formula.
constexpr int VT_W = 64, VT_R = 16;
__global__ __launch_bounds__(256) void k_blur_v(const pbr_half* __restrict__ in, int iw, int ih,
                                                 pbr_half* __restrict__ out, int ow, int oh, float tx, float ty) {
    __shared__ float4 smp[VT_R + 8][VT_W];
    const int x = blockIdx.x * VT_W + threadIdx.x;
    const int y0 = blockIdx.y * VT_R;
    const bool top_edge = (y0 & 255) == 0;
    const bool bot_edge = ((y0 + VT_R) & 255) == 0;
    const float uvx = ((float)x + 0.5f) * tx;
    for (int r = threadIdx.y; r < VT_R + 8; r += 4) {
        const int j = y0 - 4 + r;   // sampled row (may be outside [0, oh))
        float vy;
        if (r < 4 && top_edge) {
            vy = fmaxf(((float)(j + 4) + 0.5f) * ty - 4.0f * ty, 0.0f);     // Cache[gtid.y], gtid.y < 4
        } else if (r >= VT_R + 4 && bot_edge) {
            vy = fminf(((float)(j - 4) + 0.5f) * ty + 4.0f * ty, 1.0f);     // Cache[gtid.y + 8], gtid.y >= 252
        } else {
            vy = ((float)j + 0.5f) * ty;
        }
        smp[r][threadIdx.x] = to4(sample_2d_h4(in, iw, ih, iw, uvx, vy));
    }
    __syncthreads();
    if (x >= ow) return;
    for (int r = threadIdx.y; r < VT_R; r += 4) {
        const int y = y0 + r;
        if (y >= oh) break;
        F4 v = f4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
        for (int i = 0; i < 9; i++) v = fma4(from4(smp[r + i][threadIdx.x]), c_gauss[i], v);
        store_h4(out + 4 * ((size_t)y * ow + x), v);
    }
}

// ---------------------------------------------------------------- final three dispatches fused
// A0 = V(B0) (blur_vertical.hlsl), S += A0 (bloom_merge.hlsl) and — optionally — the luminance
// histogram of the merged pixel (hdr_luminance_histogram.hlsl:23-59), in one pass over the frame:
// A0 is never written to HBM (nobody reads it afterwards) and the histogram pass no longer re-reads
// the HDR buffer.  Every intermediate is rounded exactly where the separate dispatches round
// (A0 to fp16, then the fp16 sum), so the HDR result is bit-identical to the unfused chain.
// Persistent blocks walk 64x16 tiles so the per-wave LDS histograms are flushed once per block.
__device__ __forceinline__ uint32_t luminance_bin_exact(float r, float g, float b, float min_log, float inv_range) {
    const float lum = (r * 0.2126f + g * 0.7152f) + b * 0.0722f;
    if (lum < EPSILON_F) return 0u;
    // lum >= 1e-6 is a normal number: log2f's subnormal pre-scaling never applies, so the bare v_log_f32 it wraps gives the same bits
    const float l = saturatef((__builtin_amdgcn_logf(lum) - min_log) * inv_range);
    return (uint32_t)floorf(l * 254.0f + 1.0f);
}

template <bool HIST>
__global__ __launch_bounds__(256) void k_blur_v_merge(const pbr_half* __restrict__ in, int w, int h, float tx, float ty,
                                                       pbr_half* __restrict__ hdr, int pitch, int tiles_x, int tiles_y,
                                                       int hx0, int hy0, int hx1, int hy1, float min_log, float inv_range,
                                                       uint32_t* __restrict__ hist) {
    __shared__ float4 smp[VT_R + 8][VT_W];
    __shared__ uint32_t sh_hist[HIST ? 4 : 1][HIST ? PBR_HISTOGRAM_BINS : 1];
    const int tid = threadIdx.y * VT_W + threadIdx.x;
    if (HIST) {
        for (int i = tid; i < 4 * PBR_HISTOGRAM_BINS; i += 256) (&sh_hist[0][0])[i] = 0u;
    }
    const int n_tiles = tiles_x * tiles_y;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int x = (tile % tiles_x) * VT_W + threadIdx.x;
        const int y0 = (tile / tiles_x) * VT_R;
        const bool top_edge = (y0 & 255) == 0;
        const bool bot_edge = ((y0 + VT_R) & 255) == 0;
        const float uvx = ((float)x + 0.5f) * tx;
        __syncthreads();   // previous tile's reads of smp are done (and the histogram clear on the first trip)
        for (int r = threadIdx.y; r < VT_R + 8; r += 4) {
            const int j = y0 - 4 + r;
            float vy;
            if (r < 4 && top_edge) vy = fmaxf(((float)(j + 4) + 0.5f) * ty - 4.0f * ty, 0.0f);
            else if (r >= VT_R + 4 && bot_edge) vy = fminf(((float)(j - 4) + 0.5f) * ty + 4.0f * ty, 1.0f);
            else vy = ((float)j + 0.5f) * ty;
            smp[r][threadIdx.x] = to4(sample_2d_h4(in, w, h, w, uvx, vy));
        }
        __syncthreads();
        if (x < w) {
            for (int r = threadIdx.y; r < VT_R; r += 4) {
                const int y = y0 + r;
                if (y >= h) break;
                F4 v = f4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
                for (int i = 0; i < 9; i++) v = fma4(from4(smp[r + i][threadIdx.x]), c_gauss[i], v);
                // A0 texel as the separate pass would have stored it
                H4 a0;
                a0.x = to_half_rn(v.x); a0.y = to_half_rn(v.y); a0.z = to_half_rn(v.z); a0.w = to_half_rn(v.w);
                pbr_half* px = hdr + 4 * ((size_t)y * pitch + x);
                const F4 s = load_h4(px);
                H4 o;
                o.x = to_half_rn(s.x + (float)a0.x); o.y = to_half_rn(s.y + (float)a0.y); o.z = to_half_rn(s.z + (float)a0.z); o.w = to_half_rn(s.w + (float)a0.w);
                *reinterpret_cast<H4*>(px) = o;
                if (HIST) {
                    if (x >= hx0 && x < hx1 && y >= hy0 && y < hy1)
                        atomicAdd(&sh_hist[threadIdx.y][luminance_bin_exact((float)o.x, (float)o.y, (float)o.z, min_log, inv_range)], 1u);
                }
            }
        }
    }
    if (HIST) {
        __syncthreads();
        const uint32_t s = (sh_hist[0][tid] + sh_hist[1][tid]) + (sh_hist[2][tid] + sh_hist[3][tid]);
        if (s) atomicAdd(&hist[tid], s);
    }
}

// ---------------------------------------------------------------- bloom_merge.hlsl:7-11
__global__ __launch_bounds__(256) void k_bloom_merge(pbr_half* __restrict__ hdr, int pitch, const pbr_half* __restrict__ in, int w, int h) {
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= w || y >= h) return;
    pbr_half* p = hdr + 4 * ((size_t)y * pitch + x);
    store_h4(p, load_h4(p) + load_h4(in + 4 * ((size_t)y * w + x)));
}

// =====================================================================================================
// Fast paths for exact 2x pyramids (every level exactly half the one above, sizes <= 8192).
//
// With the fixed-point sampler (pbr_device.hpp::bilinear_coord) every sample position of the bloom chain then
// snaps to an exact dyadic coordinate: a same-size sample IS the texel, a 2x-down sample is the mean of a 2x2
// quad (weights 1/2), a 2x-up sample has weights 1/4 | 3/4 — whatever float formula produced the coordinate
// (the shader's group-edge formulas, `u + offset * texel`, ...).  So a sample is a function of its INTEGER
// position alone, positions outside the image read the clamped edge texel, and
//   * the prefilter's five samples per output are shared between neighbouring outputs (evaluated once per
//     position, 1.16 instead of 5 evaluations per output — they carry the three IEEE divides of the soft knee),
//   * the V pass's same-size "bilinear" sample is an exact texel of its own column, so H pass + V pass fuse
//     into one kernel: a thread owns a column, walks down the rows and keeps the last nine H-blurred texels
//     (rounded to fp16 exactly where the H pass would have stored them) in registers — the H result never
//     goes to HBM.  The final instance adds the merge and the luminance histogram.
// Results are bit-identical to the staged kernels above (tests/test_gpu_parity.py), which remain the generic path.

constexpr int PF_TW = 64, PF_TH = 16;
__global__ __launch_bounds__(256) void k_bloom_prefilter_2x(const pbr_half* __restrict__ hdr, int w, int h, int pitch,
                                                              pbr_half* __restrict__ out, OutRects rs, float threshold, float knee) {
    __shared__ float4 pos[PF_TH + 2][PF_TW + 2];   // (colour * weight, weight) of every sample position the tile touches
    const int tid = threadIdx.x;
    int r = 0;
    while (r + 1 < rs.n && (int)blockIdx.x >= rs.first[r + 1]) r++;
    const int lb = (int)blockIdx.x - rs.first[r];
    const OutRect rc{rs.x0[r], rs.y0[r], rs.x1[r], rs.y1[r], rs.ox, rs.oy, rs.pitch};
    const int bx0 = rc.x0 + (lb % rs.tiles_x[r]) * PF_TW, by0 = rc.y0 + (lb / rs.tiles_x[r]) * PF_TH;   // tiles are laid over the output rect
    const int px0 = bx0 - 1, py0 = by0 - 1;
    // one position = the quad (2p-1, 2p) x (2q-1, 2q), weights 1/2 (position p samples u = p / ow: texel coordinate 2p - 1/2).  The four
    // texels of the NEXT position of this thread are loaded before the current one is evaluated (its three IEEE divides): eight loads in
    // flight per thread instead of four (round 4)
    struct Quad { H4 a, b, c, d; };
    auto load_quad = [&](int e) {
        const int r = e / (PF_TW + 2), c = e - r * (PF_TW + 2);
        const int p = px0 + c, q = py0 + r;
        const int x0 = clampi(2 * p - 1, 0, w - 1), x1 = clampi(2 * p, 0, w - 1);
        const int y0 = clampi(2 * q - 1, 0, h - 1), y1 = clampi(2 * q, 0, h - 1);
        const H4* r0 = reinterpret_cast<const H4*>(hdr) + (size_t)y0 * pitch;
        const H4* r1 = reinterpret_cast<const H4*>(hdr) + (size_t)y1 * pitch;
        return Quad{r0[x0], r0[x1], r1[x0], r1[x1]};
    };
    constexpr int NPOS = (PF_TH + 2) * (PF_TW + 2);
    auto evaluate = [&](int e, const Quad& t) {
        const int r = e / (PF_TW + 2), c = e - r * (PF_TW + 2);
        const F4 s = bilerp(h4f(t.a), h4f(t.b), h4f(t.c), h4f(t.d), 0.5f, 0.5f);
        const float brightness = fmaxf(s.x, fmaxf(s.y, s.z));
        float soft = fminf(fmaxf(brightness - threshold + threshold * knee, 0.0f), 2.0f * threshold * knee);
        soft /= 4.0f * threshold * knee + 0.00001f;
        const float contribution = fmaxf(soft, brightness - threshold) / fmaxf(brightness, 0.00001f);
        const float cr = s.x * contribution, cg = s.y * contribution, cb = s.z * contribution;
        const float wgt = 1.0f / (luminance(cr, cg, cb) + 1.0f);
        pos[r][c] = make_float4(cr * wgt, cg * wgt, cb * wgt, wgt);
    };
    // two register sets in turn (a copy `cur = nxt` would wait for the next quad's data at the end of every trip)
    Quad qa = load_quad(tid);   // tid < 256 < NPOS
    for (int e = tid; e < NPOS; e += 512) {
        const Quad qb = load_quad(min(e + 256, NPOS - 1));
        evaluate(e, qa);
        if (e + 256 < NPOS) {
            qa = load_quad(min(e + 512, NPOS - 1));
            evaluate(e + 256, qb);
        }
    }
    __syncthreads();
    const int lx = tid & 63, x = bx0 + lx;
    if (x >= rc.x1) return;
#pragma unroll
    for (int k = 0; k < PF_TH / 4; k++) {
        const int ly = (tid >> 6) + 4 * k, y = by0 + ly;
        if (y >= rc.y1) break;
        // the shader's order: centre, (-1,-1), (-1,+1), (+1,-1), (+1,+1)
        const float4 e0 = pos[ly + 1][lx + 1], e1 = pos[ly][lx], e2 = pos[ly + 2][lx], e3 = pos[ly][lx + 2], e4 = pos[ly + 2][lx + 2];
        float tr = (((e0.x + e1.x) + e2.x) + e3.x) + e4.x;
        float tg = (((e0.y + e1.y) + e2.y) + e3.y) + e4.y;
        float tb = (((e0.z + e1.z) + e2.z) + e3.z) + e4.z;
        const float tw = (((e0.w + e1.w) + e2.w) + e3.w) + e4.w;
        if (tw > 0.0f) { tr /= tw; tg /= tw; tb /= tw; }
        store_h4(out + 4 * ((size_t)(y + rc.oy) * rc.pitch + (x + rc.ox)), f4(tr, tg, tb, 1.0f));
    }
}

enum { M_SAME = 0, M_DOWN = 1, M_UP = 2 };
// source taps of integer output position p along one axis: clamped indices and the weight of the second tap
template <int MODE>
__device__ __forceinline__ void tap1d(int p, int in_size, int& i0, int& i1, float& f) {
    int a;
    if (MODE == M_SAME) { a = p; f = 0.0f; }                       // the texel itself
    else if (MODE == M_DOWN) { a = 2 * p; f = 0.5f; }             // (2p, 2p+1)
    else { a = (p >> 1) - 1 + (p & 1); f = (p & 1) ? 0.25f : 0.75f; }   // p = 2k: (k-1, k) 3/4; p = 2k+1: (k, k+1) 1/4
    i0 = clampi(a, 0, in_size - 1);
    i1 = clampi(a + 1, 0, in_size - 1);
}
struct Tap2 { H4 c00, c10, c01, c11; };
template <int MODE>
__device__ __forceinline__ Tap2 load_tap2(const pbr_half* __restrict__ in, int iw, int x0, int x1, int y0, int y1) {
    const H4* r0 = reinterpret_cast<const H4*>(in) + (size_t)y0 * iw;
    Tap2 t;
    t.c00 = r0[x0];
    if (MODE != M_SAME) {
        const H4* r1 = reinterpret_cast<const H4*>(in) + (size_t)y1 * iw;
        t.c10 = r0[x1]; t.c01 = r1[x0]; t.c11 = r1[x1];
    }
    return t;
}
template <int MODE>
__device__ __forceinline__ float4 finish_tap2(const Tap2& t, float fx, float fy) {
    if (MODE == M_SAME) return to4(h4f(t.c00));
    // weights are 1/4, 1/2 or 3/4 here — never 0, so lerp4's zero-weight select is dead: same values, fewer instructions
    const float wx0 = 1.0f - fx, wy0 = 1.0f - fy;
    const F4 top = fma4(h4f(t.c10), fx, h4f(t.c00) * wx0);
    const F4 bot = fma4(h4f(t.c11), fx, h4f(t.c01) * wx0);
    return to4(fma4(bot, fy, top * wy0));
}

typedef _Float16 half4v __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float float2v __attribute__((ext_vector_type(2)));
// two to_half_rn in one v_cvt_pk_f16_f32 (same rounding); the halves are read back out of the packed register
__device__ __forceinline__ half2v round_h2(float a, float b) {
    asm volatile("" : "+v"(a), "+v"(b));
    float2v f; f.x = a; f.y = b;
    return __builtin_convertvector(f, half2v);
}
__device__ __forceinline__ half4v round_h4(F4 v) {
    half4v r;
    r.x = to_half_rn(v.x); r.y = to_half_rn(v.y); r.z = to_half_rn(v.z); r.w = to_half_rn(v.w);
    return r;
}

// rgb-only variants: inside pbr_bloom the alpha of every chain level is one constant per level (the prefilter writes
// 1, and every later pass filters a constant field with clamp addressing), so the fused kernels filter three
// channels per pixel and push the constant through the same fp32 operations once per thread.
__device__ __forceinline__ V3 h3f(H4 h) { return v3((float)h.x, (float)h.y, (float)h.z); }
__device__ __forceinline__ V3 fma3(V3 a, float s, V3 b) { return v3(__builtin_fmaf(a.x, s, b.x), __builtin_fmaf(a.y, s, b.y), __builtin_fmaf(a.z, s, b.z)); }
template <int MODE>
__device__ __forceinline__ float4 finish_tap2_rgb(const Tap2& t, float fx, float fy) {
    if (MODE == M_SAME) return make_float4((float)t.c00.x, (float)t.c00.y, (float)t.c00.z, 0.0f);
    const float wx0 = 1.0f - fx, wy0 = 1.0f - fy;   // weights 1/4, 1/2, 3/4: never 0 (see finish_tap2)
    const V3 top = fma3(h3f(t.c10), fx, h3f(t.c00) * wx0);
    const V3 bot = fma3(h3f(t.c11), fx, h3f(t.c01) * wx0);
    const V3 r = fma3(bot, fy, top * wy0);
    return make_float4(r.x, r.y, r.z, 0.0f);
}
__device__ __forceinline__ V3 gauss9_rgb(const float4* c) {
    V3 v = v3(0.0f, 0.0f, 0.0f);
#pragma unroll
    for (int i = 0; i < 9; i++) { const float4 e = c[i]; v = fma3(v3(e.x, e.y, e.z), c_gauss[i], v); }
    return v;
}
__device__ __forceinline__ float gauss9_const(float c) {   // the nine fused mads of gauss9 on a constant field
    float v = 0.0f;
#pragma unroll
    for (int i = 0; i < 9; i++) v = __builtin_fmaf(c, c_gauss[i], v);
    return v;
}

// H pass (+ optional second, same-size input: bloom_upsample_add) + V pass [+ merge + histogram] of one level.
// One block = one 64 x TH tile of outputs:
//   1+2. every wave owns (TH+8)/NW of the TH+8 rows the V pass will tap (rows outside the image repeat the edge).
//        It samples its rows' 72 positions (64 columns + the 4+4 halo; all global loads issued up front), then row
//        by row writes the samples to a wave-private LDS line (the H pass's Cache[]), H-gausses it and rounds to
//        fp16 exactly where the H pass stores — no block barrier: LDS operations of one wave execute in order;
//   3.   after the only barrier, V-gauss down each column from the shared fp16 tile, then store / merge into the
//        HDR buffer / histogram.
// TAIL instances: which part of the ow x oh level is merged, and where the HDR buffer sits inside it.
//   tiles  : the 64 x TH tiles (tx0 + i, ty0 + j), i < tiles_x, are walked (the whole level: tx0 = ty0 = 0);
//   merge  : only HDR texels inside [mx0,mx1) x [my0,my1) are read and updated;
//   buffer : level texel (x, y) lives at hdr[(y - by) * pitch + (x - bx)] (a whole-level buffer: bx = by = 0);
//   hist   : texels inside [hx0,hx1) x [hy0,hy1) are counted (TAIL 2).
// A single-GPU frame merges everything; a multi-GPU tile in halo mode merges (and counts) only its interior, which
// is all its HDR buffer covers beyond a 4-pixel rim.
struct TailRect { int tx0, ty0, mx0, my0, mx1, my1, bx, by, hx0, hy0, hx1, hy1; };

template <int MODE, bool DUAL, int TAIL, int TH, int NT>   // TAIL 0: store; 1: merge into hdr; 2: merge + histogram
__global__ __launch_bounds__(NT, 4) void k_blur_hv(const pbr_half* __restrict__ in, int iw, int ih,
                                                 const pbr_half* __restrict__ in2,   // DUAL: ow x oh, same-size
                                                 pbr_half* __restrict__ out, int ow, int oh, int out_pitch,
                                                 int tiles_x, int n_tiles,
                                                 TailRect tr, float min_log, float inv_range,
                                                 uint32_t* __restrict__ hist) {
    constexpr int TW = 64, SW = TW + 8, SR = TH + 8;
    constexpr int NW = NT / 64;
    constexpr int PER_T = SR / NW;                      // sampled / H-gaussed rows per wave: rows wv*PER_T .. +PER_T-1
    constexpr int PER_O = TH / NW;                      // final outputs per thread (rows wv, wv + NW, ...)
    static_assert(SR % NW == 0 && TH % NW == 0 && PER_T * 8 <= 64, "rows must split evenly over the waves; one halo tap per lane");
    __shared__ float4 sLine[DUAL ? 2 : 1][NW][SW];
    __shared__ H4 sT[SR][TW];
    __shared__ uint32_t sh_hist[TAIL == 2 ? NW : 1][TAIL == 2 ? PBR_HISTOGRAM_BINS : 1];
    const int t = threadIdx.x, lane = t & 63;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);   // wave-uniform: row arithmetic stays on the scalar unit
    if (TAIL == 2) {
        for (int i = t; i < NW * PBR_HISTOGRAM_BINS; i += NT) (&sh_hist[0][0])[i] = 0u;
    }
    // the level's constant alpha through the H pass (+ the second input's), its fp16 store, and the V pass
    // (read at a texel the launch's rectangle depends on: with a rectangle (tiled bloom) the producer of `in` was itself run on a
    //  rectangle and left texel 0 of the level untouched; M_UP: `in` is the coarser level.  No rectangle: texel 0, as ever)
    const size_t a_at = MODE == M_UP ? (size_t)min(tr.my0 >> 1, ih - 1) * iw + min(tr.mx0 >> 1, iw - 1) : (size_t)0;
    float alpha_h = gauss9_const((float)reinterpret_cast<const H4*>(in)[a_at].w);
    if (DUAL) alpha_h = alpha_h + gauss9_const((float)reinterpret_cast<const H4*>(in2)[(size_t)min(tr.my0, oh - 1) * ow + min(tr.mx0, ow - 1)].w);
    const h16 alpha_t = to_half_rn(alpha_h);
    const float alpha_v = gauss9_const((float)alpha_t);
    // 1-D grid over tiles; the histogram instance is launched with fewer blocks than tiles (each walks several) so
    // that the per-block flush of 256 global atomics stays rare
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int x0 = (tr.tx0 + tile % tiles_x) * TW, y0 = (tr.ty0 + tile / tiles_x) * TH;   // (TAIL 0 without a rectangle: tx0 = ty0 = 0)
    const int x = x0 + lane;
    // the HDR texels the merge will need: in flight from the start
    H4 hdr_in[PER_O];
    const bool in_mx = TAIL != 0 && x >= tr.mx0 && x < tr.mx1;   // merge rect lies inside the level: implies x < ow
    if (TAIL != 0) {
#pragma unroll
        for (int k = 0; k < PER_O; k++) {
            const int y = y0 + wv + NW * k;
            if (in_mx && y >= tr.my0 && y < tr.my1) hdr_in[k] = *reinterpret_cast<const H4*>(out + 4 * ((size_t)(y - tr.by) * out_pitch + (x - tr.bx)));
        }
    }
    // positions: columns x0-4+c (c = 0..71), rows clamp(y0-4+r) (r = 0..SR-1).  Main tap of row k: c = lane; the halo
    // columns c = 64..71 of the wave's PER_T rows are ONE extra tap: lane -> (row lane / 8, column 64 + lane % 8).
    const int r0 = wv * PER_T;
    const bool has_halo = lane < PER_T * 8;
    const int hk = lane >> 3, hc = 64 + (lane & 7);
    float4* line = sLine[0][wv];
    float4* line2 = sLine[DUAL ? 1 : 0][wv];
    // lanes of one wave exchange data through `line` without a block barrier: DS operations of a wave execute in
    // order, but the compiler must be told that other lanes' slots are read (per-thread alias analysis would let it
    // hoist the loads above the store)
    auto wave_sync = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    {
        Tap2 htap;
        H4 up[DUAL ? PER_T : 1], hup;   // DUAL: the same-size input of bloom_upsample_add, exact texels
        float fx, hfx, hfy = 0.0f;
        float4 smp[PER_T];              // the wave's PER_T rows of samples at this lane's column (rgb)
        int ax0, ax1, bx0, bx1;
        tap1d<MODE>(x0 - 4 + lane, iw, ax0, ax1, fx);
        tap1d<MODE>(x0 - 4 + hc, iw, bx0, bx1, hfx);
        const int sx = clampi(x0 - 4 + lane, 0, ow - 1), shx = clampi(x0 - 4 + hc, 0, ow - 1);
        const int first = y0 - 4 + r0;   // wave-uniform
        // 2x-up sampling, rows that need no vertical clamping (every wave but those on the image's first / last rows): the
        // PER_T consecutive output rows blend only 4 (5) distinct input rows — output row 2m takes (m-1, m) with weight 3/4,
        // row 2m+1 takes (m, m+1) with 1/4 — and the bilinear sample lerps in x first: hrow[j] = the x-lerp of input row j
        // is computed ONCE and shared by the output rows that use it.  Same operations on the same operands as
        // finish_tap2_rgb, half the loads and converts.  (Index patterns are compile-time per parity of the first row.)
        const bool shared_rows = MODE == M_UP && first >= 1 && first + PER_T - 1 <= oh - 2;
        if (MODE == M_UP && shared_rows) {
            constexpr int NR = PER_T / 2 + 2;   // distinct input rows: 4 for PER_T = 5; 5 (even start) / 4 (odd start) for 6
            const int base = (first >> 1) - 1 + (first & 1);
            V3 hrow[NR];
            const float wx0 = 1.0f - fx;
#pragma unroll
            for (int j = 0; j < NR; j++) {
                const H4* row = reinterpret_cast<const H4*>(in) + (size_t)min(base + j, ih - 1) * iw;
                const H4 c0 = row[ax0], c1 = row[ax1];
                hrow[j] = fma3(h3f(c1), fx, h3f(c0) * wx0);
            }
            auto blend = [&](auto parity) {
                constexpr int P = decltype(parity)::value;
#pragma unroll
                for (int k = 0; k < PER_T; k++) {
                    // output row first + k = 2m (+1): input rows (i0, i0 + 1) relative to base, second-tap weight 3/4 (1/4)
                    constexpr int dummy = 0; (void)dummy;
                    const int odd = (P + k) & 1;
                    const int i0 = (P + k + 1) / 2 - P;          // even start: 0,1,1,2,2,3   odd start: 0,0,1,1,2,2
                    const float fyk = odd ? 0.25f : 0.75f, wy0 = 1.0f - fyk;
                    const V3 r = fma3(hrow[i0 + 1], fyk, hrow[i0] * wy0);
                    smp[k] = make_float4(r.x, r.y, r.z, 0.0f);
                }
            };
            if (first & 1) blend(std::integral_constant<int, 1>{}); else blend(std::integral_constant<int, 0>{});
#pragma unroll
            for (int k = 0; k < PER_T; k++)
                if (DUAL) up[k] = reinterpret_cast<const H4*>(in2)[(size_t)(first + k) * ow + sx];
        } else {
            Tap2 taps[PER_T];
            float fy[PER_T];
#pragma unroll
            for (int k = 0; k < PER_T; k++) {
                const int jj = clampi(first + k, 0, oh - 1);
                int ay0, ay1;
                tap1d<MODE>(jj, ih, ay0, ay1, fy[k]);
                taps[k] = load_tap2<MODE>(in, iw, ax0, ax1, ay0, ay1);
                if (DUAL) up[k] = reinterpret_cast<const H4*>(in2)[(size_t)jj * ow + sx];
            }
#pragma unroll
            for (int k = 0; k < PER_T; k++) smp[k] = finish_tap2_rgb<MODE>(taps[k], fx, fy[k]);
        }
        if (has_halo) {
            const int jj = clampi(first + hk, 0, oh - 1);
            int ay0, ay1;
            tap1d<MODE>(jj, ih, ay0, ay1, hfy);
            htap = load_tap2<MODE>(in, iw, bx0, bx1, ay0, ay1);
            if (DUAL) hup = reinterpret_cast<const H4*>(in2)[(size_t)jj * ow + shx];
        }
#pragma unroll
        for (int k = 0; k < PER_T; k++) {
            line[lane] = smp[k];
            if (DUAL) line2[lane] = make_float4((float)up[k].x, (float)up[k].y, (float)up[k].z, 0.0f);
            if (has_halo && hk == k) {
                line[hc] = finish_tap2_rgb<MODE>(htap, hfx, hfy);
                if (DUAL) line2[hc] = make_float4((float)hup.x, (float)hup.y, (float)hup.z, 0.0f);
            }
            wave_sync();
            V3 g = gauss9_rgb(line + lane);
            if (DUAL) g = g + gauss9_rgb(line2 + lane);   // bloom_upsample_add: lower first, then upper
            wave_sync();
            H4 th;   // the H pass's fp16 store
            th.x = to_half_rn(g.x); th.y = to_half_rn(g.y); th.z = to_half_rn(g.z); th.w = alpha_t;
            sT[r0 + k][lane] = th;
        }
    }
    __syncthreads();
    // ---- phase 3: V-gauss + tail
#pragma unroll
    for (int k = 0; k < PER_O; k++) {
        const int r = wv + NW * k, y = y0 + r;
        if (x >= ow || y >= oh) continue;
        V3 a3 = v3(0.0f, 0.0f, 0.0f);
#pragma unroll
        for (int i = 0; i < 9; i++) a3 = fma3(h3f(sT[r + i][lane]), c_gauss[i], a3);
        const F4 a = f4(a3.x, a3.y, a3.z, alpha_v);
        if (TAIL == 0) {
            store_h4(out + 4 * ((size_t)y * out_pitch + x), a);
        } else {
            if (!(in_mx && y >= tr.my0 && y < tr.my1)) continue;
            const half4v a0 = round_h4(a);   // A0 texel as the separate V pass would have stored it
            const F4 s = h4f(hdr_in[k]);
            H4 o;
            o.x = to_half_rn(s.x + (float)a0.x); o.y = to_half_rn(s.y + (float)a0.y); o.z = to_half_rn(s.z + (float)a0.z); o.w = to_half_rn(s.w + (float)a0.w);
            *reinterpret_cast<H4*>(out + 4 * ((size_t)(y - tr.by) * out_pitch + (x - tr.bx))) = o;
            if (TAIL == 2) {
                if (x >= tr.hx0 && x < tr.hx1 && y >= tr.hy0 && y < tr.hy1)
                    atomicAdd(&sh_hist[wv][luminance_bin_exact((float)o.x, (float)o.y, (float)o.z, min_log, inv_range)], 1u);
            }
        }
    }
    if (tile + (int)gridDim.x < n_tiles) __syncthreads();   // the next tile overwrites sT
    }
    if (TAIL == 2) {
        __syncthreads();
        for (int i = t; i < PBR_HISTOGRAM_BINS; i += NT) {
            uint32_t sum = 0;
#pragma unroll
            for (int w2 = 0; w2 < NW; w2++) sum += sh_hist[w2][i];
            if (sum) atomicAdd(&hist[i], sum);
        }
    }
}

// ---------------------------------------------------------------- 2x-up levels of large images: 128-wide tiles
// The same passes as k_blur_hv<M_UP, ...> (bit-identical results), remapped so that the LDS carries half the traffic
// per output — in k_blur_hv the H pass's nine ds_reads per output and the V pass's nine keep the LDS pipe as busy as
// the VALU (~5K cycles each per tile) and the two serialise:
//   * H pass: a lane owns the column PAIR (2p, 2p+1) of its wave's rows.  The two 2x-up samples of a pair blend the
//     three level texels p-1, p, p+1 (3 loads per two samples instead of 4), a sample line is 68 float2 entries per
//     channel, and the two outputs of a lane tap the five entries lane .. lane+4: 15 ds_read_b64 per two outputs
//     instead of 18 ds_read_b96.
//   * V pass: a thread owns a column and TH/4 CONSECUTIVE rows, so the rows it taps overlap: TH/4 + 8 reads of the
//     fp16 tile for TH/4 outputs instead of nine per output.
// One block (512 threads, 8 waves) = one 128 x TH tile.  Operation order per sample / tap is k_blur_hv's.
// texel `byte_off / 8` of a half4 image: a (wave-uniform or not) base + an UNSIGNED 32-bit byte offset, the form that loads as
// `global_load_dwordx2 v, v_off, s[base]` — a signed index makes every load pay a 64-bit vector add (slow issue class, §4.7).
// Images here are < 4 GiB (8 192^2 half4 = 512 MB).
__device__ __forceinline__ H4 ld_h4(const void* base, uint32_t byte_off) {
    return *reinterpret_cast<const H4*>(reinterpret_cast<const char*>(base) + byte_off);
}
__device__ __forceinline__ void pair_tap_up(const pbr_half* __restrict__ in, int iw, int cm, int c0, int cp, int ay0, int ay1, float fy,
                                            V3& even, V3& odd) {
    const uint32_t ra = (uint32_t)(ay0 * iw) * 8u, rb = (uint32_t)(ay1 * iw) * 8u, om = (uint32_t)cm * 8u, o0 = (uint32_t)c0 * 8u, op = (uint32_t)cp * 8u;
    const V3 am = h3f(ld_h4(in, ra + om)), a0 = h3f(ld_h4(in, ra + o0)), ap = h3f(ld_h4(in, ra + op));
    const V3 bm = h3f(ld_h4(in, rb + om)), b0 = h3f(ld_h4(in, rb + o0)), bp = h3f(ld_h4(in, rb + op));
    const float wy0 = 1.0f - fy;
    // even column 2p: taps (p-1, p), second-tap weight 3/4; odd column 2p+1: (p, p+1), 1/4 (tap1d<M_UP>; finish_tap2_rgb)
    even = fma3(fma3(b0, 0.75f, bm * 0.25f), fy, fma3(a0, 0.75f, am * 0.25f) * wy0);
    odd = fma3(fma3(bp, 0.25f, b0 * 0.75f), fy, fma3(ap, 0.25f, a0 * 0.75f) * wy0);
}
// the nine taps of the two outputs of a lane from the five pair entries e[0..4] of one channel
__device__ __forceinline__ void gauss9_pair(const float2* e, float& ge, float& go) {
    const float s[10] = {e[0].x, e[0].y, e[1].x, e[1].y, e[2].x, e[2].y, e[3].x, e[3].y, e[4].x, e[4].y};
    float a = 0.0f, b = 0.0f;
#pragma unroll
    for (int i = 0; i < 9; i++) { a = __builtin_fmaf(s[i], c_gauss[i], a); b = __builtin_fmaf(s[i + 1], c_gauss[i], b); }
    ge = a; go = b;
}

#ifdef PBR_DEBUG_KNOBS   // ---- k_blur_up_wide: the shader-order (bit-exact) 2x-up kernel of round 3, superseded by k_blur_up_poly.  Only the knobs
// build carries it (PBR_BLOOM_POLY=0): as the A/B partner and as the bit-exact checker of the wide path (tests/test_gpu_parity.py)
#ifdef PBR_BLOOM_TIMING   // experiment only (-DPBR_BLOOM_TIMING=<n>, tools/debug/tail_timing.py): shader-clock stamps of wave 0 of every block's n-th tile
__device__ unsigned long long g_tail_stamp[8 * 4096];
#define TSTAMP(i) do { if (TAIL == 2 && t == 0 && tile == (int)blockIdx.x + PBR_BLOOM_TIMING * (int)gridDim.x && blockIdx.x < 4096) g_tail_stamp[blockIdx.x * 8 + (i)] = __builtin_readcyclecounter(); } while (0)
#define TWAIT_VM() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
extern "C" int pbr_debug_tail_stamps(unsigned long long* host, int n) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_tail_stamp), sizeof(unsigned long long) * n); }
#else
#define TSTAMP(i) do {} while (0)
#define TWAIT_VM() do {} while (0)
#endif
template <bool DUAL, int TAIL, int TH>
__global__ __launch_bounds__(512, TH == 16 ? 6 : 4) void k_blur_up_wide(const pbr_half* __restrict__ in, int iw, int ih,
                                                          const pbr_half* __restrict__ in2,   // DUAL: ow x oh, same-size
                                                          pbr_half* __restrict__ out, int ow, int oh, int out_pitch,
                                                          int tiles_x, int n_tiles,
                                                          TailRect tr, float min_log, float inv_range,
                                                          uint32_t* __restrict__ hist) {
    constexpr int TW = 128, NP = TW / 2 + 4, SR = TH + 8, NT = 512, NW = NT / 64;
    constexpr int PER_T = SR / NW;      // H rows per wave
    constexpr int PER_O = TH / 4;       // V outputs per thread: column t & 127, rows (t >> 7) * PER_O ..
    static_assert(SR % NW == 0 && TH % 4 == 0 && PER_T * 4 <= 64, "rows must split evenly over the waves; one halo pair per lane");
    __shared__ float2 sLine[DUAL ? 2 : 1][NW][3][NP];
    __shared__ H4 sT[SR][TW];
    __shared__ uint32_t sh_hist[TAIL == 2 ? NW : 1][TAIL == 2 ? PBR_HISTOGRAM_BINS : 1];
    const int t = threadIdx.x, lane = t & 63;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int vc = t & 127, vg = __builtin_amdgcn_readfirstlane(t >> 7);
    if (TAIL == 2) {
        for (int i = t; i < NW * PBR_HISTOGRAM_BINS; i += NT) (&sh_hist[0][0])[i] = 0u;
    }
    // (read at a texel the launch's rectangle depends on: see k_blur_hv)
    const size_t a_at = (size_t)min(tr.my0 >> 1, ih - 1) * iw + min(tr.mx0 >> 1, iw - 1);
    float alpha_h = gauss9_const((float)reinterpret_cast<const H4*>(in)[a_at].w);
    if (DUAL) alpha_h = alpha_h + gauss9_const((float)reinterpret_cast<const H4*>(in2)[(size_t)min(tr.my0, oh - 1) * ow + min(tr.mx0, ow - 1)].w);
    const h16 alpha_t = to_half_rn(alpha_h);
    const float alpha_v = gauss9_const((float)alpha_t);
    const float a0w = (float)to_half_rn(alpha_v);   // alpha of the A0 texel the separate V pass would have stored
    auto wave_sync = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int x0 = (tr.tx0 + tile % tiles_x) * TW, y0 = (tr.ty0 + tile / tiles_x) * TH;   // (TAIL 0 without a rectangle: tx0 = ty0 = 0)
    const int xv = x0 + vc, rbase = vg * PER_O;
    TSTAMP(0);
    H4 hdr_in[PER_O];
    const bool in_mx = TAIL != 0 && xv >= tr.mx0 && xv < tr.mx1;
    // this thread's HDR texels (dereferenced only inside the merge rect): row k of the thread = a wave-uniform base + the lane's
    // unsigned byte offset
    char* const hdr_row0 = reinterpret_cast<char*>(out) + (ptrdiff_t)(y0 + rbase - tr.by) * out_pitch * 8;
    const uint32_t hdr_x = (uint32_t)(xv - tr.bx) * 8u;
    const size_t hdr_pitch = (size_t)out_pitch * 8u;
    // sample positions: columns x0-4 .. x0+131 = pair entries 0..67 (entry q = level texel pair index x0/2 - 2 + q),
    // rows clamp(y0-4+r).  Main entry of a row: q = lane; the four halo entries 64..67 of the wave's PER_T rows are one
    // extra tap: lane -> (row lane / 4, entry 64 + lane % 4).
    const int r0 = wv * PER_T, first = y0 - 4 + r0;
    const int pp = (x0 >> 1) - 2 + lane;
    const int cm = clampi(pp - 1, 0, iw - 1), c0 = clampi(pp, 0, iw - 1), cp = clampi(pp + 1, 0, iw - 1);
    const bool has_halo = lane < PER_T * 4;
    const int hk = lane >> 2, hq = 64 + (lane & 3);
    {
        V3 sE[PER_T], sO[PER_T], hE = v3(0.0f, 0.0f, 0.0f), hO = hE;
        H4 upE[DUAL ? PER_T : 1], upO[DUAL ? PER_T : 1], hupE{}, hupO{};
        const int sxe = clampi(x0 - 4 + 2 * lane, 0, ow - 1), sxo = clampi(x0 - 3 + 2 * lane, 0, ow - 1);
        const bool shared_rows = first >= 1 && first + PER_T - 1 <= oh - 2;   // see k_blur_hv
        if (shared_rows) {
            constexpr int NR = PER_T / 2 + 2;
            const int base = (first >> 1) - 1 + (first & 1);
            V3 hrE[NR], hrO[NR];
            const uint32_t om = (uint32_t)cm * 8u, o0 = (uint32_t)c0 * 8u, op = (uint32_t)cp * 8u;
#pragma unroll
            for (int j = 0; j < NR; j++) {
                const char* row = reinterpret_cast<const char*>(in) + (size_t)min(base + j, ih - 1) * iw * 8u;   // wave-uniform: a scalar base
                const V3 tm = h3f(ld_h4(row, om)), t0 = h3f(ld_h4(row, o0)), tp = h3f(ld_h4(row, op));
                hrE[j] = fma3(t0, 0.75f, tm * 0.25f);
                hrO[j] = fma3(tp, 0.25f, t0 * 0.75f);
            }
            auto blend = [&](auto parity) {
                constexpr int P = decltype(parity)::value;
#pragma unroll
                for (int k = 0; k < PER_T; k++) {
                    const int odd = (P + k) & 1;
                    const int i0 = (P + k + 1) / 2 - P;
                    const float fyk = odd ? 0.25f : 0.75f, wy0 = 1.0f - fyk;
                    sE[k] = fma3(hrE[i0 + 1], fyk, hrE[i0] * wy0);
                    sO[k] = fma3(hrO[i0 + 1], fyk, hrO[i0] * wy0);
                }
            };
            if (first & 1) blend(std::integral_constant<int, 1>{}); else blend(std::integral_constant<int, 0>{});
        } else {
#pragma unroll
            for (int k = 0; k < PER_T; k++) {
                const int jj = clampi(first + k, 0, oh - 1);
                int ay0, ay1; float fy;
                tap1d<M_UP>(jj, ih, ay0, ay1, fy);
                pair_tap_up(in, iw, cm, c0, cp, ay0, ay1, fy, sE[k], sO[k]);
            }
        }
        if (DUAL) {
#pragma unroll
            for (int k = 0; k < PER_T; k++) {
                const char* row = reinterpret_cast<const char*>(in2) + (size_t)clampi(first + k, 0, oh - 1) * ow * 8u;   // wave-uniform
                upE[k] = ld_h4(row, (uint32_t)sxe * 8u); upO[k] = ld_h4(row, (uint32_t)sxo * 8u);
            }
        }
        if (has_halo) {
            const int jj = clampi(first + hk, 0, oh - 1);
            const int hp = (x0 >> 1) - 2 + hq;
            int ay0, ay1; float fy;
            tap1d<M_UP>(jj, ih, ay0, ay1, fy);
            pair_tap_up(in, iw, clampi(hp - 1, 0, iw - 1), clampi(hp, 0, iw - 1), clampi(hp + 1, 0, iw - 1), ay0, ay1, fy, hE, hO);
            if (DUAL) {
                const uint32_t row = (uint32_t)(jj * ow) * 8u;
                hupE = ld_h4(in2, row + (uint32_t)clampi(x0 - 4 + 2 * hq, 0, ow - 1) * 8u); hupO = ld_h4(in2, row + (uint32_t)clampi(x0 - 3 + 2 * hq, 0, ow - 1) * 8u);
            }
        }
        // the HDR texels of the merge: issued AFTER the level's texels (loads return in order, and the samples below wait
        // for the level's only), consumed after the H pass
        if (TAIL != 0) {
#pragma unroll
            for (int k = 0; k < PER_O; k++) {
                const int y = y0 + rbase + k;
                if (in_mx && y >= tr.my0 && y < tr.my1) hdr_in[k] = ld_h4(hdr_row0 + k * hdr_pitch, hdr_x);
            }
        }
        TSTAMP(1);
        TWAIT_VM();
        TSTAMP(2);
        float2 (*line)[NP] = sLine[0][wv];
        float2 (*line2)[NP] = sLine[DUAL ? 1 : 0][wv];
#pragma unroll
        for (int k = 0; k < PER_T; k++) {
            line[0][lane] = make_float2(sE[k].x, sO[k].x); line[1][lane] = make_float2(sE[k].y, sO[k].y); line[2][lane] = make_float2(sE[k].z, sO[k].z);
            if (DUAL) {
                line2[0][lane] = make_float2((float)upE[k].x, (float)upO[k].x); line2[1][lane] = make_float2((float)upE[k].y, (float)upO[k].y);
                line2[2][lane] = make_float2((float)upE[k].z, (float)upO[k].z);
            }
            if (has_halo && hk == k) {
                line[0][hq] = make_float2(hE.x, hO.x); line[1][hq] = make_float2(hE.y, hO.y); line[2][hq] = make_float2(hE.z, hO.z);
                if (DUAL) {
                    line2[0][hq] = make_float2((float)hupE.x, (float)hupO.x); line2[1][hq] = make_float2((float)hupE.y, (float)hupO.y);
                    line2[2][hq] = make_float2((float)hupE.z, (float)hupO.z);
                }
            }
            wave_sync();
            V3 gE, gO;
            gauss9_pair(line[0] + lane, gE.x, gO.x); gauss9_pair(line[1] + lane, gE.y, gO.y); gauss9_pair(line[2] + lane, gE.z, gO.z);
            if (DUAL) {   // bloom_upsample_add: lower first, then upper
                V3 uE, uO;
                gauss9_pair(line2[0] + lane, uE.x, uO.x); gauss9_pair(line2[1] + lane, uE.y, uO.y); gauss9_pair(line2[2] + lane, uE.z, uO.z);
                gE = gE + uE; gO = gO + uO;
            }
            wave_sync();
            struct alignas(16) H8 { H4 a, b; } th;   // the H pass's fp16 store, both columns of the lane
            th.a.x = to_half_rn(gE.x); th.a.y = to_half_rn(gE.y); th.a.z = to_half_rn(gE.z); th.a.w = alpha_t;
            th.b.x = to_half_rn(gO.x); th.b.y = to_half_rn(gO.y); th.b.z = to_half_rn(gO.z); th.b.w = alpha_t;
            *reinterpret_cast<H8*>(&sT[r0 + k][2 * lane]) = th;
        }
    }
    TSTAMP(3);
    __syncthreads();
    TSTAMP(4);
    // ---- V-gauss over a sliding window of the fp16 tile + tail
    {
        // (the compiler folds the window's fp16 -> fp32 converts into one v_fma_mix_f32 per tap; converting the window once and
        //  tapping with plain FMAs — 4.5 + n x 2.7 against n x 4.6 issue cycles, DESIGN 4.7 — was measured: 216 fewer slow-class
        //  instructions per thread and tile, the launch's time unchanged, profiles/r03_n_bloom_tail_experiments.md)
        H4 win[PER_O + 8];
#pragma unroll
        for (int i = 0; i < PER_O + 8; i++) win[i] = sT[rbase + i][vc];
#pragma unroll
        for (int k = 0; k < PER_O; k++) {
            const int y = y0 + rbase + k;
            if (xv >= ow || y >= oh) continue;
            V3 a3 = v3(0.0f, 0.0f, 0.0f);
#pragma unroll
            for (int i = 0; i < 9; i++) a3 = fma3(h3f(win[k + i]), c_gauss[i], a3);
            const F4 a = f4(a3.x, a3.y, a3.z, alpha_v);
            if (TAIL == 0) {
                store_h4(out + 4 * ((size_t)y * out_pitch + xv), a);
            } else {
                if (!(in_mx && y >= tr.my0 && y < tr.my1)) continue;
                // A0 texel as the separate V pass would have stored it (fp16), then the merge's fp16 sum
                const half2v a01 = round_h2(a3.x, a3.y);
                const h16 a2 = to_half_rn(a3.z);
                const F4 s = h4f(hdr_in[k]);
                struct alignas(8) O4 { half2v lo, hi; } o;
                o.lo = round_h2(s.x + (float)a01.x, s.y + (float)a01.y);
                o.hi = round_h2(s.z + (float)a2, s.w + a0w);
                *reinterpret_cast<O4*>(hdr_row0 + k * hdr_pitch + hdr_x) = o;
                if (TAIL == 2) {   // plain per-lane LDS atomics: wave-aggregating equal bins first (exposure.hip) costs more VALU here than it saves
                    if (xv >= tr.hx0 && xv < tr.hx1 && y >= tr.hy0 && y < tr.hy1)
                        atomicAdd(&sh_hist[wv][luminance_bin_exact((float)o.lo.x, (float)o.lo.y, (float)o.hi.x, min_log, inv_range)], 1u);
                }
            }
        }
    }
    TSTAMP(5);
    if (tile + (int)gridDim.x < n_tiles) __syncthreads();   // the next tile overwrites sT
    TSTAMP(6);
    }
    if (TAIL == 2) {
        __syncthreads();
        for (int i = t; i < PBR_HISTOGRAM_BINS; i += NT) {
            uint32_t sum = 0;
#pragma unroll
            for (int w2 = 0; w2 < NW; w2++) sum += sh_hist[w2][i];
            if (sum) atomicAdd(&hist[i], sum);
        }
    }
}
#endif   // PBR_DEBUG_KNOBS (k_blur_up_wide)

// ---------------------------------------------------------------- 2x-up levels, POLYPHASE form (round 4)
// k_blur_up_wide evaluates what the shader evaluates: every one of the nine taps of a fine-grid output is a bilinear 2x-up
// sample of the coarse level (weights 1/4 | 3/4 in x and in y), 9 taps x (TH + 8) fine rows.  The upsample and the blur are both
// linear and the upsample's weights are periodic, so the H blur of the upsampled row is two FIXED six-tap filters on the COARSE
// row — even outputs 2p tap c[p-3 .. p+2], odd outputs 2p+1 tap c[p-2 .. p+3] (coefficients below) — and the y half of the
// bilinear sample commutes with the H blur: filter the ~TH/2 + 6 coarse rows once, THEN blend neighbouring filtered rows with
// 1/4 | 3/4 into the fine rows.  Per fine output and channel that is ~6 x 22/40 + 2 multiply-adds for the H pass instead of
// 9 + the four of the bilinear sample, a lane loads ONE coarse texel per coarse row (k_blur_up_wide: three per row for its
// column pair) and reads seven 8-byte LDS entries per coarse row instead of fifteen per fine row.
// Not the shader's operation order, hence not bit-identical to the oracle: the fp32 value in front of the H pass's fp16 store
// differs by a few fp32 ulps, i.e. the stored fp16 texel differs by one fp16 ULP on ~5e-5 of the texels (SURVEY 8c allows <= 1
// fp16 ULP per bloom stage; tests/test_gpu_parity.py holds every stage to that and the whole chain to <= 2).  The sum is rounded to
// fp16 exactly where bloom_upsample_add / blur_horizontal store it, the V pass and the tail are k_blur_up_wide's, unchanged.
// Clamp addressing: a tap outside the coarse level reads the edge texel — the same linear map as the shader's clamp of the
// sample position (every fine position outside the level samples the pure edge texel either way).
namespace poly {
constexpr double G[9] = {0.0148, 0.0459, 0.1050, 0.1941, 0.2803, 0.1941, 0.1050, 0.0459, 0.0148};   // blur.hlsli:17
constexpr double g(int k) { return (k < -4 || k > 4) ? 0.0 : G[k + 4]; }
// coefficient of c[p + j] in the EVEN output 2p: sum_k g(k) * [weight of c[p + j] in the 2x-up sample at fine position 2p + k];
// the odd output 2p + 1 takes c[p + j] with E(-j)   (j = -3 .. 2 resp. -2 .. 3; both sets sum to 0.9999 like the nine weights)
constexpr float E(int j) { return (float)(0.75 * g(2 * j) + 0.25 * g(2 * j + 2) + 0.75 * g(2 * j + 1) + 0.25 * g(2 * j - 1)); }
}  // namespace poly

template <bool DUAL, int TAIL, int TH>
__global__ __launch_bounds__(512, 4) void k_blur_up_poly(const pbr_half* __restrict__ in, int iw, int ih,
                                                          const pbr_half* __restrict__ in2,   // DUAL: ow x oh, same-size
                                                          pbr_half* __restrict__ out, int ow, int oh, int out_pitch,
                                                          int tiles_x, int n_tiles,
                                                          TailRect tr, float min_log, float inv_range,
                                                          uint32_t* __restrict__ hist) {
    constexpr int TW = 128, NP = TW / 2 + 4, NC = TW / 2 + 6, SR = TH + 8, NT = 512, NW = NT / 64;
    constexpr int NPAIR = TH / 2 + 5;                 // coarse row pairs (m, m + 1) whose two blends (fine rows 2m + 1, 2m + 2) the tile's SR rows need
    constexpr int PPW = (NPAIR + NW - 1) / NW;        // pairs per wave; wave w: pairs w * PPW .. (the last waves may hold fewer, or none)
    constexpr int PER_O = TH / 4;                     // V outputs per thread: column t & 127, rows (t >> 7) * PER_O ..
    static_assert(TH % 4 == 0 && (PPW + 1) * 6 <= 64 && 2 * PPW * 4 <= 64, "one halo entry per lane");
    __shared__ H4 sLineC[NW][NC + 2];                 // one coarse row per wave at a time: entry e = coarse column x0 / 2 - 3 + e
    __shared__ float2 sLineU[DUAL ? NW : 1][3][NP];   // DUAL: the same-size input's fine row as column pairs (k_blur_up_wide's line)
    __shared__ H4 sT[SR][TW];
    __shared__ uint32_t sh_hist[TAIL == 2 ? NW : 1][TAIL == 2 ? PBR_HISTOGRAM_BINS : 1];
    // two blocks (2 x 8 waves = the 4 waves per SIMD of the launch bounds) must fit a CU's 160 KiB of LDS: an instantiation that does not
    // would still build — gfx950 allows one block the whole 160 KiB — and silently run at half the occupancy
    static_assert(sizeof(sLineC) + sizeof(sLineU) + sizeof(sT) + sizeof(sh_hist) <= 80 * 1024, "k_blur_up_poly: two blocks per CU no longer fit the LDS");
    const int t = threadIdx.x, lane = t & 63;
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    const int vc = t & 127, vg = __builtin_amdgcn_readfirstlane(t >> 7);
    if (TAIL == 2) {
        for (int i = t; i < NW * PBR_HISTOGRAM_BINS; i += NT) (&sh_hist[0][0])[i] = 0u;
    }
    // (read at a texel the launch's rectangle depends on: see k_blur_hv)
    const size_t a_at = (size_t)min(tr.my0 >> 1, ih - 1) * iw + min(tr.mx0 >> 1, iw - 1);
    float alpha_h = gauss9_const((float)reinterpret_cast<const H4*>(in)[a_at].w);
    if (DUAL) alpha_h = alpha_h + gauss9_const((float)reinterpret_cast<const H4*>(in2)[(size_t)min(tr.my0, oh - 1) * ow + min(tr.mx0, ow - 1)].w);
    const h16 alpha_t = to_half_rn(alpha_h);
    const float alpha_v = gauss9_const((float)alpha_t);
    const float a0w = (float)to_half_rn(alpha_v);
    auto wave_sync = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int x0 = (tr.tx0 + tile % tiles_x) * TW, y0 = (tr.ty0 + tile / tiles_x) * TH;   // (TAIL 0 without a rectangle: tx0 = ty0 = 0)
    const int xv = x0 + vc, rbase = vg * PER_O;
    H4 hdr_in[PER_O];
    const bool in_mx = TAIL != 0 && xv >= tr.mx0 && xv < tr.mx1;
    char* const hdr_row0 = reinterpret_cast<char*>(out) + (ptrdiff_t)(y0 + rbase - tr.by) * out_pitch * 8;
    const uint32_t hdr_x = (uint32_t)(xv - tr.bx) * 8u;
    const size_t hdr_pitch = (size_t)out_pitch * 8u;
    // ---- H pass.  Wave w owns the coarse row pairs i = w * PPW .. + PPW - 1 (i < NPAIR): coarse rows cb + i and cb + i + 1, cb = y0 / 2 - 3;
    // pair i blends into the tile's fp16 rows 2i - 1 (fine row y0 - 5 + 2i, odd) and 2i (even); rows -1 and SR do not exist.
    {
        const int i_first = wv * PPW;
        const int n_pairs = min(PPW, NPAIR - i_first);                        // wave-uniform; <= 0: nothing to do for this wave
        const int cb = (y0 >> 1) - 3 + i_first;                               // first coarse row of the wave
        const uint32_t ccol = (uint32_t)clampi((x0 >> 1) - 3 + lane, 0, iw - 1) * 8u;
        H4 cr[PPW + 1], chalo{};                                              // the wave's coarse rows at this lane's column; one halo entry
        const int hk = lane / 6, hq = 64 + lane % 6;                          // halo: lane -> (row lane / 6, entry 64 + lane % 6), lanes 0 .. 6 (PPW + 1) - 1
        const bool has_halo = lane < 6 * (PPW + 1);
        H4 upE[DUAL ? 2 * PPW : 1], upO[DUAL ? 2 * PPW : 1], hupE{}, hupO{};
        const int uk = lane >> 2, uq = 64 + (lane & 3);                       // DUAL halo: lane -> (fine row lane / 4 of the wave's 2 PPW, pair entry 64 + lane % 4)
        const bool has_uhalo = DUAL && lane < 8 * PPW;
        if (n_pairs > 0) {
#pragma unroll
            for (int k = 0; k <= PPW; k++) {
                const char* row = reinterpret_cast<const char*>(in) + (size_t)clampi(cb + k, 0, ih - 1) * iw * 8u;   // wave-uniform: a scalar base
                cr[k] = ld_h4(row, ccol);
            }
            if (has_halo) {
                const char* row = reinterpret_cast<const char*>(in) + (size_t)clampi(cb + hk, 0, ih - 1) * iw * 8u;
                chalo = ld_h4(row, (uint32_t)clampi((x0 >> 1) - 3 + hq, 0, iw - 1) * 8u);
            }
            if (DUAL) {
                const int sxe = clampi(x0 - 4 + 2 * lane, 0, ow - 1), sxo = clampi(x0 - 3 + 2 * lane, 0, ow - 1);
#pragma unroll
                for (int k = 0; k < 2 * PPW; k++) {   // fine rows of the wave: tile row 2 i_first - 1 + k
                    const char* row = reinterpret_cast<const char*>(in2) + (size_t)clampi(y0 - 4 + 2 * i_first - 1 + k, 0, oh - 1) * ow * 8u;
                    upE[k] = ld_h4(row, (uint32_t)sxe * 8u); upO[k] = ld_h4(row, (uint32_t)sxo * 8u);
                }
                if (has_uhalo) {
                    const uint32_t row = (uint32_t)(clampi(y0 - 4 + 2 * i_first - 1 + uk, 0, oh - 1) * ow) * 8u;
                    hupE = ld_h4(in2, row + (uint32_t)clampi(x0 - 4 + 2 * uq, 0, ow - 1) * 8u); hupO = ld_h4(in2, row + (uint32_t)clampi(x0 - 3 + 2 * uq, 0, ow - 1) * 8u);
                }
            }
        }
        // the HDR texels of the merge: issued AFTER the level's texels (loads return in order), consumed after the H pass
        if (TAIL != 0) {
#pragma unroll
            for (int k = 0; k < PER_O; k++) {
                const int y = y0 + rbase + k;
                if (in_mx && y >= tr.my0 && y < tr.my1) hdr_in[k] = ld_h4(hdr_row0 + k * hdr_pitch, hdr_x);
            }
        }
        if (n_pairs > 0) {
            H4* line = sLineC[wv];
            float2 (*lineU)[NP] = sLineU[DUAL ? wv : 0];
            V3 pE = v3(0.0f, 0.0f, 0.0f), pO = pE;   // the previous coarse row, H-filtered: even / odd fine column of the lane
#pragma unroll
            for (int k = 0; k <= PPW; k++) {
                if (k > n_pairs) break;              // wave-uniform
                line[lane] = cr[k];
                if (has_halo && hk == k) line[hq] = chalo;
                wave_sync();
                V3 e[7];
#pragma unroll
                for (int j = 0; j < 7; j++) e[j] = h3f(line[lane + j]);
                wave_sync();
                // even output 2p: c[p-3 .. p+2] = e[0 .. 5] with E(-3 .. 2); odd output 2p + 1: c[p-2 .. p+3] = e[1 .. 6] with E(2 .. -3)
                constexpr float PE[6] = {poly::E(-3), poly::E(-2), poly::E(-1), poly::E(0), poly::E(1), poly::E(2)};
                V3 cE = v3(0.0f, 0.0f, 0.0f), cO = cE;
#pragma unroll
                for (int j = 0; j < 6; j++) { cE = fma3(e[j], PE[j], cE); cO = fma3(e[j + 1], PE[5 - j], cO); }
                if (k > 0) {
#pragma unroll
                    for (int half = 0; half < 2; half++) {
                        const int r = 2 * (i_first + k - 1) - 1 + half;      // tile row of this blend (wave-uniform)
                        if (r < 0 || r >= SR) continue;
                        // fine row 2m + 1: (m, m + 1) with second-tap weight 1/4; fine row 2m + 2: 3/4 (tap1d<M_UP>; the sampler's lerp form)
                        const float fy = half ? 0.75f : 0.25f, wy0 = 1.0f - fy;
                        V3 bE = fma3(cE, fy, pE * wy0), bO = fma3(cO, fy, pO * wy0);
                        if (DUAL) {   // bloom_upsample_add: lower first, then upper — the same-size input's nine taps on this fine row
                            const int u = 2 * (k - 1) + half;
                            lineU[0][lane] = make_float2((float)upE[u].x, (float)upO[u].x); lineU[1][lane] = make_float2((float)upE[u].y, (float)upO[u].y);
                            lineU[2][lane] = make_float2((float)upE[u].z, (float)upO[u].z);
                            if (has_uhalo && uk == u) {
                                lineU[0][uq] = make_float2((float)hupE.x, (float)hupO.x); lineU[1][uq] = make_float2((float)hupE.y, (float)hupO.y);
                                lineU[2][uq] = make_float2((float)hupE.z, (float)hupO.z);
                            }
                            wave_sync();
                            V3 uE, uO;
                            gauss9_pair(lineU[0] + lane, uE.x, uO.x); gauss9_pair(lineU[1] + lane, uE.y, uO.y); gauss9_pair(lineU[2] + lane, uE.z, uO.z);
                            wave_sync();
                            bE = bE + uE; bO = bO + uO;
                        }
                        struct alignas(16) H8 { H4 a, b; } th;   // the H pass's fp16 store, both columns of the lane
                        th.a.x = to_half_rn(bE.x); th.a.y = to_half_rn(bE.y); th.a.z = to_half_rn(bE.z); th.a.w = alpha_t;
                        th.b.x = to_half_rn(bO.x); th.b.y = to_half_rn(bO.y); th.b.z = to_half_rn(bO.z); th.b.w = alpha_t;
                        *reinterpret_cast<H8*>(&sT[r][2 * lane]) = th;
                    }
                }
                pE = cE; pO = cO;
            }
        }
    }
    __syncthreads();
    // ---- V-gauss over a sliding window of the fp16 tile + tail (k_blur_up_wide's)
    {
        H4 win[PER_O + 8];
#pragma unroll
        for (int i = 0; i < PER_O + 8; i++) win[i] = sT[rbase + i][vc];
#pragma unroll
        for (int k = 0; k < PER_O; k++) {
            const int y = y0 + rbase + k;
            if (xv >= ow || y >= oh) continue;
            V3 a3 = v3(0.0f, 0.0f, 0.0f);
#pragma unroll
            for (int i = 0; i < 9; i++) a3 = fma3(h3f(win[k + i]), c_gauss[i], a3);
            const F4 a = f4(a3.x, a3.y, a3.z, alpha_v);
            if (TAIL == 0) {
                store_h4(out + 4 * ((size_t)y * out_pitch + xv), a);
            } else {
                if (!(in_mx && y >= tr.my0 && y < tr.my1)) continue;
                const half2v a01 = round_h2(a3.x, a3.y);
                const h16 a2 = to_half_rn(a3.z);
                const F4 s = h4f(hdr_in[k]);
                struct alignas(8) O4 { half2v lo, hi; } o;
                o.lo = round_h2(s.x + (float)a01.x, s.y + (float)a01.y);
                o.hi = round_h2(s.z + (float)a2, s.w + a0w);
                *reinterpret_cast<O4*>(hdr_row0 + k * hdr_pitch + hdr_x) = o;
                if (TAIL == 2) {
                    if (xv >= tr.hx0 && xv < tr.hx1 && y >= tr.hy0 && y < tr.hy1)
                        atomicAdd(&sh_hist[wv][luminance_bin_exact((float)o.lo.x, (float)o.lo.y, (float)o.hi.x, min_log, inv_range)], 1u);
                }
            }
        }
    }
    if (tile + (int)gridDim.x < n_tiles) __syncthreads();   // the next tile overwrites sT
    }
    if (TAIL == 2) {
        __syncthreads();
        for (int i = t; i < PBR_HISTOGRAM_BINS; i += NT) {
            uint32_t sum = 0;
#pragma unroll
            for (int w2 = 0; w2 < NW; w2++) sum += sh_hist[w2][i];
            if (sum) atomicAdd(&hist[i], sum);
        }
    }
}

// rows a k_blur_h block pipelines: as many as keep >= ~2048 blocks (8 per CU) in the grid
static int blur_h_rows(uint32_t ow, uint32_t oh) {
    const uint64_t row_blocks = (uint64_t)((ow + 255) / 256) * oh;
    int rows = (int)(row_blocks / 2048);
    return rows < 1 ? 1 : (rows > HB_MAX_ROWS ? HB_MAX_ROWS : rows);
}

// fast-path preconditions: the level below is exactly half, and the size keeps every snapped sample coordinate
// on its dyadic value (coordinate error ~4 * 2^-24 * size must stay below half a 1/256 step)
static bool exact_half(uint32_t n) { return (n & 1u) == 0u && n <= 8192u; }
static bool force_staged() { static const bool v = pbr::knob_set("PBR_BLOOM_STAGED"); return v; }   // A/B switch (knobs build only)

template <int MODE, bool DUAL, int TAIL>
static pbr_status launch_hv(pbr_ctx* ctx, const pbr_half* in, uint32_t iw, uint32_t ih, const pbr_half* in2,
                            pbr_half* out, uint32_t ow, uint32_t oh, uint32_t out_pitch,
                            const uint32_t* rect, float min_log, float inv_range, uint32_t* hist,
                            const uint32_t* merge_rect = nullptr, const uint32_t* buf_origin = nullptr) {
    // rect: histogram rect {x,y,w,h}; merge_rect: HDR texels to merge (default: the whole level); buf_origin: level
    // coordinates of out[0] (default 0,0)
    TailRect tr;
    tr.hx0 = rect ? (int)rect[0] : 0; tr.hy0 = rect ? (int)rect[1] : 0;
    tr.hx1 = rect ? (int)(rect[0] + rect[2]) : 0; tr.hy1 = rect ? (int)(rect[1] + rect[3]) : 0;
    tr.mx0 = merge_rect ? (int)merge_rect[0] : 0; tr.my0 = merge_rect ? (int)merge_rect[1] : 0;
    tr.mx1 = merge_rect ? (int)(merge_rect[0] + merge_rect[2]) : (int)ow; tr.my1 = merge_rect ? (int)(merge_rect[1] + merge_rect[3]) : (int)oh;
    tr.bx = buf_origin ? (int)buf_origin[0] : 0; tr.by = buf_origin ? (int)buf_origin[1] : 0;
    // histogram instance: ~1024 blocks that each walk the same number of tiles (an uneven split leaves the chip
    // half empty for the last round; one block per tile costs 256 contended global atomics per tile)
    static const int hist_blocks = pbr::knob_int("PBR_BLOOM_HIST_BLOCKS", 1024);
    auto even_blocks = [](int n_tiles) { const int per = (n_tiles + hist_blocks - 1) / hist_blocks; return (n_tiles + per - 1) / per; };
    if constexpr (MODE == M_UP) {
        // 2x-up levels big enough to fill the chip with 128 x 32 tiles: the two-columns-per-lane kernel (PBR_BLOOM_WIDE=0|1 forces)
        static const int wide_forced = pbr::knob_int("PBR_BLOOM_WIDE", -1);
#ifdef PBR_DEBUG_KNOBS
        static const int wide_th = pbr::knob_int("PBR_BLOOM_WIDE_TH", 32);   // experiment (knobs build): 128 x 16 tiles at 6 waves per SIMD — 53.4 us against 49.2
#endif
        const int wtx0 = tr.mx0 / 128, wty0 = tr.my0 / 32;
        const int wtiles_x = (tr.mx1 + 127) / 128 - wtx0, wn = wtiles_x * ((tr.my1 + 31) / 32 - wty0);
        if (wide_forced >= 0 ? wide_forced == 1 : (wn >= 400 && !ctx->bloom_shader_order)) {
            tr.tx0 = wtx0; tr.ty0 = wty0;
#ifdef PBR_DEBUG_KNOBS   // the shader-order (bit-identical) kernel of round 3: A/B partner and checker, PBR_BLOOM_POLY=0
            static const bool poly_off = pbr::knob_int("PBR_BLOOM_POLY", 1) == 0;
            if (poly_off && wide_th == 16) {
                tr.ty0 = tr.my0 / 16;
                const int wn16 = wtiles_x * ((tr.my1 + 15) / 16 - tr.ty0);
                hipLaunchKernelGGL((k_blur_up_wide<DUAL, TAIL, 16>), dim3(TAIL == 2 ? even_blocks(wn16) : wn16), dim3(512), 0, ctx->stream,
                                   in, (int)iw, (int)ih, in2, out, (int)ow, (int)oh, (int)out_pitch, wtiles_x, wn16, tr, min_log, inv_range, hist);
                return launched(ctx, "k_blur_up_wide<16>");
            }
            if (poly_off) {
                hipLaunchKernelGGL((k_blur_up_wide<DUAL, TAIL, 32>), dim3(TAIL == 2 ? even_blocks(wn) : wn), dim3(512), 0, ctx->stream,
                                   in, (int)iw, (int)ih, in2, out, (int)ow, (int)oh, (int)out_pitch, wtiles_x, wn, tr, min_log, inv_range, hist);
                return launched(ctx, "k_blur_up_wide");
            }
#endif
            hipLaunchKernelGGL((k_blur_up_poly<DUAL, TAIL, 32>), dim3(TAIL == 2 ? even_blocks(wn) : wn), dim3(512), 0, ctx->stream,
                               in, (int)iw, (int)ih, in2, out, (int)ow, (int)oh, (int)out_pitch, wtiles_x, wn, tr, min_log, inv_range, hist);
            return launched(ctx, "k_blur_up_poly");
        }
    }
    // 64 x 32 tiles (512 threads) when the level is large enough to fill the chip that way, 64 x 16 below (PBR_BLOOM_TILE=16: on 4 waves)
    static const int forced = pbr::knob_int("PBR_BLOOM_TILE", 0);
    const bool big = forced ? forced == 32 : (uint64_t)((tr.mx1 + 63) / 64 - tr.mx0 / 64) * ((tr.my1 + 31) / 32 - tr.my0 / 32) >= 900;
    // tiles that intersect the merge rect (TAIL 0 has no rect: every tile of the level)
    const int th = big ? 32 : 16;
    tr.tx0 = tr.mx0 / 64; tr.ty0 = tr.my0 / th;
    const int tiles_x = (tr.mx1 + 63) / 64 - tr.tx0;
    const int n_tiles = tiles_x * ((tr.my1 + th - 1) / th - tr.ty0);
    const int blocks = TAIL == 2 ? even_blocks(n_tiles) : n_tiles;
    if (big) {
        hipLaunchKernelGGL((k_blur_hv<MODE, DUAL, TAIL, 32, 512>), dim3(blocks), dim3(512), 0, ctx->stream,
                           in, (int)iw, (int)ih, in2, out, (int)ow, (int)oh, (int)out_pitch, tiles_x, n_tiles, tr, min_log, inv_range, hist);
    } else if (forced != 16) {
        // small levels are latency-bound (one tile's dependent chain + the launch): 64 x 16 tiles on EIGHT waves — 3 H rows per
        // wave, 2 outputs per thread — shorten the chain; the five small launches of a 4K frame take ~5 us less together
        hipLaunchKernelGGL((k_blur_hv<MODE, DUAL, TAIL, 16, 512>), dim3(blocks), dim3(512), 0, ctx->stream,
                           in, (int)iw, (int)ih, in2, out, (int)ow, (int)oh, (int)out_pitch, tiles_x, n_tiles, tr, min_log, inv_range, hist);
    } else {
        hipLaunchKernelGGL((k_blur_hv<MODE, DUAL, TAIL, 16, 256>), dim3(blocks), dim3(256), 0, ctx->stream,
                           in, (int)iw, (int)ih, in2, out, (int)ow, (int)oh, (int)out_pitch, tiles_x, n_tiles, tr, min_log, inv_range, hist);
    }
    return launched(ctx, "k_blur_hv");
}

extern "C" {

// rcs: n output rectangles sharing one destination offset / pitch
static pbr_status prefilter_launch(pbr_ctx* ctx, const pbr_half* hdr, uint32_t w, uint32_t h, uint32_t pitch,
                                   pbr_half* out, const OutRect* rcs, int n, float threshold, float knee) {
    const uint32_t ow = w >> 1, oh = h >> 1;
    const float tx = 1.0f / (float)ow, ty = 1.0f / (float)oh;   // DeferredPipeline.cpp:418
    if (exact_half(w) && exact_half(h) && !force_staged()) {   // shared-sample kernel (bit-identical), all rectangles in one launch
        OutRects rs{};
        rs.n = n; rs.ox = rcs[0].ox; rs.oy = rcs[0].oy; rs.pitch = rcs[0].pitch;
        int blocks = 0;
        for (int r = 0; r < n; r++) {
            rs.x0[r] = rcs[r].x0; rs.y0[r] = rcs[r].y0; rs.x1[r] = rcs[r].x1; rs.y1[r] = rcs[r].y1;
            rs.tiles_x[r] = (rcs[r].x1 - rcs[r].x0 + PF_TW - 1) / PF_TW;
            rs.first[r] = blocks;
            blocks += rs.tiles_x[r] * ((rcs[r].y1 - rcs[r].y0 + PF_TH - 1) / PF_TH);
        }
        rs.first[n] = blocks;
        hipLaunchKernelGGL(k_bloom_prefilter_2x, dim3(blocks), dim3(256), 0, ctx->stream, hdr, (int)w, (int)h, (int)pitch, out, rs, threshold, knee);
        return launched(ctx, "k_bloom_prefilter_2x");
    }
    for (int r = 0; r < n; r++) {
        const uint32_t rw = (uint32_t)(rcs[r].x1 - rcs[r].x0), rh = (uint32_t)(rcs[r].y1 - rcs[r].y0);
        dim3 grid((rw + 63) / 64, (rh + 3) / 4);
        hipLaunchKernelGGL(k_bloom_prefilter, grid, dim3(64, 4), 0, ctx->stream, hdr, (int)w, (int)h, (int)pitch, out, rcs[r], tx, ty, threshold, knee);
        pbr_status st = launched(ctx, "k_bloom_prefilter");
        if (st) return st;
    }
    return PBR_OK;
}

pbr_status pbr_bloom_prefilter(pbr_ctx* ctx, const pbr_half* hdr, uint32_t w, uint32_t h, uint32_t pitch,
                               pbr_half* out, float threshold, float knee) {
    if (!ctx) return PBR_ERR_INVALID;
    PBR_REQUIRE(ctx, hdr && out, "pbr_bloom_prefilter: null pointer");
    PBR_REQUIRE(ctx, (w >> 1) >= 1 && (h >> 1) >= 1 && w <= 65535 && h <= 65535 && pitch >= w, "pbr_bloom_prefilter: bad size");
    const OutRect rc{0, 0, (int)(w >> 1), (int)(h >> 1), 0, 0, (int)(w >> 1)};
    return prefilter_launch(ctx, hdr, w, h, pitch, out, &rc, 1, threshold, knee);
}

pbr_status pbr_bloom_prefilter_rect(pbr_ctx* ctx, const pbr_half* hdr, uint32_t w, uint32_t h, uint32_t pitch,
                                    pbr_half* out, uint32_t out_pitch, uint32_t out_x, uint32_t out_y,
                                    const uint32_t rect[4], float threshold, float knee) {
    if (!ctx) return PBR_ERR_INVALID;
    PBR_REQUIRE(ctx, hdr && out && rect, "pbr_bloom_prefilter_rect: null pointer");
    PBR_REQUIRE(ctx, (w >> 1) >= 1 && (h >> 1) >= 1 && w <= 65535 && h <= 65535 && pitch >= w, "pbr_bloom_prefilter_rect: bad size");
    PBR_REQUIRE(ctx, rect[2] >= 1 && rect[3] >= 1 && rect[0] + rect[2] <= (w >> 1) && rect[1] + rect[3] <= (h >> 1), "pbr_bloom_prefilter_rect: rect outside the half-res image");
    PBR_REQUIRE(ctx, out_pitch >= out_x + rect[0] + rect[2] && out_pitch <= 65535 && out_y <= 65535, "pbr_bloom_prefilter_rect: rect does not fit the output pitch");
    const OutRect rc{(int)rect[0], (int)rect[1], (int)(rect[0] + rect[2]), (int)(rect[1] + rect[3]), (int)out_x, (int)out_y, (int)out_pitch};
    return prefilter_launch(ctx, hdr, w, h, pitch, out, &rc, 1, threshold, knee);
}

pbr_status pbr_bloom_prefilter_rects(pbr_ctx* ctx, const pbr_half* hdr, uint32_t w, uint32_t h, uint32_t pitch,
                                     pbr_half* out, uint32_t out_pitch, uint32_t out_x, uint32_t out_y,
                                     const uint32_t (*rects)[4], uint32_t n_rects, float threshold, float knee) {
    if (!ctx) return PBR_ERR_INVALID;
    PBR_REQUIRE(ctx, hdr && out && rects && n_rects >= 1 && n_rects <= (uint32_t)PF_MAX_RECTS, "pbr_bloom_prefilter_rects: null pointer / 1 .. 5 rectangles");
    PBR_REQUIRE(ctx, (w >> 1) >= 1 && (h >> 1) >= 1 && w <= 65535 && h <= 65535 && pitch >= w && out_pitch <= 65535 && out_y <= 65535, "pbr_bloom_prefilter_rects: bad size");
    OutRect rcs[PF_MAX_RECTS];
    for (uint32_t r = 0; r < n_rects; r++) {
        const uint32_t* q = rects[r];
        PBR_REQUIRE(ctx, q[2] >= 1 && q[3] >= 1 && q[0] + q[2] <= (w >> 1) && q[1] + q[3] <= (h >> 1) && out_pitch >= out_x + q[0] + q[2],
                    "pbr_bloom_prefilter_rects: rectangle outside the half-res image / the output pitch");
        rcs[r] = OutRect{(int)q[0], (int)q[1], (int)(q[0] + q[2]), (int)(q[1] + q[3]), (int)out_x, (int)out_y, (int)out_pitch};
    }
    return prefilter_launch(ctx, hdr, w, h, pitch, out, rcs, (int)n_rects, threshold, knee);
}

pbr_status pbr_blur_h(pbr_ctx* ctx, const pbr_half* in, uint32_t iw, uint32_t ih, pbr_half* out, uint32_t ow, uint32_t oh) {
    if (!ctx) return PBR_ERR_INVALID;
    PBR_REQUIRE(ctx, in && out, "pbr_blur_h: null pointer");
    PBR_REQUIRE(ctx, iw && ih && ow && oh && iw <= 65535 && ih <= 65535 && ow <= 65535 && oh <= 65535, "pbr_blur_h: bad size");
    const float tx = 1.0f / (float)ow, ty = 1.0f / (float)oh;
    const int rows = blur_h_rows(ow, oh);
    dim3 grid((ow + 255) / 256, (oh + rows - 1) / rows);
    hipLaunchKernelGGL(k_blur_h<false>, grid, dim3(256), 0, ctx->stream, in, (int)iw, (int)ih, (const pbr_half*)nullptr, 0, 0, out, (int)ow, (int)oh, tx, ty, rows);
    return launched(ctx, "k_blur_h");
}

pbr_status pbr_blur_v(pbr_ctx* ctx, const pbr_half* in, uint32_t iw, uint32_t ih, pbr_half* out, uint32_t ow, uint32_t oh) {
    if (!ctx) return PBR_ERR_INVALID;
    PBR_REQUIRE(ctx, in && out, "pbr_blur_v: null pointer");
    PBR_REQUIRE(ctx, iw && ih && ow && oh && iw <= 65535 && ih <= 65535 && ow <= 65535 && oh <= 65535, "pbr_blur_v: bad size");
    const float tx = 1.0f / (float)ow, ty = 1.0f / (float)oh;
    dim3 grid((ow + VT_W - 1) / VT_W, (oh + VT_R - 1) / VT_R);
    hipLaunchKernelGGL(k_blur_v, grid, dim3(VT_W, 4), 0, ctx->stream, in, (int)iw, (int)ih, out, (int)ow, (int)oh, tx, ty);
    return launched(ctx, "k_blur_v");
}

pbr_status pbr_bloom_upsample_add(pbr_ctx* ctx, const pbr_half* upper, uint32_t uw, uint32_t uh,
                                  const pbr_half* lower, uint32_t lw, uint32_t lh, pbr_half* out) {
    if (!ctx) return PBR_ERR_INVALID;
    PBR_REQUIRE(ctx, upper && lower && out, "pbr_bloom_upsample_add: null pointer");
    PBR_REQUIRE(ctx, uw && uh && lw && lh && uw <= 65535 && uh <= 65535, "pbr_bloom_upsample_add: bad size");
    const float tx = 1.0f / (float)uw, ty = 1.0f / (float)uh;
    const int rows = blur_h_rows(uw, uh);
    dim3 grid((uw + 255) / 256, (uh + rows - 1) / rows);
    hipLaunchKernelGGL(k_blur_h<true>, grid, dim3(256), 0, ctx->stream, lower, (int)lw, (int)lh, upper, (int)uw, (int)uh, out, (int)uw, (int)uh, tx, ty, rows);
    return launched(ctx, "k_blur_h<dual>");
}

pbr_status pbr_bloom_merge(pbr_ctx* ctx, pbr_half* hdr, uint32_t pitch, const pbr_half* in, uint32_t w, uint32_t h) {
    if (!ctx) return PBR_ERR_INVALID;
    PBR_REQUIRE(ctx, hdr && in, "pbr_bloom_merge: null pointer");
    PBR_REQUIRE(ctx, w && h && w <= 65535 && h <= 65535 && pitch >= w, "pbr_bloom_merge: bad size");
    dim3 grid((w + 255) / 256, h);
    hipLaunchKernelGGL(k_bloom_merge, grid, dim3(256), 0, ctx->stream, hdr, (int)pitch, in, (int)w, (int)h);
    return launched(ctx, "k_bloom_merge");
}

pbr_status pbr_bloom_up_level(pbr_ctx* ctx, const pbr_half* upper, const pbr_half* lower, uint32_t lw, uint32_t lh,
                              pbr_half* out, uint32_t ow, uint32_t oh) {
    if (!ctx) return PBR_ERR_INVALID;
    PBR_REQUIRE(ctx, lower && out && out != lower && out != upper, "pbr_bloom_up_level: null pointer / out aliases an input");
    PBR_REQUIRE(ctx, lw >= 1 && lh >= 1 && ow == 2 * lw && oh == 2 * lh && exact_half(ow) && exact_half(oh), "pbr_bloom_up_level: out must be exactly twice lower, even, <= 8192");
    if (force_staged()) return pbr::fail(ctx, PBR_ERR_UNSUPPORTED, "pbr_bloom_up_level: PBR_BLOOM_STAGED is set");
    if (upper) return launch_hv<M_UP, true, 0>(ctx, lower, lw, lh, upper, out, ow, oh, ow, nullptr, 0.0f, 0.0f, nullptr);
    return launch_hv<M_UP, false, 0>(ctx, lower, lw, lh, nullptr, out, ow, oh, ow, nullptr, 0.0f, 0.0f, nullptr);
}

static pbr_status bloom_final(pbr_ctx* ctx, const pbr_half* b0, pbr_half* hdr, uint32_t w, uint32_t h, uint32_t pitch,
                              const uint32_t* hist_rect, float min_log, float inv_range, uint32_t* hist256) {
    const float tx = 1.0f / (float)w, ty = 1.0f / (float)h;
    const int tiles_x = (int)((w + VT_W - 1) / VT_W), tiles_y = (int)((h + VT_R - 1) / VT_R);
    int blocks = tiles_x * tiles_y;
    if (blocks > 1280) blocks = 1280;   // 5 blocks (24.6 + 4 KB LDS each) resident per CU x 256 CUs: one full wave of persistent blocks
    if (hist256) {
        hipLaunchKernelGGL(k_blur_v_merge<true>, dim3(blocks), dim3(VT_W, 4), 0, ctx->stream, b0, (int)w, (int)h, tx, ty, hdr, (int)pitch, tiles_x, tiles_y,
                           (int)hist_rect[0], (int)hist_rect[1], (int)(hist_rect[0] + hist_rect[2]), (int)(hist_rect[1] + hist_rect[3]), min_log, inv_range, hist256);
    } else {
        hipLaunchKernelGGL(k_blur_v_merge<false>, dim3(blocks), dim3(VT_W, 4), 0, ctx->stream, b0, (int)w, (int)h, tx, ty, hdr, (int)pitch, tiles_x, tiles_y,
                           0, 0, 0, 0, 0.0f, 0.0f, (uint32_t*)nullptr);
    }
    return launched(ctx, "k_blur_v_merge");
}

// Levels 1..4 of BloomPass::Execute from a filled level 1 of chain A: the three downsample pairs and the three
// upsample-add pairs (DeferredPipeline.cpp:428-540).  *res = where the finished level 1 lives (chain A or B).
// need0 (optional): the rectangle {x, y, w, h} of level 0 the caller will merge.  The up-pass of level l is then run only on the tiles
// of level l the finished image inside need0 depends on — the rectangle shrinks towards need0 / 2^l as the remaining filter support
// does (a tiled frame's bloom works on the tile +- 256 px, but only the DOWN-pass needs that apron in full: SURVEY 8e's "cheaper
// apron").  Whole tiles are computed, so every texel inside the rectangles is what the full pass computes: the merged interior is
// bit-identical.  Texels of the up-levels outside them are left as they were (the chains are scratch).
static pbr_status bloom_pyramid(pbr_ctx* ctx, uint32_t w, uint32_t h, pbr_half* A, pbr_half* B, const pbr_half** res_out, const uint32_t* need0 = nullptr) {
    auto a = [&](uint32_t l) { return A + 4 * pbr_bloom_level_offset(w, h, l); };
    auto b = [&](uint32_t l) { return B + 4 * pbr_bloom_level_offset(w, h, l); };
    auto W = [&](uint32_t l) { return w >> l; };
    auto H = [&](uint32_t l) { return h >> l; };
    pbr_status r;
    // Per level pair: where level l+1 is exactly half of level l (and both fit the fast path's size limit) the H and V
    // pass run as one kernel and the H result (chain B of the reference schedule) is never written; elsewhere the two
    // staged kernels run.  1920x1080, for instance, is exact down to 240x135 and staged for 135 -> 67.  Fused up-levels
    // write chain B (a block must not overwrite what its neighbours still read), so `res` tracks where the finished
    // level below lives.  Chain contents after the call are scratch.
    auto exact = [&](uint32_t l) { return !force_staged() && exact_half(W(l)) && exact_half(H(l)); };
    for (uint32_t i = 0; i < PBR_BLOOM_STEP; i++) {   // downsample
        const uint32_t up = i + 1, lo = i + 2;
        if (exact(up)) {
            if ((r = launch_hv<M_DOWN, false, 0>(ctx, a(up), W(up), H(up), nullptr, a(lo), W(lo), H(lo), W(lo), nullptr, 0.0f, 0.0f, nullptr))) return r;
        } else {
            if ((r = pbr_blur_h(ctx, a(up), W(up), H(up), b(lo), W(lo), H(lo)))) return r;
            if ((r = pbr_blur_v(ctx, b(lo), W(lo), H(lo), a(lo), W(lo), H(lo)))) return r;
        }
    }
    // rectangles of the up-levels (level coordinates): level 1's result is read by the merge within need0 / 2 +- 3 texels (nine taps one
    // level-0 texel apart + the bilinear footprint); level l's up-pass reads the level below within +- 4 of its own taps, halved, + the
    // bilinear footprint.  Margins are rounded up: a superset costs a tile at most
    uint32_t need[PBR_BLOOM_MIPS][4];
    static const bool shrink = pbr::knob_int("PBR_BLOOM_SHRINK", 1) != 0;
    const bool use_need = need0 != nullptr && shrink;
    if (use_need) {
        int x0 = (int)need0[0], y0 = (int)need0[1], x1 = (int)(need0[0] + need0[2]), y1 = (int)(need0[1] + need0[3]);
        for (uint32_t l = 1; l < PBR_BLOOM_MIPS; l++) {
            x0 = (x0 - 4) / 2 - 2; y0 = (y0 - 4) / 2 - 2; x1 = (x1 + 4 + 1) / 2 + 2; y1 = (y1 + 4 + 1) / 2 + 2;   // (C division of a negative numerator rounds towards 0: clipped below anyway)
            const int cx0 = x0 < 0 ? 0 : x0, cy0 = y0 < 0 ? 0 : y0, cx1 = x1 > (int)W(l) ? (int)W(l) : x1, cy1 = y1 > (int)H(l) ? (int)H(l) : y1;
            need[l][0] = (uint32_t)cx0; need[l][1] = (uint32_t)cy0; need[l][2] = (uint32_t)(cx1 - cx0); need[l][3] = (uint32_t)(cy1 - cy0);
            x0 = cx0; y0 = cy0; x1 = cx1; y1 = cy1;
        }
    }
    const pbr_half* res = a(PBR_BLOOM_MIPS - 1);
    for (int i = PBR_BLOOM_STEP - 1; i >= 0; i--) {   // upsample: V(H(lower) + H(upper))
        const uint32_t up = (uint32_t)i + 1;
        if (exact(up)) {
            if ((r = launch_hv<M_UP, true, 0>(ctx, res, W(up + 1), H(up + 1), a(up), b(up), W(up), H(up), W(up), nullptr, 0.0f, 0.0f, nullptr,
                                              use_need ? need[up] : nullptr))) return r;
            res = b(up);
        } else {
            if ((r = pbr_bloom_upsample_add(ctx, a(up), W(up), H(up), res, W(up + 1), H(up + 1), b(up)))) return r;
            if ((r = pbr_blur_v(ctx, b(up), W(up), H(up), a(up), W(up), H(up)))) return r;
            res = a(up);
        }
    }
    *res_out = res;
    return PBR_OK;
}

static pbr_status bloom_impl(pbr_ctx* ctx, pbr_half* hdr, uint32_t w, uint32_t h, uint32_t pitch, pbr_half* A, pbr_half* B,
                             float threshold, float knee, const uint32_t* hist_rect, float min_log, float inv_range, uint32_t* hist256) {
    PBR_REQUIRE(ctx, hdr && A && B, "pbr_bloom: null pointer");
    // BloomStep < CalculateMaxMipLevels (DeferredPipeline.cpp:343): every level must be >= 1 texel
    PBR_REQUIRE(ctx, (w >> (PBR_BLOOM_MIPS - 1)) >= 1 && (h >> (PBR_BLOOM_MIPS - 1)) >= 1, "pbr_bloom: image too small for 5 mips");
    PBR_REQUIRE(ctx, w <= 65535 && h <= 65535 && pitch >= w, "pbr_bloom: bad size");
    pbr_status r;
    if ((r = pbr_bloom_prefilter(ctx, hdr, w, h, pitch, A + 4 * pbr_bloom_level_offset(w, h, 1), threshold, knee))) return r;
    const pbr_half* res = nullptr;
    if ((r = bloom_pyramid(ctx, w, h, A, B, &res))) return r;
    if (!force_staged() && exact_half(w) && exact_half(h)) {   // H + V + merge (+ histogram) in one kernel
        if (hist256) return launch_hv<M_UP, false, 2>(ctx, res, w >> 1, h >> 1, nullptr, hdr, w, h, pitch, hist_rect, min_log, inv_range, hist256);
        return launch_hv<M_UP, false, 1>(ctx, res, w >> 1, h >> 1, nullptr, hdr, w, h, pitch, nullptr, 0.0f, 0.0f, nullptr);
    }
    pbr_half* b0 = B;   // level 0 of chain B
    if ((r = pbr_blur_h(ctx, res, w >> 1, h >> 1, b0, w, h))) return r;
    // A0 = V(B0); S += A0 [; histogram(S)] in one pass — chain A level 0 is not materialised
    return bloom_final(ctx, b0, hdr, w, h, pitch, hist_rect, min_log, inv_range, hist256);
}

// BloomPass::Execute (DeferredPipeline.cpp:400-570; schedule comment :379-399)
pbr_status pbr_bloom(pbr_ctx* ctx, pbr_half* hdr, uint32_t w, uint32_t h, uint32_t pitch,
                     pbr_half* A, pbr_half* B, float threshold, float knee) {
    if (!ctx) return PBR_ERR_INVALID;
    return bloom_impl(ctx, hdr, w, h, pitch, A, B, threshold, knee, nullptr, 0.0f, 0.0f, nullptr);
}

// BloomPass::Execute followed by the luminance-histogram dispatch of AutoExposurePass::Execute
// (DeferredPipeline.cpp:276-298) on the pixels of `rect` = {x, y, w, h} of the bloomed image.
pbr_status pbr_bloom_histogram(pbr_ctx* ctx, pbr_half* hdr, uint32_t w, uint32_t h, uint32_t pitch,
                               pbr_half* A, pbr_half* B, float threshold, float knee,
                               const uint32_t rect[4], float min_log, float inv_range, uint32_t* hist256) {
    if (!ctx) return PBR_ERR_INVALID;
    PBR_REQUIRE(ctx, rect && hist256, "pbr_bloom_histogram: null pointer");
    PBR_REQUIRE(ctx, rect[2] >= 1 && rect[3] >= 1 && rect[0] + rect[2] <= w && rect[1] + rect[3] <= h, "pbr_bloom_histogram: rect outside the image");
    return bloom_impl(ctx, hdr, w, h, pitch, A, B, threshold, knee, rect, min_log, inv_range, hist256);
}

// Multi-GPU halo path (SURVEY 8e option 2): BloomPass::Execute minus the prefilter, on the extended rectangle E
// (ew x eh) of a tile whose level 1 (chain A) is already filled — the interior by pbr_bloom_prefilter_rect, the
// rest by the neighbours' level 1 — and with the final merge (+ histogram) restricted to merge_rect, the part of E
// this rank owns.  hdr covers hdr_rect = {x, y, w, h} of E only (hdr[0] = texel (x, y), pitch hdr_pitch).
pbr_status pbr_bloom_tiled(pbr_ctx* ctx, pbr_half* hdr, uint32_t hdr_pitch, const uint32_t hdr_rect[4],
                           uint32_t ew, uint32_t eh, pbr_half* A, pbr_half* B, const uint32_t merge_rect[4],
                           float min_log, float inv_range, uint32_t* hist256) {
    if (!ctx) return PBR_ERR_INVALID;
    PBR_REQUIRE(ctx, hdr && hdr_rect && A && B && merge_rect, "pbr_bloom_tiled: null pointer");
    PBR_REQUIRE(ctx, ew <= 65535 && eh <= 65535 && (ew >> (PBR_BLOOM_MIPS - 1)) >= 1 && (eh >> (PBR_BLOOM_MIPS - 1)) >= 1, "pbr_bloom_tiled: bad size");
    PBR_REQUIRE(ctx, hdr_rect[2] >= 1 && hdr_rect[3] >= 1 && hdr_rect[0] + hdr_rect[2] <= ew && hdr_rect[1] + hdr_rect[3] <= eh && hdr_pitch >= hdr_rect[2],
                "pbr_bloom_tiled: hdr_rect outside the extended tile");
    PBR_REQUIRE(ctx, merge_rect[2] >= 1 && merge_rect[3] >= 1 && merge_rect[0] >= hdr_rect[0] && merge_rect[1] >= hdr_rect[1] &&
                     merge_rect[0] + merge_rect[2] <= hdr_rect[0] + hdr_rect[2] && merge_rect[1] + merge_rect[3] <= hdr_rect[1] + hdr_rect[3],
                "pbr_bloom_tiled: merge_rect outside hdr_rect");
    if (force_staged() || !exact_half(ew) || !exact_half(eh))
        return pbr::fail(ctx, PBR_ERR_UNSUPPORTED, "pbr_bloom_tiled: the extended tile must be even and <= 8192 on a side");
    const pbr_half* res = nullptr;
    pbr_status r;
    if ((r = bloom_pyramid(ctx, ew, eh, A, B, &res, merge_rect))) return r;
    const uint32_t origin[2] = {hdr_rect[0], hdr_rect[1]};
    if (hist256) return launch_hv<M_UP, false, 2>(ctx, res, ew >> 1, eh >> 1, nullptr, hdr, ew, eh, hdr_pitch, merge_rect, min_log, inv_range, hist256, merge_rect, origin);
    return launch_hv<M_UP, false, 1>(ctx, res, ew >> 1, eh >> 1, nullptr, hdr, ew, eh, hdr_pitch, nullptr, 0.0f, 0.0f, nullptr, merge_rect, origin);
}

}  // extern "C"
