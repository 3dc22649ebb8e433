// exposure.hip — auto-exposure (256-bin log2-luminance histogram + weighted-bin average) and
// ACES tone-map (hdr_luminance_histogram.hlsl, hdr_average_histogram.hlsl, hdr_tone_mapping.hlsl).
//
// Histogram on MI355X: the reference uses 16x16 groups with 256 LDS atomics and then 256
// global atomics PER GROUP (32 400 groups at 4K = 8.3 M global atomics).  Here a fixed grid of
// persistent blocks streams pixel pairs with 16-byte loads, each WAVE owns a private 256-bin LDS
// histogram (no cross-wave LDS contention), and a block issues at most 256 global atomics in
// total — ~0.3 M for a 4K frame.  Integer sums are order-independent, so the result is
// bit-identical to the reference's.
#include "pbr_internal.hpp"
#include "pbr_device.hpp"

using namespace pbr;

// hdr_luminance_histogram.hlsl:23-35.  log2f is the OCML (1 ulp) function, not v_log_f32 raw:
// the value is floor()ed into a bin.  Un-contracted so the arithmetic matches the shader's.
__device__ __forceinline__ uint32_t luminance_bin(float r, float g, float b, float min_log, float inv_range) {
#pragma clang fp contract(off)
    const float lum = (r * 0.2126f + g * 0.7152f) + b * 0.0722f;
    if (lum < EPSILON_F) return 0u;
    const float l = saturatef((log2f(lum) - min_log) * inv_range);
    return (uint32_t)floorf(l * 254.0f + 1.0f);
}

// Few, large blocks: every block ends with up to 256 global atomics on the same 256 addresses, which the L2 serialises
// per address — with 1024 blocks of 256 threads that flush, not the 66 MB read, set the kernel's time (31 us at 4K).
constexpr int HIST_BLOCKS = 512, HIST_NT = 1024, HIST_NW = HIST_NT / 64;

__global__ __launch_bounds__(HIST_NT) void k_lum_histogram(const pbr_half* __restrict__ hdr, uint32_t w, uint32_t h, uint32_t pitch,
                                                             float min_log, float inv_range, uint32_t* __restrict__ hist, bool vec2) {
    __shared__ uint32_t sh[HIST_NW][PBR_HISTOGRAM_BINS];
    const int wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < HIST_NW * PBR_HISTOGRAM_BINS; i += HIST_NT) (&sh[0][0])[i] = 0u;
    __syncthreads();
    uint32_t* my = sh[wave];
    // Work unit = one 256-lane segment of a row; a block takes four at a time (one per 256-thread group), grid-stride.  The
    // row / segment split is one group-uniform 32-bit division per unit (a per-pixel 64-bit t / w cost more than the binning).
    const uint32_t group = threadIdx.x >> 8, gl = threadIdx.x & 255u, stride = gridDim.x * 4u;
    // vec2 (host-checked): base 16-byte aligned, w and pitch even -> rows are whole 16-byte pixel pairs
    if (vec2) {
        // two units per trip: both 16-byte loads are in flight before either is binned
        const uint32_t wp = w >> 1, upr = (wp + 255u) >> 8, units = upr * h;
        for (uint32_t u = blockIdx.x * 4u + group; u < units; u += 2 * stride) {
            const uint32_t u2 = u + stride;
            const uint32_t ya = u / upr, xa = (u - ya * upr) * 256u + gl;
            const uint32_t yb = u2 / upr, xb = (u2 - yb * upr) * 256u + gl;
            const bool ina = xa < wp, inb = u2 < units && xb < wp;
            uint4 ra = make_uint4(0u, 0u, 0u, 0u), rb = ra;
            if (ina) ra = *reinterpret_cast<const uint4*>(hdr + 4 * ((size_t)ya * pitch + 2 * xa));
            if (inb) rb = *reinterpret_cast<const uint4*>(hdr + 4 * ((size_t)yb * pitch + 2 * xb));
            const H4 p0 = *reinterpret_cast<const H4*>(&ra.x), p1 = *reinterpret_cast<const H4*>(&ra.z);
            const H4 p2 = *reinterpret_cast<const H4*>(&rb.x), p3 = *reinterpret_cast<const H4*>(&rb.z);
            hist_count(my, luminance_bin((float)p0.x, (float)p0.y, (float)p0.z, min_log, inv_range), ina);
            hist_count(my, luminance_bin((float)p1.x, (float)p1.y, (float)p1.z, min_log, inv_range), ina);
            hist_count(my, luminance_bin((float)p2.x, (float)p2.y, (float)p2.z, min_log, inv_range), inb);
            hist_count(my, luminance_bin((float)p3.x, (float)p3.y, (float)p3.z, min_log, inv_range), inb);
        }
    } else {
        const uint32_t upr = (w + 255u) >> 8, units = upr * h;
        for (uint32_t u = blockIdx.x * 4u + group; u < units; u += stride) {
            const uint32_t y = u / upr, x = (u - y * upr) * 256u + gl;
            const bool in = x < w;
            F4 c = f4(0.0f, 0.0f, 0.0f, 0.0f);
            if (in) c = load_h4(hdr + 4 * ((size_t)y * pitch + x));
            hist_count(my, luminance_bin(c.x, c.y, c.z, min_log, inv_range), in);
        }
    }
    __syncthreads();
    if (threadIdx.x < PBR_HISTOGRAM_BINS) {
        uint32_t s = 0;
#pragma unroll
        for (int k = 0; k < HIST_NW; k++) s += sh[k][threadIdx.x];
        if (s) atomicAdd(&hist[threadIdx.x], s);
    }
}

// hdr_average_histogram.hlsl:26-73 — one 256-thread group, the same LDS tree (fixed fp32 order).
// hdr_average_histogram.hlsl:26-73 for a 256-thread block: thread 0 returns the adapted luminance (others 0)
__device__ __forceinline__ float adapted_luminance(float* sh, const uint32_t* __restrict__ hist, uint32_t pixel_count, float min_log, float range,
                                                   float delta_time, float prev) {
#pragma clang fp contract(off)
    const uint32_t index = threadIdx.x;
    const uint32_t num_pixels = hist[index];
    sh[index] = (float)(uint32_t)(num_pixels * index);   // uint32 product (Q15)
    __syncthreads();
    for (uint32_t step = PBR_HISTOGRAM_BINS >> 1; step > 0; step >>= 1) {
        if (index < step) sh[index] += sh[index + step];
        __syncthreads();
    }
    float result = 0.0f;
    if (index == 0) {
        const float sum_value = sh[0];
        const float average_bin = sum_value / (float)(pixel_count - num_pixels);   // Q14
        // BinIndexToLuminance(uint): float -> uint truncation (Q13); NaN / negative -> 0
        uint32_t bin = (average_bin == average_bin && average_bin > 0.0f)
                           ? (average_bin >= 4294967296.0f ? 0xFFFFFFFFu : (uint32_t)average_bin) : 0u;
        const float log_l = ((float)bin - 1.0f) / 254.0f;
        const float lum = exp2f(log_l * range + min_log);
        const float tt = saturatef(1.0f - expf(-delta_time * 1.6f));   // SMOOTH_TIME 1.6
        result = prev + tt * (lum - prev);
    }
    return result;
}

__global__ __launch_bounds__(256) void k_lum_average(uint32_t* __restrict__ hist, uint32_t pixel_count, float min_log, float range,
                                                       float delta_time, float* __restrict__ avg) {
    __shared__ float sh[PBR_HISTOGRAM_BINS];
    const float r = adapted_luminance(sh, hist, pixel_count, min_log, range, delta_time, avg[0]);
    hist[threadIdx.x] = 0u;   // clear for the next frame (every thread read its bin before the first barrier above)
    if (threadIdx.x == 0) avg[0] = r;
}

// hdr_tone_mapping.hlsl:27-36.  The result is quantised to 8 bits (tolerance 1 LSB), so the fast
// reciprocal / log / exp instructions are used throughout: the pass must stay HBM-bound (12 B/pixel),
// and three IEEE divisions + a libm pow per channel would make it VALU-bound.
__device__ __forceinline__ float aces1(float x) {
    const float a = 2.51f, b = 0.03f, c = 2.43f, d = 0.59f, e = 0.14f;
    return saturatef((x * (a * x + b)) * rcp(x * (c * x + d) + e));
}
__device__ __forceinline__ uint32_t tonemap_px(float r, float g, float b, float inv_den) {
    // exposed = luminance / (l_max + 0.001); pow(x, 0.454545) = exp2(0.454545*log2(x)) (x in [0,1])
    const float m[3] = {aces1(r * inv_den), aces1(g * inv_den), aces1(b * inv_den)};
    uint32_t px = 0xFF000000u;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float gc = m[k] > 0.0f ? __builtin_amdgcn_exp2f(0.454545f * __builtin_amdgcn_logf(m[k])) : 0.0f;
        px |= (uint32_t)(saturatef(gc) * 255.0f + 0.5f) << (8 * k);
    }
    return px;
}

// two pixels at once in the halves of packed-fp32 registers (v_pk_fma / v_pk_mul: half the VALU issues of the polynomial
// part; the transcendentals stay one per value)
typedef float tm2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ tm2 aces2(tm2 x) {
    const float a = 2.51f, b = 0.03f, c = 2.43f, d = 0.59f, e = 0.14f;
    const tm2 num = x * (x * a + b), den = x * (x * c + d) + e;
    tm2 q;
    q.x = saturatef(num.x * rcp(den.x));
    q.y = saturatef(num.y * rcp(den.y));
    return q;
}
__device__ __forceinline__ void tonemap_px2(tm2 r, tm2 g, tm2 b, float inv_den, uint32_t& p0, uint32_t& p1) {
    const tm2 m[3] = {aces2(r * inv_den), aces2(g * inv_den), aces2(b * inv_den)};
    p0 = p1 = 0xFF000000u;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        tm2 lg, gc;
        lg.x = __builtin_amdgcn_logf(m[k].x); lg.y = __builtin_amdgcn_logf(m[k].y);
        lg = lg * 0.454545f;
        gc.x = m[k].x > 0.0f ? __builtin_amdgcn_exp2f(lg.x) : 0.0f;
        gc.y = m[k].y > 0.0f ? __builtin_amdgcn_exp2f(lg.y) : 0.0f;
        gc.x = saturatef(gc.x); gc.y = saturatef(gc.y);
        const tm2 v = gc * 255.0f + 0.5f;
        p0 |= (uint32_t)v.x << (8 * k);
        p1 |= (uint32_t)v.y << (8 * k);
    }
}

// persistent grid (<= 2048 blocks of 256), two pixels per lane per trip: 16-byte load, 8-byte store
__device__ __forceinline__ void tonemap_pixels(const pbr_half* __restrict__ hdr, uint32_t w, uint32_t h, uint32_t pitch,
                                               float avg_lum, uint32_t* __restrict__ out, uint32_t out_pitch, bool aligned) {
    const float l_max = 9.6f * avg_lum;
    const float inv_den = 1.0f / (l_max + 0.001f);
    const uint32_t wp = (w + 1) >> 1;   // pixel pairs per row
    // work unit = one 256-lane segment of a row (one wave-uniform 32-bit division per unit, none per pixel)
    const uint32_t upr = (wp + 255u) >> 8, units = upr * h;
    for (uint32_t u = blockIdx.x; u < units; u += gridDim.x) {
        const uint32_t y = u / upr, xp = (u - y * upr) * 256u + threadIdx.x;
        if (xp >= wp) continue;
        const uint32_t x = xp * 2;
        if (aligned && x + 1 < w) {
            const uint4 raw = *reinterpret_cast<const uint4*>(hdr + 4 * ((size_t)y * pitch + x));
            const H4 p0 = *reinterpret_cast<const H4*>(&raw.x);
            const H4 p1 = *reinterpret_cast<const H4*>(&raw.z);
            uint2 o;
            tm2 r2, g2, b2;
            r2.x = (float)p0.x; r2.y = (float)p1.x; g2.x = (float)p0.y; g2.y = (float)p1.y; b2.x = (float)p0.z; b2.y = (float)p1.z;
            tonemap_px2(r2, g2, b2, inv_den, o.x, o.y);
            *reinterpret_cast<uint2*>(out + (size_t)y * out_pitch + x) = o;
        } else {
            for (uint32_t xx = x; xx < min(x + 2, w); xx++) {
                const F4 c = load_h4(hdr + 4 * ((size_t)y * pitch + xx));
                out[(size_t)y * out_pitch + xx] = tonemap_px(c.x, c.y, c.z, inv_den);
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_tonemap(const pbr_half* __restrict__ hdr, uint32_t w, uint32_t h, uint32_t pitch,
                                                   const float* __restrict__ avg, uint32_t* __restrict__ out, uint32_t out_pitch, bool aligned) {
    tonemap_pixels(hdr, w, h, pitch, avg[0], out, out_pitch, aligned);
}

// hdr_average_histogram.hlsl + hdr_tone_mapping.hlsl in ONE launch (round 4).  The average is a 256-bin reduction — microseconds of
// work behind a launch of its own (4.7 us + a dependency gap in the 4K frame): here every block of the tone-map re-derives it from the
// bins with the same fixed-order LDS tree (bit-identical in every block), then tone-maps its pixels.  Nothing the blocks read is
// written by this launch: the adapted luminance goes to avg_OUT (!= avg_in), and the histogram that is zeroed "for the next frame" is
// ANOTHER one (hist_clear: the one the next frame accumulates into; the counts read here are cleared by the next frame's call) — the
// caller alternates two histograms and two luminance cells, so no block can race block 0's writes.
__global__ __launch_bounds__(256) void k_average_tonemap(const uint32_t* __restrict__ hist, uint32_t pixel_count, float min_log, float range,
                                                           float delta_time, const float* __restrict__ avg_in, float* __restrict__ avg_out,
                                                           uint32_t* __restrict__ hist_clear,
                                                           const pbr_half* __restrict__ hdr, uint32_t w, uint32_t h, uint32_t pitch,
                                                           uint32_t* __restrict__ out, uint32_t out_pitch, bool aligned) {
    __shared__ float sh[PBR_HISTOGRAM_BINS];
    __shared__ float s_avg;
    const float r = adapted_luminance(sh, hist, pixel_count, min_log, range, delta_time, avg_in[0]);
    if (threadIdx.x == 0) s_avg = r;
    if (blockIdx.x == 0) {
        if (hist_clear) hist_clear[threadIdx.x] = 0u;
        if (threadIdx.x == 0) avg_out[0] = r;
    }
    __syncthreads();
    tonemap_pixels(hdr, w, h, pitch, s_avg, out, out_pitch, aligned);
}

extern "C" {

pbr_status pbr_lum_histogram(pbr_ctx* ctx, const pbr_half* hdr, uint32_t w, uint32_t h, uint32_t pitch,
                             float min_log, float inv_range, uint32_t* hist256) {
    if (!ctx) return PBR_ERR_INVALID;
    PBR_REQUIRE(ctx, hdr && hist256, "pbr_lum_histogram: null pointer");
    PBR_REQUIRE(ctx, w && h && w <= 65535 && h <= 65535 && pitch >= w, "pbr_lum_histogram: bad size");
    static const int max_blocks = pbr::knob_int("PBR_HIST_BLOCKS", HIST_BLOCKS);   // sweep switch (knobs build only)
    size_t n = (size_t)w * h;
    int blocks = (int)((n + 4095) / 4096);   // >= two trips of a block's four 256-pair segments
    if (blocks > max_blocks) blocks = max_blocks;
    if (blocks < 1) blocks = 1;
    const bool vec2 = (((uintptr_t)hdr & 15u) == 0u) && (((w | pitch) & 1u) == 0u);
    hipLaunchKernelGGL(k_lum_histogram, dim3(blocks), dim3(HIST_NT), 0, ctx->stream, hdr, w, h, pitch, min_log, inv_range, hist256, vec2);
    return launched(ctx, "k_lum_histogram");
}

pbr_status pbr_lum_average(pbr_ctx* ctx, uint32_t* hist256, uint32_t pixel_count, float min_log, float range,
                           float delta_time, float* avg_inout) {
    if (!ctx) return PBR_ERR_INVALID;
    PBR_REQUIRE(ctx, hist256 && avg_inout, "pbr_lum_average: null pointer");
    hipLaunchKernelGGL(k_lum_average, dim3(1), dim3(256), 0, ctx->stream, hist256, pixel_count, min_log, range, delta_time, avg_inout);
    return launched(ctx, "k_lum_average");
}

pbr_status pbr_tonemap(pbr_ctx* ctx, const pbr_half* hdr, uint32_t w, uint32_t h, uint32_t pitch,
                       const float* avg, uint32_t* rgba8, uint32_t out_pitch) {
    if (!ctx) return PBR_ERR_INVALID;
    PBR_REQUIRE(ctx, hdr && avg && rgba8, "pbr_tonemap: null pointer");
    PBR_REQUIRE(ctx, w && h && w <= 65535 && h <= 65535 && pitch >= w && out_pitch >= w, "pbr_tonemap: bad size");
    size_t pairs = (size_t)((w + 1) / 2) * h;
    unsigned blocks = (unsigned)((pairs + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    dim3 grid(blocks);
    const bool aligned = (((uintptr_t)hdr & 15u) == 0u) && (((uintptr_t)rgba8 & 7u) == 0u) && (((pitch | out_pitch) & 1u) == 0u);
    hipLaunchKernelGGL(k_tonemap, grid, dim3(256), 0, ctx->stream, hdr, w, h, pitch, avg, rgba8, out_pitch, aligned);
    return launched(ctx, "k_tonemap");
}

pbr_status pbr_average_tonemap(pbr_ctx* ctx, const uint32_t* hist256, uint32_t pixel_count, float min_log, float range, float delta_time,
                               const float* avg_in, float* avg_out, uint32_t* hist_clear256,
                               const pbr_half* hdr, uint32_t w, uint32_t h, uint32_t pitch, uint32_t* rgba8, uint32_t out_pitch) {
    if (!ctx) return PBR_ERR_INVALID;
    PBR_REQUIRE(ctx, hist256 && avg_in && avg_out && hdr && rgba8, "pbr_average_tonemap: null pointer");
    PBR_REQUIRE(ctx, avg_in != avg_out && hist_clear256 != hist256, "pbr_average_tonemap: avg_out must differ from avg_in and hist_clear256 from hist256 (every block reads them)");
    PBR_REQUIRE(ctx, w && h && w <= 65535 && h <= 65535 && pitch >= w && out_pitch >= w, "pbr_average_tonemap: bad size");
    size_t pairs = (size_t)((w + 1) / 2) * h;
    unsigned blocks = (unsigned)((pairs + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    const bool aligned = (((uintptr_t)hdr & 15u) == 0u) && (((uintptr_t)rgba8 & 7u) == 0u) && (((pitch | out_pitch) & 1u) == 0u);
    hipLaunchKernelGGL(k_average_tonemap, dim3(blocks), dim3(256), 0, ctx->stream, hist256, pixel_count, min_log, range, delta_time, avg_in, avg_out,
                       hist_clear256, hdr, w, h, pitch, rgba8, out_pitch, aligned);
    return launched(ctx, "k_average_tonemap");
}

}  // extern "C"
