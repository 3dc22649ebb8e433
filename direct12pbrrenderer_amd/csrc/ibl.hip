// ibl.hip — one-shot IBL precompute on gfx950: split-sum BRDF LUT, cube box mips,
// GGX-prefiltered environment cube, SH9 irradiance projection.
//
// Design notes (MI355X): all three integrators are ALU/L2-gather bound, not HBM bound.
// The 1 024 Hammersley/GGX sample directions of a launch depend only on (i, roughness), so
// every block builds them ONCE into LDS with the precise libm-class functions and the
// per-texel loop reads them back as wave-uniform (broadcast) ds_reads — the inner loop then
// has no sin/cos/sqrt/log2 at all.  Samples are accumulated sequentially per texel in the
// reference's order (precompute_brdf.hlsl:33-56, env_map_gen.hlsl:69-101).
#include "pbr_internal.hpp"
#include "pbr_device.hpp"

using namespace pbr;

// ============================================================================ BRDF LUT (a3)
// grid (res, ceil(res/256)), block 256: one block = one roughness column x, 256 NdotV rows.
// a * b and a + b saturated to [0, 1] through the clamp modifier: a plain-class instruction where `max(x, 0)` is a second one of the
// slow class (v_max_f32 issues at ~4.6 cycles per wave against 2.7, profiles/r03_valu_rate3b.txt).  The operands here never exceed 1
// by more than rounding.  (s_nop: the wait state a VALU consumer needs behind v_rsq / v_rcp, which the compiler cannot see into asm.)
__device__ __forceinline__ float mul_sat(float a, float b) {
    float r;
    asm("s_nop 0\n\tv_mul_f32_e64 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float add_sat(float a, float b) {
    float r;
    asm("v_add_f32_e64 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// a * b + c saturated to [0, 1]
__device__ __forceinline__ float fma_sat(float a, float b, float c) {
    float r;
    asm("v_fma_f32 %0, %1, %2, %3 clamp" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// The sample loop of one LUT texel.  Same estimator as IntegrateBRDF (brdf.hlsli:137-167 — restated in oracle/pbr_oracle.cpp
// orc_brdf_lut), arranged for what gfx950 issues fast (fp32 mul / add / fma ~2.7 cycles per wave, v_max / v_min ~4.6,
// transcendentals 8.5); each step stays within a few 1e-7 relative of the shader's expression, three orders below the fp16
// rounding of the result (every texel within 1 fp16 ULP of the oracle: tests/test_gpu_parity.py):
//  * L = 2 (V.H) H - V is a unit vector to ~3e-7 when V and H are: normalize(L) — two more components, a dot product and a v_rsq
//    — is dropped, N.L = L.z comes out of one FMA whose clamp modifier is the max(., 0);
//  * 1 / max(NdotH NdotV, 1e-4) = (1 / NdotV) min(1 / NdotH, 1e4 NdotV): 1 / NdotH rides in the sample table, 1 / NdotV and the
//    per-texel factor gv leave the loop: no reciprocal here, and ONE transcendental per sample (the v_rcp of gl's denominator);
//  * no branch on NdotL > 0: the sample's weight is an exact +0 there and every factor is finite;
//  * multiply-adds are fused (this TU is built -ffp-contract=off for the table's sin(theta), not for this loop).
// tab[i] = (H.x, 2 H.z, H.z, 1 / max(H.z, 0)).  KSAFE: k = roughness^2 / 2 >= 1e-6, the shader's max(NdotL (1 - k) + k, 1e-6)
// never binds (NdotL >= 0).
template <bool KSAFE>
__device__ __forceinline__ void brdf_lut_samples(const float4* tab, float Vx, float Vz, float NdotV, float k, float one_k, float& A_out, float& B_out) {
    float A = 0.0f, B = 0.0f;
    const float c1 = 1.0e4f * NdotV, nVz = -Vz;
#pragma unroll 4
    for (uint32_t i = 0; i < PBR_SAMPLE_COUNT; i++) {
        const float4 H = tab[i];
        const float vx = Vx * H.x, vz = Vz * H.z;   // V.y == 0
        const float VdH = vx + vz;
        const float VdotH = add_sat(vx, vz);        // max(V.H, 0)
        const float NdotL = fma_sat(VdH, H.y, nVz); // max(2 (V.H) H.z - V.z, 0)
        const float omv = 1.0f - VdotH;
        const float o2 = omv * omv;
        const float Fc = o2 * o2 * omv;
        float den = __builtin_fmaf(NdotL, one_k, k);
        if (!KSAFE) den = fmaxf(den, EPSILON_F);
        const float g = (NdotL * __builtin_amdgcn_rcpf(den)) * (VdotH * fminf(H.w, c1));   // G_Vis / (gv / NdotV)
        A = __builtin_fmaf(1.0f - Fc, g, A);
        B = __builtin_fmaf(Fc, g, B);
    }
    A_out = A; B_out = B;
}

__global__ __launch_bounds__(256) void k_brdf_lut(uint32_t res, pbr_half* __restrict__ out) {
    __shared__ float4 tab[PBR_SAMPLE_COUNT];   // per sample, from the normalized H in the N=(0,0,1) frame: (H.x, 2 H.z, H.z, 1 / NdotH)
    const uint32_t x = blockIdx.x;
    const float roughness = (float)x / (float)(res - 1);
    for (uint32_t i = threadIdx.x; i < PBR_SAMPLE_COUNT; i += 256) {
        float xi_x = (float)i / (float)PBR_SAMPLE_COUNT;
        float xi_y = radical_inverse_vdc(i);
        V3 H = ggx_important_sample(roughness, v3(0.0f, 0.0f, 1.0f), xi_x, xi_y);
        tab[i] = make_float4(H.x, 2.0f * H.z, H.z, 1.0f / fmaxf(H.z, 0.0f));   // 1 / 0 = +inf: min(inf, 1e4 NdotV) in the loop is the shader's floor
    }
    __syncthreads();
    const uint32_t y = blockIdx.y * 256 + threadIdx.x;
    if (y >= res) return;
    const float NdotV = (float)(y + 1) / (float)res;
    const float Vx = sqrtf(1.0f - NdotV * NdotV), Vz = NdotV;
    const float k = roughness * roughness / 2.0f;   // Q6: k = r^2/2 in the LUT
    const float one_k = 1.0f - k;
    const float gv = NdotV / fmaxf(NdotV * one_k + k, EPSILON_F);
    float A, B;
    if (k >= EPSILON_F) brdf_lut_samples<true>(tab, Vx, Vz, NdotV, k, one_k, A, B);   // block-uniform (every column but roughness ~ 0)
    else brdf_lut_samples<false>(tab, Vx, Vz, NdotV, k, one_k, A, B);
    const float scale = gv / NdotV;   // the factors of G_Vis that do not depend on the sample
    A *= scale; B *= scale;
    A = A / (float)PBR_SAMPLE_COUNT;
    B = B / (float)PBR_SAMPLE_COUNT;
    H2 o;
    o.x = to_half_rn(A);
    o.y = to_half_rn(B);
    *reinterpret_cast<H2*>(out + 2 * ((size_t)y * res + x)) = o;
}

// ============================================================================ cube box mips
__global__ __launch_bounds__(256) void k_cube_downsample(const float* __restrict__ src, float* __restrict__ dst, uint32_t s) {
    // dst mip edge s, src edge 2s; one thread per dst texel
    const size_t n = (size_t)6 * s * s;
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    const uint32_t x = (uint32_t)(t % s), y = (uint32_t)((t / s) % s), f = (uint32_t)(t / ((size_t)s * s));
    const uint32_t sp = 2 * s;
    const float4* r0 = reinterpret_cast<const float4*>(src) + ((size_t)f * sp + 2 * y) * sp + 2 * x;
    const float4* r1 = r0 + sp;
    float4 a = r0[0], b = r0[1], c = r1[0], d = r1[1];
    float4 o;
    o.x = ((a.x + b.x) + (c.x + d.x)) * 0.25f;
    o.y = ((a.y + b.y) + (c.y + d.y)) * 0.25f;
    o.z = ((a.z + b.z) + (c.z + d.z)) * 0.25f;
    o.w = ((a.w + b.w) + (c.w + d.w)) * 0.25f;
    reinterpret_cast<float4*>(dst)[t] = o;
}

// ============================================================================ env prefilter (a4)
// One launch per output mip.  grid ceil(6*s*s/256), block 256, one thread per output texel.
// LDS table per block: tangent-space h_i (x,y,z) and the source LOD of sample i (with N = V the
// pdf — hence the LOD — depends only on i and the roughness: pdf = D(h.z)*h.z / (4 h.z + 1e-4)).
__device__ __forceinline__ void prefilter_block(float4* tab, const float* __restrict__ sky, uint32_t sky_size, uint32_t sky_mips,
                                                uint32_t size, uint32_t s, float roughness, pbr_half* __restrict__ out,
                                                uint32_t block_in_mip) {
    for (uint32_t i = threadIdx.x; i < PBR_SAMPLE_COUNT; i += 256) {
        float xi_x = (float)i / (float)PBR_SAMPLE_COUNT;
        float xi_y = radical_inverse_vdc(i);
        float a = roughness * roughness;
        float phi = TWO_PI_F * xi_x;
        float cos_theta = sqrtf((1.0f - xi_y) / (1.0f + (a * a - 1.0f) * xi_y));
        float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
        float NdotH = fmaxf(cos_theta, 0.0f), HdotV = NdotH;
        float D = distribution_ggx(NdotH, roughness);
        float pdf = D * NdotH / (4.0f * HdotV + 0.0001f);
        float texel_sa = 4.0f * PI_F / ((float)(6u * size * size));   // base size for every mip (Q8)
        float sample_sa = 1.0f / ((float)PBR_SAMPLE_COUNT * pdf + 0.0001f);
        float lod = roughness == 0.0f ? 0.0f : 0.5f * log2f(sample_sa / texel_sa);
        tab[i] = make_float4(sin_theta * cosf(phi), sin_theta * sinf(phi), cos_theta, lod);
    }
    __syncthreads();
    const size_t n = (size_t)6 * s * s;
    const size_t t = (size_t)block_in_mip * 256 + threadIdx.x;
    if (t >= n) return;
    const uint32_t x = (uint32_t)(t % s), y = (uint32_t)((t / s) % s), face = (uint32_t)(t / ((size_t)s * s));
    const float u = (float)x / (float)s, v = (float)y / (float)s;   // texel corner (Q8)
    const V3 N = normalize3_exact(cube_dir_raw(face, 2.0f * u - 1.0f, 2.0f * v - 1.0f));
    const V3 up = fabsf(N.z) < 0.999f ? v3(0.0f, 0.0f, 1.0f) : v3(1.0f, 0.0f, 0.0f);
    const V3 T = normalize3_exact(cross3(N, up));
    const V3 Bt = cross3(N, T);
    float cr = 0.0f, cg = 0.0f, cb = 0.0f, wsum = 0.0f;
    if (roughness == 0.0f) {
        // Roughness 0 (mip 0): sin(theta) is exactly 0 for every sample, so H, L, N.L, the LOD (0) and the
        // fetched colour are the same 1 024 times.  Fetch once and replay only the accumulation, which keeps
        // the reference's running-sum rounding (env_map_gen.hlsl:69-101) without 1 023 redundant fetches.
        const float4 h = tab[0];
        const V3 H = normalize3(T * h.x + Bt * h.y + N * h.z);
        const float VdH = dot3(N, H);
        const V3 L = normalize3(H * (2.0f * VdH) - N);
        const float NdotL = fmaxf(dot3(N, L), 0.0f);
        if (NdotL > 0.0f) {
            const F4 c = cube_trilinear<CubeTexelF32>(sky, sky_size, sky_mips, L, 0.0f);
            const float pr = c.x * NdotL, pg = c.y * NdotL, pb = c.z * NdotL;
            for (uint32_t i = 0; i < PBR_SAMPLE_COUNT; i++) {
                cr += pr; cg += pg; cb += pb;
                wsum += NdotL;
            }
        }
    } else {
        for (uint32_t i = 0; i < PBR_SAMPLE_COUNT; i++) {
            const float4 h = tab[i];
            V3 H = normalize3(T * h.x + Bt * h.y + N * h.z);
            float VdH = dot3(N, H);   // V = N
            V3 L = normalize3(H * (2.0f * VdH) - N);
            float NdotL = fmaxf(dot3(N, L), 0.0f);
            if (NdotL > 0.0f) {
                F4 c = cube_trilinear<CubeTexelF32>(sky, sky_size, sky_mips, L, h.w);
                cr += c.x * NdotL;
                cg += c.y * NdotL;
                cb += c.z * NdotL;
                wsum += NdotL;
            }
        }
    }
    const float inv = 1.0f / wsum;   // wsum == 0 -> NaN like the reference's 0/0
    store_h4(out + 4 * t, f4(cr * inv, cg * inv, cb * inv, 1.0f));
}

// one output mip (one dispatch of PreFilterEnvMapPass::Execute)
__global__ __launch_bounds__(256) void k_prefilter_env(const float* __restrict__ sky, uint32_t sky_size, uint32_t sky_mips,
                                                         uint32_t size, uint32_t s, float roughness,
                                                         pbr_half* __restrict__ out) {
    __shared__ float4 tab[PBR_SAMPLE_COUNT];
    prefilter_block(tab, sky, sky_size, sky_mips, size, s, roughness, out, blockIdx.x);
}

#ifdef PBR_DEBUG_KNOBS   // measured alternative of round 1 (PBR_PREFILTER_SEQ=1): only the knobs build carries it
// All mips in ONE launch.  A thread walks its 1 024 samples one after the other (the reference's running sum), so a
// dispatch lasts as long as that serial chain whatever the mip's size — five dispatches in a row cost five chains
// with a mostly idle chip (mip 4 is 96 waves).  Run together the mips overlap: blocks are ordered mip 1, 2, ..,
// mips-1, then mip 0 (roughness 0: one fetch per texel) so the long chains start first.  Same arithmetic per texel.
__global__ __launch_bounds__(256) void k_prefilter_env_all(const float* __restrict__ sky, uint32_t sky_size, uint32_t sky_mips,
                                                             uint32_t size, uint32_t mips, pbr_half* __restrict__ out) {
    __shared__ float4 tab[PBR_SAMPLE_COUNT];
    uint32_t b = blockIdx.x, mip = 0;
    for (uint32_t k = 0; k < mips; k++) {
        const uint32_t m = (k + 1 < mips) ? k + 1 : 0;   // order 1, 2, .., mips-1, 0
        const uint32_t sm = size >> m;
        const uint32_t nb = (uint32_t)(((size_t)6 * sm * sm + 255) / 256);
        if (b < nb) { mip = m; break; }
        b -= nb;
    }
    const float roughness = mips > 1 ? (float)mip / (float)(mips - 1) : 0.0f;   // DeferredPipeline.cpp:99
    prefilter_block(tab, sky, sky_size, sky_mips, size, size >> mip, roughness, out + 4 * cube_mip_offset(size, mip), b);
}
#endif

// ---------------------------------------------------------------------------- wave-parallel prefilter (pbr_prefilter_env)
// The same integral, mapped the way the hardware likes it:
//  * ONE WAVE PER OUTPUT TEXEL: the 64 lanes take the texel's samples 64 at a time (<= 16 trips instead of a 1 024-step
//    dependent chain), keep per-lane partial sums and combine them with a fixed xor-shuffle tree — deterministic, but
//    not the shader's strictly sequential fp32 sum (SURVEY 8c allows <= 1 fp16 ULP or 1e-3 for this output; the
//    sequential kernel above remains what pbr_prefilter_env_mip runs, one reference dispatch at a time).  N, T, B of the
//    texel are wave-uniform.  mip 4 is 6 144 waves instead of 96.
//  * the sample set of a mip is a TABLE built on the host with the oracle's own libm calls: with V = N the half vector
//    in tangent space depends only on (i, roughness), and so do the reflected direction L_t = (2 hz hx, 2 hz hy,
//    2 hz^2 - 1), the weight N.L = L_t.z and the source LOD.  Samples with N.L <= 0 are dropped when the table is built
//    (a quarter of them at roughness 0.75, half at 1) and the weight sum is one number per mip.  The loop keeps no
//    normalize, no sqrt, no log2: L = T lx + B ly + N lz.
//  * the source cube is sampled from a PADDED copy (every face of every mip with the 1-texel border the seamless rule
//    selects, like the shade's env chain): the trilinear fetch is branch-free, 8 x 16-byte loads per sample.
//  * roughness 0 (mip 0) degenerates to one bilinear fetch per texel: its own thread-per-texel kernel.
constexpr int PF_TEXELS_PER_WAVE = 8;
constexpr int PF_TEXELS_PER_BLOCK = 4 * PF_TEXELS_PER_WAVE;
struct PfLaunch {
    uint32_t first_block[17];   // first block of output mip m (m = 1 .. mips-1), [mips] = total
    uint32_t count[16];         // valid samples of mip m
    float wsum[16];             // sum of their weights, accumulated in sample order in fp32 like the shader's total_weight
    uint32_t src_off[16];       // texel offset of padded SOURCE mip l
    uint32_t mips, size, sky_size, sky_mips;
    uint32_t xcd_groups;        // k_prefilter_foot: 8 = blocks sharing blockIdx % 8 (one XCD) take one contiguous eighth of a mip's texels; 1 = plain order
};

// trilinear fetch of the padded fp32 chain along `d` at the (already clamped and x.8-snapped) LOD; rgb only
__device__ __forceinline__ V3 padded_trilinear(const float4* __restrict__ sky, const PfLaunch& pl, V3 d, float lod_s) {
    uint32_t face;
    float cu, cv;
    cube_face_uv(d, face, cu, cv);
    const float fl = floorf(lod_s), f = lod_s - fl;
    const uint32_t l0 = (uint32_t)fl, l1 = min(l0 + 1u, pl.sky_mips - 1u);
    auto level = [&](uint32_t l) {
        const int s = (int)(pl.sky_size >> l), sp = s + 2;
        const float fxp = snap8(cu * (float)s) - 0.5f, fyp = snap8(cv * (float)s) - 0.5f;
        const float flx = floorf(fxp), fly = floorf(fyp);
        const float fx = fxp - flx, fy = fyp - fly;
        const float4* m = sky + pl.src_off[l] + ((size_t)face * sp + (size_t)((int)fly + 1)) * sp + (size_t)((int)flx + 1);
        const float4 c00 = m[0], c10 = m[1], c01 = m[sp], c11 = m[sp + 1];
        // the sampler's lerps on rgb (alpha is not used): far tap fused onto the weighted near tap, the oracle's order.
        // Its "a tap of weight exactly 0 does not contribute" select is left out: it only matters for inf / NaN texels.
        const float wx0 = 1.0f - fx, wy0 = 1.0f - fy;
        const float tr = __builtin_fmaf(c10.x, fx, c00.x * wx0), br = __builtin_fmaf(c11.x, fx, c01.x * wx0);
        const float tg = __builtin_fmaf(c10.y, fx, c00.y * wx0), bg = __builtin_fmaf(c11.y, fx, c01.y * wx0);
        const float tb = __builtin_fmaf(c10.z, fx, c00.z * wx0), bb = __builtin_fmaf(c11.z, fx, c01.z * wx0);
        return v3(__builtin_fmaf(br, fy, tr * wy0), __builtin_fmaf(bg, fy, tg * wy0), __builtin_fmaf(bb, fy, tb * wy0));
    };
    const V3 a = level(l0);
    if (f == 0.0f || l1 == l0) return a;   // wave-divergent only where lanes sit on different LODs: both sides are cheap
    const V3 b = level(l1);
    const float w0 = 1.0f - f;
    return v3(__builtin_fmaf(b.x, f, a.x * w0), __builtin_fmaf(b.y, f, a.y * w0), __builtin_fmaf(b.z, f, a.z * w0));
}

#ifdef PBR_DEBUG_KNOBS   // the wave-per-texel mapping: only the knobs build can select it (PBR_PREFILTER_WAVE=1); measured 2.4x slower than lane-per-texel
__global__ __launch_bounds__(256) void k_prefilter_fast(const float4* __restrict__ sky_padded, const float4* __restrict__ tables,
                                                          PfLaunch pl, pbr_half* __restrict__ out) {
    __shared__ float4 tab[PBR_SAMPLE_COUNT];
    uint32_t mip = 1;
    while (mip + 1 < pl.mips && blockIdx.x >= pl.first_block[mip + 1]) mip++;
    const uint32_t count = pl.count[mip];
    for (uint32_t i = threadIdx.x; i < count; i += 256) tab[i] = tables[(size_t)mip * PBR_SAMPLE_COUNT + i];
    __syncthreads();
    const uint32_t s = pl.size >> mip;
    const uint32_t n = 6u * s * s;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane((blockIdx.x - pl.first_block[mip]) * 4u + (threadIdx.x >> 6));
    pbr_half* out_mip = out + 4 * cube_mip_offset(pl.size, mip);
    for (uint32_t k = 0; k < (uint32_t)PF_TEXELS_PER_WAVE; k++) {
        const uint32_t t = wave * PF_TEXELS_PER_WAVE + k;
        if (t >= n) break;
        const uint32_t x = t % s, y = (t / s) % s, face = t / (s * s);
        const float u = (float)x / (float)s, v = (float)y / (float)s;   // texel corner (Q8)
        const V3 N = normalize3_exact(cube_dir_raw(face, 2.0f * u - 1.0f, 2.0f * v - 1.0f));
        const V3 up = fabsf(N.z) < 0.999f ? v3(0.0f, 0.0f, 1.0f) : v3(1.0f, 0.0f, 0.0f);
        const V3 T = normalize3_exact(cross3(N, up));
        const V3 Bt = cross3(N, T);
        float cr = 0.0f, cg = 0.0f, cb = 0.0f;
        for (uint32_t j = lane; j < count; j += 64u) {
            const float4 e = tab[j];
            const V3 L = T * e.x + Bt * e.y + N * e.z;
            const V3 c = padded_trilinear(sky_padded, pl, L, e.w);
            cr += c.x * e.z; cg += c.y * e.z; cb += c.z * e.z;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {   // fixed combine order
            cr += __shfl_xor(cr, off, 64); cg += __shfl_xor(cg, off, 64); cb += __shfl_xor(cb, off, 64);
        }
        if (lane == 0) {
            const float w = pl.wsum[mip];   // 0 samples -> 0/0 = NaN like the reference
            store_h4(out_mip + 4 * (size_t)t, f4(cr / w, cg / w, cb / w, 1.0f));
        }
    }
}
#endif

#ifdef PBR_DEBUG_KNOBS
// The same table-driven, branch-free inner loop with the OTHER mapping: one lane per output texel, the 64 lanes of a
// wave are 64 neighbouring texels and all of them take sample j at the same time.  Neighbouring texels reflect the same
// tangent-space direction into neighbouring source positions, so a wave's 8 x 64 texel fetches fall into a few dozen
// cache lines (measured: 2.4x faster than the wave-per-texel mapping, whose 64 lanes scatter over the whole lobe and
// pull ~256 lines per trip through a 32 KB L1).  The sum runs in sample order like the shader's.  This is what
// pbr_prefilter_env launches; k_prefilter_fast stays selectable (PBR_PREFILTER_WAVE=1) as the measured alternative.
// Round-2 kernel, superseded by k_prefilter_foot<false>: only the knobs build carries it (PBR_PREFILTER_F32=1), as the measured A/B partner.
__global__ __launch_bounds__(256) void k_prefilter_tex(const float4* __restrict__ sky_padded, const float4* __restrict__ tables,
                                                         PfLaunch pl, pbr_half* __restrict__ out) {
    __shared__ float4 tab[PBR_SAMPLE_COUNT];
    uint32_t mip = 1;
    while (mip + 1 < pl.mips && blockIdx.x >= pl.first_block[mip + 1]) mip++;
    const uint32_t count = pl.count[mip];
    for (uint32_t i = threadIdx.x; i < count; i += 256) tab[i] = tables[(size_t)mip * PBR_SAMPLE_COUNT + i];
    __syncthreads();
    const uint32_t s = pl.size >> mip;
    const uint32_t n = 6u * s * s;
    const uint32_t t = (blockIdx.x - pl.first_block[mip]) * 256u + threadIdx.x;
    if (t >= n) return;
    const uint32_t x = t % s, y = (t / s) % s, face = t / (s * s);
    const float u = (float)x / (float)s, v = (float)y / (float)s;   // texel corner (Q8)
    const V3 N = normalize3_exact(cube_dir_raw(face, 2.0f * u - 1.0f, 2.0f * v - 1.0f));
    const V3 up = fabsf(N.z) < 0.999f ? v3(0.0f, 0.0f, 1.0f) : v3(1.0f, 0.0f, 0.0f);
    const V3 T = normalize3_exact(cross3(N, up));
    const V3 Bt = cross3(N, T);
    float cr = 0.0f, cg = 0.0f, cb = 0.0f;
#pragma unroll 2
    for (uint32_t j = 0; j < count; j++) {
        const float4 e = tab[j];   // wave-uniform: one broadcast LDS read
        const V3 L = T * e.x + Bt * e.y + N * e.z;
        const V3 c = padded_trilinear(sky_padded, pl, L, e.w);
        cr += c.x * e.z; cg += c.y * e.z; cb += c.z * e.z;
    }
    const float w = pl.wsum[mip];   // 0 samples -> 0/0 = NaN like the reference
    store_h4(out + 4 * (cube_mip_offset(pl.size, mip) + (size_t)t), f4(cr / w, cg / w, cb / w, 1.0f));
}
#endif

// ---- the same table-driven, texel-per-lane loop on a HALF-precision padded copy of the source chain (what pbr_prefilter_env
// launches for mips >= 1).  Counters of k_prefilter_tex (profiles/r03_*): VALU busy ~100 % (168 instructions per sample) AND
// the texture addresser 74 % busy — eight 16-byte gathers per lane and sample move 128 B through a 64 B / clk path.  Here:
//  * the source chain is copied once per call as half4 with its seam borders (k_cube_pad_chain): a trilinear sample is FOUR
//    16-byte loads = 64 B.  The copy is used ONLY WHEN IT IS EXACT: k_cube_pad_chain raises a flag when any texel does not
//    survive the conversion bit for bit, and then this kernel returns at once and k_prefilter_tex (fp32 chain) does the work —
//    decided on the device, no host round trip.  The reference's sky textures are BC6H_UF16 on disk (BasicStorage.h:10-11,
//    TextureCompression.cpp:95-110): every texel of every mip it can feed this pass IS a half value, so its inputs take this
//    path; an arbitrary fp32 cube keeps the fp32 path and its result.
//    mip 0 — one fetch per texel — stays on the fp32 chain either way (k_prefilter_mip0);
//  * cube face / coordinates from v_cubeid / v_cubesc / v_cubetc / v_cubema + one v_rcp (continuous consumers only);
//  * the sample's eight texels enter the sum as eight weighted v_fma_mix_f32 per channel, the sample weight N.L folded into
//    the weights: no separate lerps, no fp16 -> fp32 converts.
// The LOD of a sample comes from the table: wave-uniform, so level sizes and offsets stay in scalar registers.
typedef uint32_t pf_u4 __attribute__((ext_vector_type(4)));   // two x-adjacent half4 texels as loaded: (a.xy, a.zw, b.xy, b.zw)
struct PfFoot { pf_u4 r0, r1; float w00, w10, w01, w11; };
// rgb += the footprint's four texels x their weights: twelve v_fma_mix_f32 (fp16 operand converted inside the instruction).
// Written as asm because the compiler pairs the channels into v_pk_fma_f32 behind sixteen separate converts whichever way the
// C is written: measured issue costs at >= 5 waves per SIMD (profiles/r03_valu_rate3.txt): v_cvt_f32_f16 4.5 cycles, v_pk_fma_f32
// 4.7 -> 6.9 cycles per product that way, against 4.6 for one v_fma_mix_f32 (which, like the converts, is NOT a 2.6-cycle op).  The s_nop is the wait state a consumer needs behind a packed-fp32
// producer (the weights may come out of v_pk_mul_f32), which the compiler cannot insert for asm (see shade.hip mul2_sat).
__device__ __forceinline__ void pf_accumulate(float& r, float& g, float& b, const PfFoot& f) {
    asm("s_nop 0\n\t"
        "v_fma_mix_f32 %0, %3, %11, %0 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %1, %3, %11, %1 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %2, %4, %11, %2 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %0, %5, %12, %0 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %1, %5, %12, %1 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %2, %6, %12, %2 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %0, %7, %13, %0 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %1, %7, %13, %1 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %2, %8, %13, %2 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %0, %9, %14, %0 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %1, %9, %14, %1 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %2, %10, %14, %2 op_sel_hi:[1,0,0]"
        : "+v"(r), "+v"(g), "+v"(b)
        : "v"(f.r0.x), "v"(f.r0.y), "v"(f.r0.z), "v"(f.r0.w), "v"(f.r1.x), "v"(f.r1.y), "v"(f.r1.z), "v"(f.r1.w),
          "v"(f.w00), "v"(f.w10), "v"(f.w01), "v"(f.w11));
}
// One level of the padded chain as the kernel wants it, 16 bytes = one scalar load (the sample's level is wave-uniform)
struct PfLevel { uint32_t off, sp; float fs256, spf; };   // texel offset of the level, padded edge (level edge + 2), level edge x 256, padded edge as float
struct PfLevels { PfLevel lev[16]; };
struct alignas(8) PfRow { uint32_t x, y, z, w; };   // two x-adjacent half4 texels of a padded row: 8-byte aligned, one 16-byte load
// The sampler's fixed-point footprint of (cu, cv) on one level: texel index of its upper-left texel in the level + the two x.8
// fractions.  Instruction choice follows the measured issue classes (profiles/r03_valu_rate3b.txt: fp32 mul / add / fma and
// integer add / and issue in ~2.7 cycles per wave; converts, floor, shifts, integer multiply-adds, v_readfirstlane in ~4.5):
// everything stays in fp32, where x.8 coordinates below 2^15 and texel indices below 2^24 are exact, up to ONE convert:
//   t  = floor(c * size * 256 + 0.5)            snap8(c * size) in 1/256 texels: bit for bit the shader-side snap8 (* 256 is exact)
//   q  = t / 256 + 0.5                          = snap - 0.5 (texel-centre convention) + 1 (the border column of the padded layout)
//   i  = floor(q), f = q - i                    footprint origin and fraction, both exact
//   o  = (face * sp + iy) * sp + ix             two fp32 FMAs while 6 sp^2 < 2^24 (levels up to 1670 texels), integer multiply-adds above
struct PfCoord { uint32_t o; float fx, fy; };
__device__ __forceinline__ PfCoord pf_coord(const PfLevel& lv, float facef, float cu, float cv) {
    const float tx = floorf(cu * lv.fs256 + 0.5f), ty = floorf(cv * lv.fs256 + 0.5f);
    const float xq = __builtin_fmaf(tx, 1.0f / 256.0f, 0.5f), yq = __builtin_fmaf(ty, 1.0f / 256.0f, 0.5f);
    const float ixf = floorf(xq), iyf = floorf(yq);
    PfCoord c;
    c.fx = xq - ixf; c.fy = yq - iyf;
    c.o = (uint32_t)__builtin_fmaf(__builtin_fmaf(facef, lv.spf, iyf), lv.spf, ixf);
    if (lv.sp > 1672u) c.o = __umul24(__umul24((uint32_t)facef, lv.sp) + (uint32_t)iyf, lv.sp) + (uint32_t)ixf;   // wave-uniform (scalar) branch
    return c;
}
struct PfWeights { float w00, w10, w01, w11; };
__device__ __forceinline__ PfWeights pf_weights(const PfCoord& c, float wl) {   // bilinear weights x the level's weight (N.L folded in)
    PfWeights w;
    const float wy1 = c.fy * wl, wy0 = wl - wy1;
    w.w10 = c.fx * wy0; w.w00 = wy0 - w.w10;
    w.w11 = c.fx * wy1; w.w01 = wy1 - w.w11;
    return w;
}
__device__ __forceinline__ void pf_level_half(float& r, float& g, float& b, const pbr_half* __restrict__ foot, const PfLevel& lv, float facef, float cu, float cv, float wl) {
    const PfCoord c = pf_coord(lv, facef, cu, cv);
    // scalar bases of the two rows, one 32-bit vector offset (host-checked: a level is < 4 GiB)
    const char* row0 = reinterpret_cast<const char*>(foot) + (size_t)lv.off * 8u;
    const char* row1 = row0 + (size_t)lv.sp * 8u;
    const uint32_t vo = c.o * 8u;
    PfFoot f;
    const PfRow a = *reinterpret_cast<const PfRow*>(row0 + vo), bb = *reinterpret_cast<const PfRow*>(row1 + vo);
    f.r0 = pf_u4{a.x, a.y, a.z, a.w};
    f.r1 = pf_u4{bb.x, bb.y, bb.z, bb.w};
    const PfWeights w = pf_weights(c, wl);
    f.w00 = w.w00; f.w10 = w.w10; f.w01 = w.w01; f.w11 = w.w11;
    pf_accumulate(r, g, b, f);
}
// the fp32 twin (the padded fp32 chain has the same texel layout at 16 bytes per texel): what runs when the half copy is not exact
__device__ __forceinline__ void pf_level_f32(float& r, float& g, float& b, const float4* __restrict__ chain, const PfLevel& lv, float facef, float cu, float cv, float wl) {
    const PfCoord c = pf_coord(lv, facef, cu, cv);
    const char* row0 = reinterpret_cast<const char*>(chain) + (size_t)lv.off * 16u;
    const char* row1 = row0 + (size_t)lv.sp * 16u;
    const uint32_t vo = c.o * 16u;   // host-checked: the fp32 chain is < 4 GiB
    // rgb only: a 12-byte load per texel (alpha is never sampled)
    struct Rgb { float x, y, z; };
    const Rgb t00 = *reinterpret_cast<const Rgb*>(row0 + vo), t10 = *reinterpret_cast<const Rgb*>(row0 + vo + 16u);
    const Rgb t01 = *reinterpret_cast<const Rgb*>(row1 + vo), t11 = *reinterpret_cast<const Rgb*>(row1 + vo + 16u);
    const PfWeights w = pf_weights(c, wl);
    r = __builtin_fmaf(t00.x, w.w00, r); g = __builtin_fmaf(t00.y, w.w00, g); b = __builtin_fmaf(t00.z, w.w00, b);
    r = __builtin_fmaf(t10.x, w.w10, r); g = __builtin_fmaf(t10.y, w.w10, g); b = __builtin_fmaf(t10.z, w.w10, b);
    r = __builtin_fmaf(t01.x, w.w01, r); g = __builtin_fmaf(t01.y, w.w01, g); b = __builtin_fmaf(t01.z, w.w01, b);
    r = __builtin_fmaf(t11.x, w.w11, r); g = __builtin_fmaf(t11.y, w.w11, g); b = __builtin_fmaf(t11.z, w.w11, b);
}

// HALF: sample the half copy (runs when it is exact: *lossy == 0); !HALF: the same loop on the padded fp32 chain (runs when
// *lossy != 0).  pbr_prefilter_env launches both; the flag k_cube_pad_chain wrote picks the one that works on the device.
constexpr uint32_t PF_TWO_LEVELS = 0x100u, PF_EXACT_FACE = 0x200u;   // flags beside the level number in a table entry's .w
// Blocks of 512 lanes: the two sample tables take 24 KB of LDS, and the 8 192 waves of a 512^2 cube's mips 1-4 are exactly 8 per
// SIMD — with 256-lane blocks LDS allowed 6 of them at a time and the last two ran alone, their load latency exposed.
constexpr uint32_t PF_FOOT_BLOCK = 512u;
template <bool HALF>
__global__ __launch_bounds__(PF_FOOT_BLOCK) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_prefilter_foot(const void* __restrict__ chain, PfLevels fo, const float4* __restrict__ tables,
                                                          PfLaunch pl, pbr_half* __restrict__ out, const uint32_t* __restrict__ lossy) {
    if ((*lossy != 0u) == HALF) return;
    const pbr_half* foot = reinterpret_cast<const pbr_half*>(chain);
    const float4* chain32 = reinterpret_cast<const float4*>(chain);
    // per sample of this block's mip: (L_t.x, L_t.y, L_t.z = N.L, level | flags) and its two level weights with N.L folded in,
    // ((1 - f) N.L, f N.L), f = the LOD's x.8 fraction.  One v_readfirstlane per sample (the .w word) makes level, "second level
    // used" and "exact face rule" scalar: level constants come by scalar loads, branches are scalar.
    __shared__ float4 tab[PBR_SAMPLE_COUNT];
    __shared__ float2 lvl[PBR_SAMPLE_COUNT];
    uint32_t mip = 1;
    while (mip + 1 < pl.mips && blockIdx.x >= pl.first_block[mip + 1]) mip++;
    const uint32_t count = pl.count[mip];
    for (uint32_t i = threadIdx.x; i < count; i += PF_FOOT_BLOCK) {
        float4 e = tables[(size_t)mip * PBR_SAMPLE_COUNT + i];
        const float fl = floorf(e.w), f = e.w - fl;
        const uint32_t l0 = (uint32_t)fl, l1 = min(l0 + 1u, pl.sky_mips - 1u);
        const float w1 = f * e.z;
        // e.x == e.y == 0: L = N exactly (sample 0: H = N).  N is a texel CORNER, so for a whole row and column of every face it lies
        // exactly on a face edge, at the cube's corners on three faces at once: an exact tie, which v_cubeid breaks towards z, y, x
        // and the shader towards x, y, z.  On the coarsest source levels (1 x 1 texels) the two faces' footprints differ enough to
        // move a 2 x 2 output mip by 3 fp16 ULP (tools/prefilter_small_diag.py: 32^2 cube, mip 4, texel (0,0) of +X): such a sample
        // takes the shader's rule.
        const uint32_t word = l0 | (w1 != 0.0f && l1 != l0 ? PF_TWO_LEVELS : 0u) | (e.x == 0.0f && e.y == 0.0f ? PF_EXACT_FACE : 0u);
        lvl[i] = make_float2((1.0f - f) * e.z, w1);
        e.w = __builtin_bit_cast(float, word);
        tab[i] = e;
    }
    __syncthreads();
    const uint32_t s = pl.size >> mip;
    const uint32_t n = 6u * s * s;
    // XCD-aware block -> texel mapping, fp32 instance.  The dispatcher deals consecutive workgroups round-robin to the 8 XCDs, each with
    // its own 4 MB L2: with the plain mapping every XCD samples for texels all over the cube, i.e. needs the WHOLE source chain in its
    // L2 — the fp32 chain (33.6 MB; levels 1 + 2 alone are 8 MB) then thrashes: round 3 measured FETCH 3.1 GB for a 50 MB compulsory
    // footprint at a TCC hit rate of 85 %.  Here the blocks that share blockIdx % 8 — one XCD — take one CONTIGUOUS eighth of the mip's
    // texels, so an XCD's working set is the source region behind ~3/4 of a cube face: FETCH 17 MB, TCC hit 99.9 %
    // (profiles/r04_c_pmc_prefilter_f32.json), 2.34-2.39 -> 2.18-2.31 ms — the launch is bound by the texture addresser's data
    // return (TA busy 81 %: four 12-byte loads per level) once the misses are gone.  (Any consistent function of blockIdx % 8 keeps
    // the grouping; which physical XCD serves a group does not matter.)  The HALF instance keeps the plain order: its 16.8 MB chain
    // hits L2 at 99.5 % either way, and with the grouping the 128 blocks of an XCD contend for the same L2 channels — measured
    // 1.42-1.44 -> 1.51-1.56 ms (profiles/r04_c_prefilter_xcd_ab.txt).
    const uint32_t lb = blockIdx.x - pl.first_block[mip], nb = pl.first_block[mip + 1] - pl.first_block[mip];
    uint32_t block = lb;
    if (!HALF && pl.xcd_groups == 8u) {
        const uint32_t xcd = lb & 7u, chunk = nb >> 3, rem = nb & 7u;
        block = xcd * chunk + min(xcd, rem) + (lb >> 3);   // group x holds chunk + (x < rem) blocks; lb >> 3 < that count by construction
    }
    const uint32_t t = block * PF_FOOT_BLOCK + threadIdx.x;
    if (t >= n) return;
    const uint32_t x = t % s, y = (t / s) % s, face_o = t / (s * s);
    const float u = (float)x / (float)s, v = (float)y / (float)s;   // texel corner (Q8)
    const V3 N = normalize3_exact(cube_dir_raw(face_o, 2.0f * u - 1.0f, 2.0f * v - 1.0f));
    const V3 up = fabsf(N.z) < 0.999f ? v3(0.0f, 0.0f, 1.0f) : v3(1.0f, 0.0f, 0.0f);
    const V3 T = normalize3_exact(cross3(N, up));
    const V3 Bt = cross3(N, T);
    float ar = 0.0f, ag = 0.0f, ab = 0.0f, br = 0.0f, bg = 0.0f, bb = 0.0f;   // level-l0 and level-l1 halves of the sum: two chains
    for (uint32_t j = 0; j < count; j++) {
        const float4 e = tab[j];   // wave-uniform: one broadcast LDS read
        const float2 wl = lvl[j];
        const uint32_t word = (uint32_t)__builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, e.w));
        // L = T e.x + Bt e.y + N e.z as nine plain multiply-adds (they issue faster than the five packed ops of the unfused form)
        const V3 L = v3(__builtin_fmaf(T.x, e.x, __builtin_fmaf(Bt.x, e.y, N.x * e.z)), __builtin_fmaf(T.y, e.x, __builtin_fmaf(Bt.y, e.y, N.y * e.z)),
                        __builtin_fmaf(T.z, e.x, __builtin_fmaf(Bt.z, e.y, N.z * e.z)));
        float facef, cu, cv;
        if (word & PF_EXACT_FACE) {   // scalar branch
            uint32_t face;
            cube_face_uv(L, face, cu, cv);
            facef = (float)face;
        } else {
            const float hinv = rcp(fabsf(__builtin_amdgcn_cubema(L.x, L.y, L.z)));   // v_cubema = 2 x the major axis: (sc / ma + 1) / 2 = sc * hinv + 0.5
            const float sc = __builtin_amdgcn_cubesc(L.x, L.y, L.z), tc = __builtin_amdgcn_cubetc(L.x, L.y, L.z);
            facef = __builtin_amdgcn_cubeid(L.x, L.y, L.z);
            cu = __builtin_fmaf(sc, hinv, 0.5f);
            cv = __builtin_fmaf(tc, hinv, 0.5f);
        }
        const uint32_t l0 = word & 0xFFu;
        const PfLevel la = fo.lev[l0];
        if (HALF) pf_level_half(ar, ag, ab, foot, la, facef, cu, cv, wl.x);
        else pf_level_f32(ar, ag, ab, chain32, la, facef, cu, cv, wl.x);
        if (word & PF_TWO_LEVELS) {   // scalar branch
            const PfLevel lnext = fo.lev[l0 + 1u];
            if (HALF) pf_level_half(br, bg, bb, foot, lnext, facef, cu, cv, wl.y);
            else pf_level_f32(br, bg, bb, chain32, lnext, facef, cu, cv, wl.y);
        }
    }
    const float w = pl.wsum[mip];   // 0 samples -> 0/0 = NaN like the reference
    store_h4(out + 4 * (cube_mip_offset(pl.size, mip) + (size_t)t), f4((ar + br) / w, (ag + bg) / w, (ab + bb) / w, 1.0f));
}

// Padded copies of the WHOLE fp32 source chain in one launch: every face of every level with the 1-texel border the seamless rule
// selects, as float4 (mip 0's single fetch and the fp32 instance read it) and, when `half_dst` is given,
// as half4 at 8 bytes per texel: a bilinear footprint is two 16-byte loads, and the lanes of a wave — 64 neighbouring output texels
// taking the same sample — share the rows' cache lines.  (The shade's FOOTPRINT layout — four texels of a footprint stored together
// — was tried first: its fourfold duplication of every texel cost more L1 misses than its one-line-per-level saved here, FETCH
// 2.5 -> 4.5 GB per call; it suits the shade's incoherent gathers, not this kernel's coherent ones.)
// *lossy is set when a texel's rgb does not survive the conversion to half bit for bit.
// (One launch instead of two per level: the twenty small launches of a ten-level chain cost ~0.15 ms of gaps, see profiles/r03_n_*.)
struct PfPad {
    uint32_t first_block[17];   // first block of source level l, [mips] = total
    uint32_t src_off[16];       // texel offset of level l in the unpadded chain
    uint32_t dst_off[16];       // ... in the padded chains
    uint32_t size, mips;
};
__global__ __launch_bounds__(256) void k_cube_pad_chain(const float4* __restrict__ chain, float4* __restrict__ dst, pbr_half* __restrict__ half_dst, PfPad pp,
                                                          uint32_t* __restrict__ lossy) {
    uint32_t l = 0;
    while (l + 1 < pp.mips && blockIdx.x >= pp.first_block[l + 1]) l++;
    const int s = (int)(pp.size >> l), sp = s + 2;
    const uint32_t n = 6u * (uint32_t)sp * (uint32_t)sp;   // host-checked: the padded chain has < 2^32 texels
    const uint32_t t = (blockIdx.x - pp.first_block[l]) * 256u + threadIdx.x;
    bool lost = false;
    if (t < n) {
        const int xp = (int)(t % (uint32_t)sp), yp = (int)((t / (uint32_t)sp) % (uint32_t)sp);
        uint32_t face = t / ((uint32_t)sp * (uint32_t)sp);
        int x = xp - 1, y = yp - 1;
        const bool xo = (x < 0) | (x >= s), yo = (y < 0) | (y >= s);
        if (xo | yo) {   // same rule as pbr::cube_fetch_seamless / the oracle
            if (xo & yo) y = clampi(y, 0, s - 1);
            const float uu = 2.0f * ((float)x + 0.5f) / (float)s - 1.0f;
            const float vv = 2.0f * ((float)y + 0.5f) / (float)s - 1.0f;
            float u2, v2;
            cube_face_uv(cube_dir_raw(face, uu, vv), face, u2, v2);
            x = clampi((int)floorf(u2 * (float)s), 0, s - 1);
            y = clampi((int)floorf(v2 * (float)s), 0, s - 1);
        }
        const float4 c = chain[(size_t)pp.src_off[l] + ((size_t)face * s + y) * s + x];
        dst[(size_t)pp.dst_off[l] + t] = c;
        if (half_dst) {
            store_h4(half_dst + 4 * ((size_t)pp.dst_off[l] + t), f4(c.x, c.y, c.z, c.w));
            lost = (float)to_half_rn(c.x) != c.x || (float)to_half_rn(c.y) != c.y || (float)to_half_rn(c.z) != c.z;   // (NaN counts as lost)
        }
    }
    // one look at the flag per BLOCK at most, and no atomic once it is up: a lossy source would otherwise serialise one device-scope
    // access per wave (~25 000 for a 512^2 level 0) on one address — measured +0.4 ms as atomicOr, +0.09 ms as guarded loads per wave
    if (half_dst && __syncthreads_or(lost) && threadIdx.x == 0 && __hip_atomic_load(lossy, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) atomicOr(lossy, 1u);
}

// roughness 0: H = L = N for every sample, weight 1: the filtered value IS the bilinear fetch at the texel-corner
// direction (the reference's 1 024-fold running sum of one value differs from it by < 1e-4 relative, far inside the fp16 ULP)
__global__ __launch_bounds__(256) void k_prefilter_mip0(const float4* __restrict__ sky_padded, PfLaunch pl, pbr_half* __restrict__ out) {
    const uint32_t s = pl.size;
    const size_t n = (size_t)6 * s * s;
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    const uint32_t x = (uint32_t)(t % s), y = (uint32_t)((t / s) % s), face = (uint32_t)(t / ((size_t)s * s));
    const float u = (float)x / (float)s, v = (float)y / (float)s;
    const V3 N = normalize3_exact(cube_dir_raw(face, 2.0f * u - 1.0f, 2.0f * v - 1.0f));
    const V3 c = padded_trilinear(sky_padded, pl, N, 0.0f);
    store_h4(out + 4 * t, f4(c.x, c.y, c.z, 1.0f));
}

// host: the sample table of one output mip (the per-sample part of env_map_gen.hlsl:69-94 with V = N), libm in fp32
// exactly as the oracle evaluates it; returns the number of samples kept and their weight sum
static uint32_t build_prefilter_table(float roughness, uint32_t size, uint32_t sky_mips, float* tab4, float* wsum_out) {
    uint32_t n = 0;
    float wsum = 0.0f;
    const float maxl = (float)(sky_mips - 1);
    for (uint32_t i = 0; i < PBR_SAMPLE_COUNT; i++) {
        uint32_t bits = i;   // radical inverse (brdf.hlsli:101-109)
        bits = (bits << 16u) | (bits >> 16u);
        bits = ((bits & 0x55555555u) << 1u) | ((bits & 0xAAAAAAAAu) >> 1u);
        bits = ((bits & 0x33333333u) << 2u) | ((bits & 0xCCCCCCCCu) >> 2u);
        bits = ((bits & 0x0F0F0F0Fu) << 4u) | ((bits & 0xF0F0F0F0u) >> 4u);
        bits = ((bits & 0x00FF00FFu) << 8u) | ((bits & 0xFF00FF00u) >> 8u);
        const float xi_x = (float)i / (float)PBR_SAMPLE_COUNT, xi_y = (float)bits * 2.3283064365386963e-10f;
        const float a = roughness * roughness;
        const float phi = 6.28318530718f * xi_x;
        const float cos_theta = sqrtf((1.0f - xi_y) / (1.0f + (a * a - 1.0f) * xi_y));
        const float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
        const float hx = sin_theta * cosf(phi), hy = sin_theta * sinf(phi), hz = cos_theta;
        const float lz = 2.0f * hz * hz - 1.0f;   // N.L with V = N
        if (!(lz > 0.0f)) continue;
        const float NdotH = fmaxf(hz, 0.0f), HdotV = NdotH;
        const float t = (NdotH * NdotH) * (a * a - 1.0f) + 1.0f;
        const float D = a * a / fmaxf(3.14159265359f * t * t, 1e-6f);
        const float pdf = D * NdotH / (4.0f * HdotV + 0.0001f);
        const float texel_sa = 4.0f * 3.14159265359f / ((float)(6u * size * size));   // base size for every mip (Q8)
        const float sample_sa = 1.0f / ((float)PBR_SAMPLE_COUNT * pdf + 0.0001f);
        float lod = roughness == 0.0f ? 0.0f : 0.5f * log2f(sample_sa / texel_sa);
        if (!(lod == lod)) lod = 0.0f;
        lod = lod < 0.0f ? 0.0f : (lod > maxl ? maxl : lod);
        lod = floorf(lod * 256.0f + 0.5f) * (1.0f / 256.0f);   // D3D12_MIP_LOD_FRACTIONAL_BIT_COUNT = 8
        tab4[4 * n + 0] = 2.0f * hz * hx;
        tab4[4 * n + 1] = 2.0f * hz * hy;
        tab4[4 * n + 2] = lz;
        tab4[4 * n + 3] = lod;
        wsum += lz;
        n++;
    }
    *wsum_out = wsum;
    return n;
}

// ============================================================================ SH9 (a5)
// Stage 1: per-block partial sums of colour * Y_n(dir) * dOmega over all mip-0 texels
// (27 accumulators per thread -> wave shuffle reduce -> LDS -> one row of 27 per block).
// Stage 2: one block sums the rows in a fixed order in fp64 and applies SH.cpp:135-151 + the
// pack of SH.cpp:201-222.  Deterministic (no float atomics).
constexpr int SH_MAX_BLOCKS = 512;    // rows of the partial-sum table in the context's scratch area (27 floats each)

// wave-level sum in a FIXED order (xor butterfly: every lane ends with the same value)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Stage 1.  grid (ceil(size / 256), ceil(size / rows), 6), block 256: a thread owns ONE column x of a face and walks `rows`
// rows of it — no index division at all (u depends on x, v on the row: wave-uniform), every wave reads 1 KiB row segments,
// several 16-byte loads in flight per lane.  27 accumulators per thread -> butterfly over the wave -> LDS over the four waves
// -> one row of the partial table per block.
// Stage 2 (k_sh9_finish, one block): column sums of the table in fp64 in a fixed order, SH.cpp:140-151 / :204-219, the pack.
// (One launch with the LAST block finishing — ticket from a device-scope atomic behind a release fence — was built and
// measured: 47 us against 43 for the old pair.  An agent-scope fence on this part writes back / invalidates the XCD's L2,
// ~1 us a time, and every one of the ~400 blocks pays it; the kernel boundary does the same once.)
__global__ __launch_bounds__(256) void k_sh9_partial(const float* __restrict__ sky, uint32_t size, uint32_t rows, float* __restrict__ partial) {
    const uint32_t x = blockIdx.x * 256 + threadIdx.x, face = blockIdx.z;
    const uint32_t y0 = blockIdx.y * rows, y1 = min(y0 + rows, size);
    const uint32_t bid = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    float acc[27];
#pragma unroll
    for (int i = 0; i < 27; i++) acc[i] = 0.0f;
    if (x < size) {
        const float inv_size = 1.0f / (float)size;
        const float u = 2.0f * ((float)x + 0.5f) * inv_size - 1.0f;
        const float dw0 = 4.0f * inv_size * inv_size;
        const float4* col = reinterpret_cast<const float4*>(sky) + ((size_t)face * size) * size + x;
#pragma unroll 4
        for (uint32_t y = y0; y < y1; y++) {
            const float4 c = col[(size_t)y * size];
            const float v = 2.0f * ((float)y + 0.5f) * inv_size - 1.0f;
            const V3 raw = cube_dir_raw(face, u, v);
            const float inv = rsq(dot3(raw, raw));
            const V3 d = raw * inv;
            const float dw = dw0 * (inv * inv * inv);   // texel solid angle
            float Y[9];   // SH.cpp:6-37
            Y[0] = 0.282095f;
            Y[1] = 0.488603f * d.y;
            Y[2] = 0.488603f * d.z;
            Y[3] = 0.488603f * d.x;
            Y[4] = 1.092548f * d.x * d.y;
            Y[5] = 1.092548f * d.y * d.z;
            Y[6] = 0.315392f * (3.0f * d.z * d.z - 1.0f);
            Y[7] = 1.092548f * d.x * d.z;
            Y[8] = 0.546274f * (d.x * d.x - d.y * d.y);
            const float r = c.x * dw, g = c.y * dw, b = c.z * dw;
#pragma unroll
            for (int k = 0; k < 9; k++) {
                acc[k] += r * Y[k];
                acc[9 + k] += g * Y[k];
                acc[18 + k] += b * Y[k];
            }
        }
    }
    __shared__ float red[4][27];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 27; i++) {
        const float s = wave_sum(acc[i]);
        if (lane == 0) red[wave][i] = s;
    }
    __syncthreads();
    if (threadIdx.x < 27)
        partial[(size_t)bid * 27 + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// column sums of the nblocks x 27 table in fp64, fixed order: thread t takes rows t, t + 256, ... (27 independent loads per
// row, all in flight), then a butterfly over the wave and a fixed tree over the four waves
__global__ __launch_bounds__(256) void k_sh9_finish(const float* __restrict__ partial, uint32_t nblocks, float* __restrict__ out_pack) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __shared__ double dred[4][27];
    __shared__ float cfin[27];
    double sum[27];
#pragma unroll
    for (int i = 0; i < 27; i++) sum[i] = 0.0;
    for (uint32_t b = threadIdx.x; b < nblocks; b += 256) {
        const float* row = partial + (size_t)b * 27;
#pragma unroll
        for (int i = 0; i < 27; i++) sum[i] += (double)row[i];
    }
#pragma unroll
    for (int i = 0; i < 27; i++) {
        const double sfull = wave_sum_d(sum[i]);
        if (lane == 0) dred[wave][i] = sfull;
    }
    __syncthreads();
    const int t = threadIdx.x;
    if (t < 27) {
        const double sfull = (dred[0][t] + dred[1][t]) + (dred[2][t] + dred[3][t]);
        const int n = t % 9;
        const int l = n == 0 ? 0 : (n < 4 ? 1 : 2);
        // SH.cpp:140-151: c = InvPI * K * A * L, then * basis constant (SH.cpp:204-209)
        const float K = sqrtf(4.0f * PI_F / (float)(2 * l + 1));
        const float A = l == 0 ? sqrtf(PI_F) / 2.0f : (l == 1 ? sqrtf(PI_F / 3.0f) : sqrtf(5.0f * PI_F) / 8.0f);
        const float basis[9] = {0.282095f, 0.488603f, 0.488603f, 0.488603f, 1.092548f, 1.092548f, 0.315392f, 1.092548f, 0.546274f};
        const float v = INV_PI_F * K * A * (float)sfull;
        cfin[t] = v * basis[n];
    }
    __syncthreads();
    if (t < 3) {   // channel t: sha_* = (c3,c1,c2,c0), shb_* = (c4,c5,3*c6,c7)   (SH.cpp:213-218, Q16)
        const float* cc = cfin + 9 * t;
        float* sha = out_pack + 8 * t;
        float* shb = sha + 4;
        sha[0] = cc[3]; sha[1] = cc[1]; sha[2] = cc[2]; sha[3] = cc[0];
        shb[0] = cc[4]; shb[1] = cc[5]; shb[2] = cc[6] * 3.0f; shb[3] = cc[7];
    }
    if (t == 3) {   // shc = (c8r, c8g, c8b, 0)  (SH.cpp:219)
        out_pack[24] = cfin[8]; out_pack[25] = cfin[17]; out_pack[26] = cfin[26]; out_pack[27] = 0.0f;
    }
}

extern "C" {

pbr_status pbr_brdf_lut(pbr_ctx* ctx, uint32_t res, pbr_half* out_rg) {
    if (!ctx) return PBR_ERR_INVALID;
    PBR_REQUIRE(ctx, out_rg != nullptr, "pbr_brdf_lut: null output");
    PBR_REQUIRE(ctx, res >= 2 && res <= 8192, "pbr_brdf_lut: res must be in [2, 8192]");
    dim3 grid(res, (res + 255) / 256);
    hipLaunchKernelGGL(k_brdf_lut, grid, dim3(256), 0, ctx->stream, res, out_rg);
    return launched(ctx, "k_brdf_lut");
}

pbr_status pbr_cube_gen_mips(pbr_ctx* ctx, float* cube, uint32_t size, uint32_t mips) {
    if (!ctx) return PBR_ERR_INVALID;
    PBR_REQUIRE(ctx, cube != nullptr, "pbr_cube_gen_mips: null cube");
    PBR_REQUIRE(ctx, size >= 1 && size <= 8192 && (size & (size - 1)) == 0, "pbr_cube_gen_mips: size must be a power of two");
    PBR_REQUIRE(ctx, mips >= 1 && (size >> (mips - 1)) >= 1, "pbr_cube_gen_mips: too many mips");
    for (uint32_t m = 1; m < mips; m++) {
        uint32_t s = size >> m;
        size_t n = (size_t)6 * s * s;
        const float* src = cube + 4 * cube_mip_offset(size, m - 1);
        float* dst = cube + 4 * cube_mip_offset(size, m);
        hipLaunchKernelGGL(k_cube_downsample, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, src, dst, s);
        pbr_status r = launched(ctx, "k_cube_downsample");
        if (r) return r;
    }
    return PBR_OK;
}

pbr_status pbr_prefilter_env_mip(pbr_ctx* ctx, const pbr_cube_f32* sky, uint32_t size, uint32_t mip_level,
                                 float roughness, pbr_half* out_mip) {
    if (!ctx) return PBR_ERR_INVALID;
    PBR_REQUIRE(ctx, sky && sky->data && out_mip, "pbr_prefilter_env_mip: null pointer");
    PBR_REQUIRE(ctx, sky->size >= 1 && sky->mips >= 1 && (sky->size >> (sky->mips - 1)) >= 1, "pbr_prefilter_env_mip: bad sky cube");
    PBR_REQUIRE(ctx, size >= 1 && size <= 8192 && mip_level < 16 && (size >> mip_level) >= 1, "pbr_prefilter_env_mip: bad output size/mip");
    PBR_REQUIRE(ctx, roughness >= 0.0f && roughness <= 1.0f, "pbr_prefilter_env_mip: roughness outside [0,1]");
    const uint32_t s = size >> mip_level;
    const size_t n = (size_t)6 * s * s;
    hipLaunchKernelGGL(k_prefilter_env, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                       sky->data, sky->size, sky->mips, size, s, roughness, out_mip);
    return launched(ctx, "k_prefilter_env");
}

pbr_status pbr_prefilter_env(pbr_ctx* ctx, const pbr_cube_f32* sky, uint32_t size, uint32_t mips, pbr_half* out) {
    if (!ctx) return PBR_ERR_INVALID;
    PBR_REQUIRE(ctx, out != nullptr, "pbr_prefilter_env: null pointer");
    PBR_REQUIRE(ctx, size >= 1 && mips >= 1 && mips <= 16 && (size >> (mips - 1)) >= 1, "pbr_prefilter_env: bad output size/mips");
    PBR_REQUIRE(ctx, sky && sky->data && sky->size >= 1 && sky->mips >= 1 && (sky->size >> (sky->mips - 1)) >= 1, "pbr_prefilter_env: bad sky cube");
    PBR_REQUIRE(ctx, size <= 8192, "pbr_prefilter_env: bad output size/mips");
#ifdef PBR_DEBUG_KNOBS
    static const bool sequential = pbr::knob_set("PBR_PREFILTER_SEQ");   // A/B switch: the thread-per-texel kernel, all mips in one launch
    if (sequential) {
        size_t blocks = 0;
        for (uint32_t m = 0; m < mips; m++) blocks += ((size_t)6 * (size >> m) * (size >> m) + 255) / 256;
        PBR_REQUIRE(ctx, blocks <= 0x7FFFFFFFull, "pbr_prefilter_env: cube too large");
        hipLaunchKernelGGL(k_prefilter_env_all, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, sky->data, sky->size, sky->mips, size, mips, out);
        return launched(ctx, "k_prefilter_env_all");
    }
#endif
    // ---- padded source chain (stream-ordered scratch) + per-mip sample tables (kept on the device)
    PfLaunch pl{};
    pl.mips = mips; pl.size = size; pl.sky_size = sky->size; pl.sky_mips = sky->mips;
    static const int xcd_groups = pbr::knob_int("PBR_PREFILTER_XCD", 8);   // A/B switch (knobs build): 1 = the plain block order
    pl.xcd_groups = xcd_groups == 8 ? 8u : 1u;
    PBR_REQUIRE(ctx, sky->mips <= 16, "pbr_prefilter_env: bad sky cube");
    for (uint32_t l = 0; l < sky->mips; l++) pl.src_off[l] = (uint32_t)cube_border_mip_offset(sky->size, l);
    const size_t padded_texels = cube_border_mip_offset(sky->size, sky->mips);
    // the kernels address the padded chains with 32-bit byte offsets: sky cubes up to 4096^2 with a full mip chain (2.1 GiB padded);
    // larger ones go through pbr_prefilter_env_mip, one reference dispatch at a time
    PBR_REQUIRE(ctx, padded_texels * 16u < (1ull << 32), "pbr_prefilter_env: sky cube too large (padded fp32 chain must stay below 4 GiB; use pbr_prefilter_env_mip)");
    // which kernel samples mips >= 1: k_prefilter_foot on the half / fp32 padded chains.  Knobs build only: PBR_PREFILTER_WAVE=1 (the
    // wave-per-texel mapping) and PBR_PREFILTER_F32=1 (the round-2 kernel on the padded fp32 chain) stay selectable as A/B partners
    bool wave_per_texel = false, force_f32 = false;
#ifdef PBR_DEBUG_KNOBS
    static const bool k_wave = pbr::knob_set("PBR_PREFILTER_WAVE"), k_f32 = pbr::knob_set("PBR_PREFILTER_F32");
    wave_per_texel = k_wave; force_f32 = k_f32 && !k_wave;
#endif
    const bool use_foot = !force_f32 && !wave_per_texel;
    const uint32_t per_block = wave_per_texel ? (uint32_t)PF_TEXELS_PER_BLOCK : use_foot ? PF_FOOT_BLOCK : 256u;
    // The sample tables depend on (size, mips, sky mips) only: built on the host and uploaded when that key changes, then reused
    // (a renderer prefilters the same shapes again and again; the build is ~0.1 ms of libm and the upload a blocking copy).  A new key
    // takes a NEW device buffer — hipFree of the old one waits for whatever still reads it, whichever stream that is on.
    const size_t table_bytes = (size_t)mips * PBR_SAMPLE_COUNT * 16;
    if (!ctx->pf_dev || ctx->pf_key[0] != size || ctx->pf_key[1] != mips || ctx->pf_key[2] != sky->mips) {
        ctx->host_tmp.assign((size_t)mips * PBR_SAMPLE_COUNT * 4, 0.0f);
        for (uint32_t m = 1; m < mips; m++) {
            const float roughness = (float)m / (float)(mips - 1);   // DeferredPipeline.cpp:99
            ctx->pf_count[m] = build_prefilter_table(roughness, size, sky->mips, ctx->host_tmp.data() + (size_t)m * PBR_SAMPLE_COUNT * 4, &ctx->pf_wsum[m]);
        }
        if (ctx->pf_dev) { (void)hipFree(ctx->pf_dev); ctx->pf_dev = nullptr; }
        PBR_HIP(ctx, hipSetDevice(ctx->device));
        PBR_HIP(ctx, hipMalloc(&ctx->pf_dev, table_bytes));
        const hipError_t e = hipMemcpy(ctx->pf_dev, ctx->host_tmp.data(), table_bytes, hipMemcpyHostToDevice);   // blocking: host_tmp may be reused at once
        if (e != hipSuccess) { (void)hipFree(ctx->pf_dev); ctx->pf_dev = nullptr; return hip_fail(ctx, e, "prefilter tables"); }
        ctx->pf_key[0] = size; ctx->pf_key[1] = mips; ctx->pf_key[2] = sky->mips;
    }
    const float4* tables = reinterpret_cast<const float4*>(ctx->pf_dev);
    uint32_t blocks = 0;
    for (uint32_t m = 1; m < mips; m++) {
        pl.count[m] = ctx->pf_count[m];
        pl.wsum[m] = ctx->pf_wsum[m];
        pl.first_block[m] = blocks;
        const uint32_t sm = size >> m;
        blocks += (6u * sm * sm + per_block - 1) / per_block;
    }
    pl.first_block[mips] = blocks;
    const size_t foot_texels = padded_texels + 1;   // the half4 copy has the padded fp32 chain's layout (+ 1: the last row pair reads 8 bytes past a texel)
    // one stream-ordered block per call: [16 bytes: the "half copy is lossy" word][padded fp32 chain][padded half chain].  The flag
    // belongs to the CALL (round 3 kept one per context behind the cached tables: two calls of one context on different streams
    // could race on it — ADVICE r03); it is zeroed in stream order in front of the kernel that raises it.
    char* block = nullptr;
    PBR_HIP(ctx, hipMallocAsync((void**)&block, 16 + padded_texels * 16 + (use_foot ? foot_texels * 8 : 0), ctx->stream));
    uint32_t* lossy = reinterpret_cast<uint32_t*>(block);   // 1 = the half copy is not exact
    float4* padded = reinterpret_cast<float4*>(block + 16);
    pbr_half* foot = use_foot ? reinterpret_cast<pbr_half*>(padded + padded_texels) : nullptr;
    pbr_status r = PBR_OK;
    {
        const hipError_t e = hipMemsetAsync(lossy, 0, 16, ctx->stream);
        if (e != hipSuccess) r = hip_fail(ctx, e, "pbr_prefilter_env: flag reset");
    }
    PfLevels fo{};
    PfPad pp{};
    pp.size = sky->size; pp.mips = sky->mips;
    uint64_t pad_blocks = 0;
    for (uint32_t l = 0; l < sky->mips; l++) {
        const uint32_t sl = sky->size >> l;
        fo.lev[l] = PfLevel{pl.src_off[l], sl + 2u, (float)sl * 256.0f, (float)(sl + 2u)};
        pp.first_block[l] = (uint32_t)pad_blocks;
        pp.src_off[l] = (uint32_t)cube_mip_offset(sky->size, l);
        pp.dst_off[l] = pl.src_off[l];
        pad_blocks += ((uint64_t)6 * (sl + 2) * (sl + 2) + 255) / 256;
    }
    pp.first_block[sky->mips] = (uint32_t)pad_blocks;
    if (r == PBR_OK && pad_blocks > 0x7FFFFFFFull) r = pbr::fail(ctx, PBR_ERR_INVALID, "pbr_prefilter_env: sky cube too large");
    if (r == PBR_OK) {
        hipLaunchKernelGGL(k_cube_pad_chain, dim3((unsigned)pad_blocks), dim3(256), 0, ctx->stream, reinterpret_cast<const float4*>(sky->data), padded, foot, pp, lossy);
        r = launched(ctx, "k_cube_pad_chain");
    }
    if (r == PBR_OK && blocks) {
#ifdef PBR_DEBUG_KNOBS
        if (wave_per_texel) hipLaunchKernelGGL(k_prefilter_fast, dim3(blocks), dim3(256), 0, ctx->stream, padded, tables, pl, out);
        else if (force_f32) hipLaunchKernelGGL(k_prefilter_tex, dim3(blocks), dim3(256), 0, ctx->stream, padded, tables, pl, out);
        else
#endif
        {   // both instances; the flag k_cube_pad_chain wrote picks the one that works, the other returns at once
            hipLaunchKernelGGL(k_prefilter_foot<true>, dim3(blocks), dim3(PF_FOOT_BLOCK), 0, ctx->stream, (const void*)foot, fo, tables, pl, out, lossy);
            r = launched(ctx, "k_prefilter_foot<half>");
            if (r == PBR_OK) hipLaunchKernelGGL(k_prefilter_foot<false>, dim3(blocks), dim3(PF_FOOT_BLOCK), 0, ctx->stream, (const void*)padded, fo, tables, pl, out, lossy);
        }
        if (r == PBR_OK) r = launched(ctx, "k_prefilter_*");
    }
    if (r == PBR_OK) {
        const size_t n0 = (size_t)6 * size * size;
        hipLaunchKernelGGL(k_prefilter_mip0, dim3((unsigned)((n0 + 255) / 256)), dim3(256), 0, ctx->stream, padded, pl, out);
        r = launched(ctx, "k_prefilter_mip0");
    }
    (void)hipFreeAsync(block, ctx->stream);
    return r;
}

pbr_status pbr_sh9_project(pbr_ctx* ctx, const pbr_cube_f32* sky, float* out_pack) {
    if (!ctx) return PBR_ERR_INVALID;
    PBR_REQUIRE(ctx, sky && sky->data && out_pack, "pbr_sh9_project: null pointer");
    PBR_REQUIRE(ctx, sky->size >= 1 && sky->size <= 8192, "pbr_sh9_project: bad cube size");
    const uint32_t size = sky->size, cols = (size + 255) / 256;
    // rows per block: as few as keep the partial table within SH_MAX_BLOCKS rows (512^2: 16 rows -> 2 x 32 x 6 = 384 blocks)
    uint32_t rows = 1;
    while ((uint64_t)cols * ((size + rows - 1) / rows) * 6 > (uint64_t)SH_MAX_BLOCKS) rows *= 2;
    const uint32_t gy = (size + rows - 1) / rows;
    PBR_REQUIRE(ctx, (size_t)SH_MAX_BLOCKS * 27 * sizeof(float) <= ctx->scratch_bytes, "pbr_sh9_project: scratch too small");
    float* partial = (float*)ctx->scratch;
    hipLaunchKernelGGL(k_sh9_partial, dim3(cols, gy, 6), dim3(256), 0, ctx->stream, sky->data, size, rows, partial);
    pbr_status r = launched(ctx, "k_sh9_partial");
    if (r) return r;
    hipLaunchKernelGGL(k_sh9_finish, dim3(1), dim3(256), 0, ctx->stream, partial, cols * gy * 6u, out_pack);
    return launched(ctx, "k_sh9_finish");
}

}  // extern "C"
