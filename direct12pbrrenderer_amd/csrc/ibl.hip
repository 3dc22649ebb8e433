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
__global__ __launch_bounds__(256) void k_brdf_lut(uint32_t res, pbr_half* __restrict__ out) {
    __shared__ float4 tab[PBR_SAMPLE_COUNT];   // normalized H in the N=(0,0,1) frame
    const uint32_t x = blockIdx.x;
    const float roughness = (float)x / (float)(res - 1);
    for (uint32_t i = threadIdx.x; i < PBR_SAMPLE_COUNT; i += 256) {
        float xi_x = (float)i / (float)PBR_SAMPLE_COUNT;
        float xi_y = radical_inverse_vdc(i);
        V3 H = ggx_important_sample(roughness, v3(0.0f, 0.0f, 1.0f), xi_x, xi_y);
        tab[i] = make_float4(H.x, H.y, H.z, 0.0f);
    }
    __syncthreads();
    const uint32_t y = blockIdx.y * 256 + threadIdx.x;
    if (y >= res) return;
    const float NdotV = (float)(y + 1) / (float)res;
    const float Vx = sqrtf(1.0f - NdotV * NdotV), Vz = NdotV;
    const float k = roughness * roughness / 2.0f;   // Q6: k = r^2/2 in the LUT
    const float one_k = 1.0f - k;
    const float gv = NdotV / fmaxf(NdotV * one_k + k, EPSILON_F);
    float A = 0.0f, B = 0.0f;
#pragma unroll 4
    for (uint32_t i = 0; i < PBR_SAMPLE_COUNT; i++) {
        const float4 H = tab[i];
        const float VdH = Vx * H.x + Vz * H.z;   // V.y == 0
        const float t2 = 2.0f * VdH;
        const float Lx = t2 * H.x - Vx, Ly = t2 * H.y, Lz = t2 * H.z - Vz;
        const float invl = 1.0f / sqrtf(Lx * Lx + Ly * Ly + Lz * Lz);
        const float NdotL = fmaxf(Lz * invl, 0.0f);
        const float NdotH = fmaxf(H.z, 0.0f);
        const float VdotH = fmaxf(VdH, 0.0f);
        if (NdotL > 0.0f) {
            const float omv = 1.0f - VdotH;
            const float o2 = omv * omv;
            const float Fc = o2 * o2 * omv;
            const float gl = NdotL / fmaxf(NdotL * one_k + k, EPSILON_F);
            const float G = gv * gl;
            const float G_Vis = (G * VdotH) / fmaxf(NdotH * NdotV, 0.0001f);
            A += (1.0f - Fc) * G_Vis;
            B += Fc * G_Vis;
        }
    }
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

// ============================================================================ SH9 (a5)
// Stage 1: per-block partial sums of colour * Y_n(dir) * dOmega over all mip-0 texels
// (27 accumulators per thread -> wave shuffle reduce -> LDS -> one row of 27 per block).
// Stage 2: one block sums the rows in a fixed order in fp64 and applies SH.cpp:135-151 + the
// pack of SH.cpp:201-222.  Deterministic (no float atomics).
constexpr int SH_BLOCKS = 512;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

__global__ __launch_bounds__(256) void k_sh9_partial(const float* __restrict__ sky, uint32_t size, float* __restrict__ partial) {
    const size_t n = (size_t)6 * size * size;
    float acc[27];
#pragma unroll
    for (int i = 0; i < 27; i++) acc[i] = 0.0f;
    for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < n; t += (size_t)gridDim.x * 256) {
        const uint32_t x = (uint32_t)(t % size), y = (uint32_t)((t / size) % size), f = (uint32_t)(t / ((size_t)size * size));
        const float u = 2.0f * ((float)x + 0.5f) / (float)size - 1.0f;
        const float v = 2.0f * ((float)y + 0.5f) / (float)size - 1.0f;
        const V3 raw = cube_dir_raw(f, u, v);
        const float r2 = dot3(raw, raw);
        const float inv = 1.0f / sqrtf(r2);
        const V3 d = raw * inv;
        const float dw = (4.0f / ((float)size * (float)size)) * (inv * inv * inv);   // texel solid angle
        const float4 c = reinterpret_cast<const float4*>(sky)[t];
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
    __shared__ float red[4][27];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 27; i++) {
        float s = wave_sum(acc[i]);
        if (lane == 0) red[wave][i] = s;
    }
    __syncthreads();
    if (threadIdx.x < 27) {
        partial[(size_t)blockIdx.x * 27 + threadIdx.x] =
            (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    }
}

// 27 coefficients x 8 interleaved fp64 partial sums (rows j, j+8, ...), combined in a fixed tree: deterministic, and
// eight times shorter than one serial chain of dependent loads per coefficient
__global__ __launch_bounds__(256) void k_sh9_finish(const float* __restrict__ partial, int nblocks, float* __restrict__ out_pack) {
    __shared__ float c[27];
    __shared__ double part[8][27];
    const int t = threadIdx.x;
    if (t < 216) {
        const int coef = t % 27, j = t / 27;
        double s = 0.0;
        for (int b = j; b < nblocks; b += 8) s += (double)partial[(size_t)b * 27 + coef];
        part[j][coef] = s;
    }
    __syncthreads();
    if (t < 27) {
        const double s = ((part[0][t] + part[1][t]) + (part[2][t] + part[3][t])) + ((part[4][t] + part[5][t]) + (part[6][t] + part[7][t]));
        const int n = t % 9;
        const int l = n == 0 ? 0 : (n < 4 ? 1 : 2);
        // SH.cpp:140-151: c = InvPI * K * A * L, then * basis constant (SH.cpp:204-209)
        const float K = sqrtf(4.0f * PI_F / (float)(2 * l + 1));
        const float A = l == 0 ? sqrtf(PI_F) / 2.0f : (l == 1 ? sqrtf(PI_F / 3.0f) : sqrtf(5.0f * PI_F) / 8.0f);
        const float basis[9] = {0.282095f, 0.488603f, 0.488603f, 0.488603f, 1.092548f, 1.092548f, 0.315392f, 1.092548f, 0.546274f};
        float v = INV_PI_F * K * A * (float)s;
        c[t] = v * basis[n];
    }
    __syncthreads();
    if (t < 3) {   // channel t: sha_* = (c3,c1,c2,c0), shb_* = (c4,c5,3*c6,c7)   (SH.cpp:213-218, Q16)
        const float* cc = c + 9 * t;
        float* sha = out_pack + 8 * t;
        float* shb = sha + 4;
        sha[0] = cc[3]; sha[1] = cc[1]; sha[2] = cc[2]; sha[3] = cc[0];
        shb[0] = cc[4]; shb[1] = cc[5]; shb[2] = cc[6] * 3.0f; shb[3] = cc[7];
    }
    if (t == 3) {   // shc = (c8r, c8g, c8b, 0)  (SH.cpp:219)
        out_pack[24] = c[8]; out_pack[25] = c[17]; out_pack[26] = c[26]; out_pack[27] = 0.0f;
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
    size_t blocks = 0;
    for (uint32_t m = 0; m < mips; m++) blocks += ((size_t)6 * (size >> m) * (size >> m) + 255) / 256;
    PBR_REQUIRE(ctx, blocks <= 0x7FFFFFFFull, "pbr_prefilter_env: cube too large");
    hipLaunchKernelGGL(k_prefilter_env_all, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, sky->data, sky->size, sky->mips, size, mips, out);
    return launched(ctx, "k_prefilter_env_all");
}

pbr_status pbr_sh9_project(pbr_ctx* ctx, const pbr_cube_f32* sky, float* out_pack) {
    if (!ctx) return PBR_ERR_INVALID;
    PBR_REQUIRE(ctx, sky && sky->data && out_pack, "pbr_sh9_project: null pointer");
    PBR_REQUIRE(ctx, sky->size >= 1 && sky->size <= 8192, "pbr_sh9_project: bad cube size");
    size_t n = (size_t)6 * sky->size * sky->size;
    int blocks = (int)((n + 255) / 256);
    if (blocks > SH_BLOCKS) blocks = SH_BLOCKS;
    PBR_REQUIRE(ctx, (size_t)blocks * 27 * sizeof(float) <= ctx->scratch_bytes, "pbr_sh9_project: scratch too small");
    float* partial = (float*)ctx->scratch;
    hipLaunchKernelGGL(k_sh9_partial, dim3(blocks), dim3(256), 0, ctx->stream, sky->data, sky->size, partial);
    pbr_status r = launched(ctx, "k_sh9_partial");
    if (r) return r;
    hipLaunchKernelGGL(k_sh9_finish, dim3(1), dim3(256), 0, ctx->stream, partial, blocks, out_pack);
    return launched(ctx, "k_sh9_finish");
}

}  // extern "C"
