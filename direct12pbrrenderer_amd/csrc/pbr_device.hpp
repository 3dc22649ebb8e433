// pbr_device.hpp — device-side helpers shared by the gfx950 kernels of the deferred-PBR path.
// References are to /root/reference (zrlhahaha/Direct12PBRRenderer), "Shader/" =
// DeferredRendering/Shader/.  Nothing here is shared with oracle/ — the oracle is an
// independent CPU restatement; this file is the product's own arithmetic.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/pbr_hip.h"

namespace pbr {

// global.hlsli:4-7
constexpr float PI_F = 3.14159265359f;
constexpr float INV_PI_F = 0.31830988618f;
constexpr float EPSILON_F = 1e-6f;
constexpr float TWO_PI_F = (float)(2.0 * 3.14159265359);

typedef _Float16 h16;

struct V3 { float x, y, z; };
__device__ __forceinline__ V3 v3(float x, float y, float z) { return V3{x, y, z}; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator*(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ float dot3(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross3(V3 a, V3 b) {
    return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
// 1/sqrt: v_rsq_f32 (1 ulp) — fine inside the 1e-4 tolerance of the float paths
__device__ __forceinline__ float rsq(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ V3 normalize3(V3 v) { return v * rsq(dot3(v, v)); }
// IEEE-exact variant (correctly rounded sqrt + divide) for the places whose result is thresholded
__device__ __forceinline__ V3 normalize3_exact(V3 v) { return v * (1.0f / sqrtf(dot3(v, v))); }
__device__ __forceinline__ float saturatef(float x) { return __builtin_amdgcn_fmed3f(x, 0.0f, 1.0f); }  // NaN -> 0
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return min(max(v, lo), hi); }

struct F4 { float x, y, z, w; };
__device__ __forceinline__ F4 f4(float x, float y, float z, float w) { return F4{x, y, z, w}; }
__device__ __forceinline__ F4 operator+(F4 a, F4 b) { return f4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ F4 operator*(F4 a, float s) { return f4(a.x * s, a.y * s, a.z * s, a.w * s); }

// ---- half4 texel access (R16G16B16A16_FLOAT) : one 8-byte load/store per texel -------------
struct alignas(8) H4 { h16 x, y, z, w; };
__device__ __forceinline__ F4 load_h4(const pbr_half* p) {
    H4 h = *reinterpret_cast<const H4*>(p);
    return f4((float)h.x, (float)h.y, (float)h.z, (float)h.w);
}
// fp32 -> fp16 as the render-target write does it: the fp32 value is rounded to nearest-even half.
// The asm barrier pins the fp32 value: without it hipcc may fold a preceding fma into
// v_fma_mixlo_f16 (ONE rounding straight to half), which differs from fp32-then-half by an ulp.
__device__ __forceinline__ h16 to_half_rn(float v) {
    asm volatile("" : "+v"(v));
    return (h16)v;   // v_cvt_f16_f32: RNE, overflow -> inf
}
__device__ __forceinline__ void store_h4(pbr_half* p, F4 v) {
    H4 h;
    h.x = to_half_rn(v.x); h.y = to_half_rn(v.y); h.z = to_half_rn(v.z); h.w = to_half_rn(v.w);
    *reinterpret_cast<H4*>(p) = h;
}
struct alignas(4) H2 { h16 x, y; };

// ---- software samplers (sampler s3 LinearClamp / s1 PointClamp, D3D12Device.cpp:665-684) -----
struct BilinearCoord { int i0, i1; float f; };
// D3D fixed-function addressing: u*size snapped to x.8 fixed point (round to nearest,
// D3D12_SUBTEXEL_FRACTIONAL_BIT_COUNT = 8), then the half-texel offset; weight = multiple of 1/256
__device__ __forceinline__ float snap8(float x) { return floorf(x * 256.0f + 0.5f) * (1.0f / 256.0f); }
__device__ __forceinline__ BilinearCoord bilinear_coord(float u, int size) {
    float c = u * (float)size;
    c = (c == c) ? c : 0.5f;
    c = fminf(fmaxf(c, -1.5f), (float)size + 1.5f);
    float x = snap8(c) - 0.5f;
    float fl = floorf(x);
    BilinearCoord b;
    b.i0 = (int)fl;
    b.i1 = b.i0 + 1;
    b.f = x - fl;
    return b;
}
// a*s + b with one rounding — written explicitly so the result does not depend on -ffp-contract
__device__ __forceinline__ F4 fma4(F4 a, float s, F4 b) {
    return f4(__builtin_fmaf(a.x, s, b.x), __builtin_fmaf(a.y, s, b.y), __builtin_fmaf(a.z, s, b.z), __builtin_fmaf(a.w, s, b.w));
}
// sampler lerps: the far tap is fused onto the weighted near tap (same operation order as the oracle); a tap
// with weight exactly 0 does not contribute (a sample at a texel centre is that texel, inf/NaN neighbours or not)
__device__ __forceinline__ F4 lerp4(F4 a, F4 b, float f) {
    const F4 r = fma4(b, f, a * (1.0f - f));
    return f == 0.0f ? a : r;
}
__device__ __forceinline__ F4 bilerp(F4 c00, F4 c10, F4 c01, F4 c11, float fx, float fy) {
    return lerp4(lerp4(c00, c10, fx), lerp4(c01, c11, fx), fy);
}
// Texture2D<half4>.SampleLevel(LinearClamp, uv, 0)
__device__ __forceinline__ F4 sample_2d_h4(const pbr_half* img, int w, int h, int pitch, float u, float v) {
    BilinearCoord cx = bilinear_coord(u, w), cy = bilinear_coord(v, h);
    int x0 = clampi(cx.i0, 0, w - 1), x1 = clampi(cx.i1, 0, w - 1);
    int y0 = clampi(cy.i0, 0, h - 1), y1 = clampi(cy.i1, 0, h - 1);
    F4 c00 = load_h4(img + 4 * ((size_t)y0 * pitch + x0));
    F4 c10 = load_h4(img + 4 * ((size_t)y0 * pitch + x1));
    F4 c01 = load_h4(img + 4 * ((size_t)y1 * pitch + x0));
    F4 c11 = load_h4(img + 4 * ((size_t)y1 * pitch + x1));
    return bilerp(c00, c10, c01, c11, cx.f, cy.f);
}

// ---- cube addressing: env_map_gen.hlsl:20-44 and its D3D inverse -----------------------------
__device__ __forceinline__ V3 cube_dir_raw(uint32_t face, float u, float v) {
    switch (face) {
        case 0: return v3(1.0f, -v, -u);
        case 1: return v3(-1.0f, -v, u);
        case 2: return v3(u, 1.0f, v);
        case 3: return v3(u, -1.0f, -v);
        case 4: return v3(u, -v, 1.0f);
        default: return v3(-u, -v, -1.0f);
    }
}
__device__ __forceinline__ void cube_face_uv(V3 d, uint32_t& face, float& u, float& v) {
    float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z);
    float sc, tc, ma;
    if (ax >= ay && ax >= az) {
        ma = ax;
        if (d.x >= 0.0f) { face = 0; sc = -d.z; tc = -d.y; }
        else             { face = 1; sc = d.z;  tc = -d.y; }
    } else if (ay >= az) {
        ma = ay;
        if (d.y >= 0.0f) { face = 2; sc = d.x; tc = d.z; }
        else             { face = 3; sc = d.x; tc = -d.z; }
    } else {
        ma = az;
        if (d.z >= 0.0f) { face = 4; sc = d.x;  tc = -d.y; }
        else             { face = 5; sc = -d.x; tc = -d.y; }
    }
    float inv = 1.0f / ma;   // IEEE divide: u,v feed floor()
    u = (sc * inv + 1.0f) * 0.5f;
    v = (tc * inv + 1.0f) * 0.5f;
}
__host__ __device__ __forceinline__ size_t cube_mip_offset(uint32_t size, uint32_t mip) {
    size_t off = 0;
    for (uint32_t m = 0; m < mip; m++) { size_t s = size >> m; off += 6 * s * s; }
    return off;
}

// border layout (the prefilter's padded fp32 source): every face of every mip with a 1-texel border, (s+2)^2 texels
__host__ __device__ __forceinline__ size_t cube_border_mip_offset(uint32_t size, uint32_t mip) {
    size_t off = 0;
    for (uint32_t m = 0; m < mip; m++) { size_t s = (size >> m) + 2; off += 6 * s * s; }
    return off;
}
// FOOTPRINT layout (the shade's env chain, pbr_env_pad): for every bilinear footprint origin (x, y) in [-1, s-1]^2 of
// every face of every mip the four texels (x,y), (x+1,y), (x,y+1), (x+1,y+1) — already resolved by the seamless rule —
// stored together: 32 contiguous bytes, ONE cache line per trilinear level instead of two rows 4 KB apart.  4 x the
// plain chain (67 MB at 512^2 x 5): HBM capacity is what this part has to spare; the per-pixel IBL gathers with random
// reflection vectors are bound by the lines they pull, not by arithmetic.
__host__ __device__ __forceinline__ size_t env_padded_mip_offset(uint32_t size, uint32_t mip) {
    size_t off = 0;
    for (uint32_t m = 0; m < mip; m++) { size_t s = (size >> m) + 1; off += 6 * s * s * 4; }
    return off;
}

// Seamless edge rule (same definition as the oracle): a tap outside the face is re-projected
// onto the neighbouring face through the direction of its texel centre; a tap that leaves the
// face in both axes is clamped in y first.
template <class Texel>
__device__ __forceinline__ F4 cube_fetch_seamless(int s, uint32_t face, int x, int y, const Texel& texel) {
    bool xo = (x < 0) | (x >= s), yo = (y < 0) | (y >= s);
    if (xo | yo) {
        if (xo & yo) y = clampi(y, 0, s - 1);
        float uu = 2.0f * ((float)x + 0.5f) / (float)s - 1.0f;
        float vv = 2.0f * ((float)y + 0.5f) / (float)s - 1.0f;
        V3 d = cube_dir_raw(face, uu, vv);
        float u2, v2;
        cube_face_uv(d, face, u2, v2);
        x = clampi((int)floorf(u2 * (float)s), 0, s - 1);
        y = clampi((int)floorf(v2 * (float)s), 0, s - 1);
    }
    return texel(face, x, y);
}
template <class Texel>
__device__ __forceinline__ F4 cube_bilinear(int s, V3 dir, const Texel& texel) {
    uint32_t face; float u, v;
    cube_face_uv(dir, face, u, v);
    BilinearCoord cx = bilinear_coord(u, s), cy = bilinear_coord(v, s);
    F4 c00 = cube_fetch_seamless(s, face, cx.i0, cy.i0, texel);
    F4 c10 = cube_fetch_seamless(s, face, cx.i1, cy.i0, texel);
    F4 c01 = cube_fetch_seamless(s, face, cx.i0, cy.i1, texel);
    F4 c11 = cube_fetch_seamless(s, face, cx.i1, cy.i1, texel);
    return bilerp(c00, c10, c01, c11, cx.f, cy.f);
}

struct CubeTexelF32 {
    const float* base; int s;
    __device__ __forceinline__ F4 operator()(uint32_t f, int x, int y) const {
        const float4 v = *reinterpret_cast<const float4*>(base + 4 * (((size_t)f * s + y) * s + x));
        return f4(v.x, v.y, v.z, v.w);
    }
};
struct CubeTexelF16 {
    const pbr_half* base; int s;
    __device__ __forceinline__ F4 operator()(uint32_t f, int x, int y) const {
        return load_h4(base + 4 * (((size_t)f * s + y) * s + x));
    }
};
// TextureCube.SampleLevel(LinearClamp, dir, lod): trilinear, lod clamped to [0, mips-1]
template <class TexelT, class Ptr>
__device__ __forceinline__ F4 cube_trilinear(Ptr data, uint32_t size, uint32_t mips, V3 dir, float lod) {
    float maxl = (float)(mips - 1);
    lod = (lod == lod) ? lod : 0.0f;
    lod = snap8(fminf(fmaxf(lod, 0.0f), maxl));   // D3D12_MIP_LOD_FRACTIONAL_BIT_COUNT = 8
    float fl = floorf(lod);
    uint32_t l0 = (uint32_t)fl;
    uint32_t l1 = min(l0 + 1, mips - 1);
    float f = lod - fl;
    TexelT t0{data + 4 * cube_mip_offset(size, l0), (int)(size >> l0)};
    F4 a = cube_bilinear((int)(size >> l0), dir, t0);
    if (f == 0.0f || l1 == l0) return a;
    TexelT t1{data + 4 * cube_mip_offset(size, l1), (int)(size >> l1)};
    F4 b = cube_bilinear((int)(size >> l1), dir, t1);
    return fma4(b, f, a * (1.0f - f));
}

// ---- brdf.hlsli:71-114 --------------------------------------------------------------------------
__device__ __forceinline__ float radical_inverse_vdc(uint32_t bits) {
    return (float)__brev(bits) * 2.3283064365386963e-10f;   // the 5 swap steps == 32-bit bit reversal
}
// precise version (sinf/cosf/sqrtf from OCML): used to build per-sample tables once per block
__device__ inline V3 ggx_important_sample(float roughness, V3 normal, float xi_x, float xi_y) {
    float a = roughness * roughness;
    float phi = TWO_PI_F * xi_x;
    float cos_theta = sqrtf((1.0f - xi_y) / (1.0f + (a * a - 1.0f) * xi_y));
    float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
    V3 h = v3(sin_theta * cosf(phi), sin_theta * sinf(phi), cos_theta);
    V3 up = fabsf(normal.z) < 0.999f ? v3(0.0f, 0.0f, 1.0f) : v3(1.0f, 0.0f, 0.0f);
    V3 tangent = normalize3_exact(cross3(normal, up));
    V3 bitangent = cross3(normal, tangent);
    return normalize3_exact(tangent * h.x + bitangent * h.y + normal * h.z);
}
__device__ __forceinline__ float distribution_ggx(float NdotH, float roughness) {   // brdf.hlsli:6-11
    float a = roughness * roughness;
    float t = (NdotH * NdotH) * (a * a - 1.0f) + 1.0f;
    return a * a / fmaxf(PI_F * t * t, EPSILON_F);
}

__device__ __forceinline__ float luminance(float r, float g, float b) {   // global.hlsli:140-143
    return r * 0.2126f + g * 0.7152f + b * 0.0722f;
}

// ---- one histogram count per active lane into a per-wave LDS histogram ---------------------------------------------
// LDS atomics of one wave on the SAME address serialise (up to 64 deep), and neighbouring pixels of a rendered frame
// mostly fall into the same few luminance bins.  Two rounds of leader aggregation first: the lowest pending lane's
// bin is broadcast, every lane with that bin is counted by ballot and the leader adds the popcount once; lanes still
// pending afterwards (noise-like content) add their own 1.  Same counts, whatever the content.
__device__ __forceinline__ void hist_count(uint32_t* wave_hist, uint32_t bin, bool valid) {
    const uint32_t lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    unsigned long long todo = __ballot(valid);
#pragma unroll
    for (int r = 0; r < 2; r++) {
        if (todo == 0ull) return;   // wave-uniform
        const int leader = __ffsll((long long)todo) - 1;
        const uint32_t b = (uint32_t)__builtin_amdgcn_readlane((int)bin, leader);
        const unsigned long long m = __ballot(valid && bin == b) & todo;
        if ((int)lane == leader) atomicAdd(&wave_hist[b], (uint32_t)__popcll(m));
        todo &= ~m;
    }
    if (valid && ((todo >> lane) & 1ull)) atomicAdd(&wave_hist[bin], 1u);
}

}  // namespace pbr
