/*
 * pbr_oracle.cpp — CPU restatement of the reference's deferred-PBR shading arithmetic.
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (see pbr_oracle.h for both statements and the
 * arithmetic model).  All citations are relative to /root/reference
 * (zrlhahaha/Direct12PBRRenderer): "Shader/" = DeferredRendering/Shader/.
 *
 * Build: see oracle/Makefile (g++ -O2 -ffp-contract=off -fopenmp, no fast-math).
 */
#include "pbr_oracle.h"

#include <cmath>
#include <cstring>
#include <cstdlib>
#include <random>
#include <vector>
#include <algorithm>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

// global.hlsli:4-7
constexpr float PI_F = 3.14159265359f;
constexpr float INV_PI_F = 0.31830988618f;
constexpr float EPSILON_F = 1e-6f;
constexpr float TWO_PI_F = (float)(2.0 * 3.14159265359);

struct V3 { float x, y, z; };
static inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
static inline V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline V3 operator*(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
static inline V3 operator/(V3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
static inline float dot3(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline V3 cross3(V3 a, V3 b) {
    return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline V3 normalize3(V3 v) {
    float inv = 1.0f / sqrtf(dot3(v, v));
    return v * inv;
}
static inline float saturate(float x) {
    if (!(x == x)) return 0.0f;  // HLSL saturate(NaN) = 0
    return x < 0.0f ? 0.0f : (x > 1.0f ? 1.0f : x);
}
static inline float lerpf(float a, float b, float t) { return a + t * (b - a); }

// ------------------------------------------------------------------ fp16 (RNE, overflow->inf)
static inline uint16_t f32_to_f16(float f) {
    uint32_t x;
    std::memcpy(&x, &f, 4);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t ax = x & 0x7FFFFFFFu;
    if (ax >= 0x7F800000u) {  // inf / nan
        return (uint16_t)(sign | 0x7C00u | ((ax > 0x7F800000u) ? (0x0200u | ((ax >> 13) & 0x3FFu)) : 0u));
    }
    if (ax >= 0x477FF000u) {  // >= 65520 rounds to inf
        return (uint16_t)(sign | 0x7C00u);
    }
    if (ax < 0x38800000u) {   // subnormal half or zero (|f| < 2^-14)
        if (ax < 0x33000000u) return (uint16_t)sign;  // < 2^-25 -> 0 (2^-25 itself ties to even -> 0)
        uint32_t e = ax >> 23;
        uint32_t m = (ax & 0x7FFFFFu) | 0x800000u;
        uint32_t shift = 126u - e;            // 14..24
        uint32_t half_m = m >> shift;
        uint32_t rem = m & ((1u << shift) - 1u);
        uint32_t halfway = 1u << (shift - 1u);
        if (rem > halfway || (rem == halfway && (half_m & 1u))) half_m++;
        return (uint16_t)(sign | half_m);
    }
    uint32_t e = (ax >> 23) - 112u;
    uint32_t m = ax & 0x7FFFFFu;
    uint32_t h = (e << 10) | (m >> 13);
    uint32_t rem = m & 0x1FFFu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) h++;
    return (uint16_t)(sign | h);
}
static inline float f16_to_f32(uint16_t h) {
    uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
    uint32_t e = (h >> 10) & 0x1Fu;
    uint32_t m = h & 0x3FFu;
    uint32_t x;
    if (e == 0) {
        if (m == 0) {
            x = sign;
        } else {
            int shift = 0;
            while (!(m & 0x400u)) { m <<= 1; shift++; }
            m &= 0x3FFu;
            x = sign | ((uint32_t)(113 - shift) << 23) | (m << 13);
        }
    } else if (e == 31) {
        x = sign | 0x7F800000u | (m << 13);
    } else {
        x = sign | ((e + 112u) << 23) | (m << 13);
    }
    float f;
    std::memcpy(&f, &x, 4);
    return f;
}

struct F4 { float x, y, z, w; };
static inline F4 f4(float x, float y, float z, float w) { return F4{x, y, z, w}; }
static inline F4 operator+(F4 a, F4 b) { return f4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
static inline F4 operator*(F4 a, float s) { return f4(a.x * s, a.y * s, a.z * s, a.w * s); }

static inline F4 load_h4(const uint16_t* p) {
    return f4(f16_to_f32(p[0]), f16_to_f32(p[1]), f16_to_f32(p[2]), f16_to_f32(p[3]));
}
static inline void store_h4(uint16_t* p, F4 v) {
    p[0] = f32_to_f16(v.x); p[1] = f32_to_f16(v.y); p[2] = f32_to_f16(v.z); p[3] = f32_to_f16(v.w);
}

// ------------------------------------------------------------------ a19: software samplers
static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

struct BilinearCoord { int i0, i1; float f; };
// Fixed-function filter addressing as D3D specifies it: the scaled coordinate u*size is snapped to x.8 fixed
// point, round to nearest (D3D12_SUBTEXEL_FRACTIONAL_BIT_COUNT = 8), THEN the half-texel offset is removed; the
// filter weight is therefore a multiple of 1/256 in [0, 255/256].  Indices are NOT yet clamped.
static inline float snap8(float x) { return floorf(x * 256.0f + 0.5f) * (1.0f / 256.0f); }
static inline BilinearCoord bilinear_coord(float u, int size) {
    float c = u * (float)size;
    if (!(c == c)) c = 0.5f;
    float lim = (float)size + 1.5f;
    if (c > lim) c = lim;
    if (c < -1.5f) c = -1.5f;
    float x = snap8(c) - 0.5f;
    float fl = floorf(x);
    BilinearCoord b;
    b.i0 = (int)fl;
    b.i1 = b.i0 + 1;
    b.f = x - fl;
    return b;
}
// a*s + b with ONE rounding (explicit fmaf — the only fused operations of the arithmetic model:
// the sampler's lerps and the blur's multiply-accumulate, i.e. HLSL `mad`)
static inline F4 fma4(F4 a, float s, F4 b) { return f4(fmaf(a.x, s, b.x), fmaf(a.y, s, b.y), fmaf(a.z, s, b.z), fmaf(a.w, s, b.w)); }
// A tap whose weight is exactly 0 does not contribute (so a sample at a texel centre IS that texel, whatever its
// neighbours hold — inf/NaN included).
static inline F4 lerp4(F4 a, F4 b, float f) { return f == 0.0f ? a : fma4(b, f, a * (1.0f - f)); }
static inline F4 bilerp(F4 c00, F4 c10, F4 c01, F4 c11, float fx, float fy) {
    return lerp4(lerp4(c00, c10, fx), lerp4(c01, c11, fx), fy);
}

// Texture2D<half4>.SampleLevel(SamplerLinearClamp, uv, 0)
static inline F4 sample_2d_h4(const uint16_t* img, int w, int h, int pitch, float u, float v) {
    BilinearCoord cx = bilinear_coord(u, w), cy = bilinear_coord(v, h);
    int x0 = clampi(cx.i0, 0, w - 1), x1 = clampi(cx.i1, 0, w - 1);
    int y0 = clampi(cy.i0, 0, h - 1), y1 = clampi(cy.i1, 0, h - 1);
    F4 c00 = load_h4(img + 4 * ((size_t)y0 * pitch + x0));
    F4 c10 = load_h4(img + 4 * ((size_t)y0 * pitch + x1));
    F4 c01 = load_h4(img + 4 * ((size_t)y1 * pitch + x0));
    F4 c11 = load_h4(img + 4 * ((size_t)y1 * pitch + x1));
    return bilerp(c00, c10, c01, c11, cx.f, cy.f);
}

// cube face <-> direction.  env_map_gen.hlsl:20-44 / MathLib.cpp:138-159 (u,v in [-1,1], unnormalized)
static inline V3 cube_dir_raw(uint32_t face, float u, float v) {
    switch (face) {
        case 0: return v3(1.0f, -v, -u);
        case 1: return v3(-1.0f, -v, u);
        case 2: return v3(u, 1.0f, v);
        case 3: return v3(u, -1.0f, -v);
        case 4: return v3(u, -v, 1.0f);
        default: return v3(-u, -v, -1.0f);
    }
}
// D3D cube addressing (inverse of the above; MathLib.cpp:73-136 uses strict '>' and falls
// through on ties, the oracle breaks ties X, then Y, then Z).  u,v returned in [0,1].
static inline void cube_face_uv(V3 d, uint32_t& face, float& u, float& v) {
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
    u = (sc / ma + 1.0f) * 0.5f;
    v = (tc / ma + 1.0f) * 0.5f;
}

// the same projection onto the face of a GIVEN major axis (0 x, 1 y, 2 z) — conditioning reports only: what the
// sample would be had a near-tie between two components been decided the other way
static inline void cube_face_uv_axis(V3 d, int axis, uint32_t& face, float& u, float& v) {
    float sc, tc, ma;
    if (axis == 0) {
        ma = fabsf(d.x);
        if (d.x >= 0.0f) { face = 0; sc = -d.z; tc = -d.y; } else { face = 1; sc = d.z; tc = -d.y; }
    } else if (axis == 1) {
        ma = fabsf(d.y);
        if (d.y >= 0.0f) { face = 2; sc = d.x; tc = d.z; } else { face = 3; sc = d.x; tc = -d.z; }
    } else {
        ma = fabsf(d.z);
        if (d.z >= 0.0f) { face = 4; sc = d.x; tc = -d.y; } else { face = 5; sc = -d.x; tc = -d.y; }
    }
    u = (sc / ma + 1.0f) * 0.5f;
    v = (tc / ma + 1.0f) * 0.5f;
}

static inline size_t cube_mip_offset(uint32_t size, uint32_t mip) {
    size_t off = 0;
    for (uint32_t m = 0; m < mip; m++) { size_t s = size >> m; off += 6 * s * s; }
    return off;
}

template <class Texel>  // Texel(face, x, y) -> F4 for in-range coordinates of one mip
static inline F4 cube_fetch_seamless(int s, uint32_t face, int x, int y, Texel texel) {
    if (x >= 0 && x < s && y >= 0 && y < s) return texel(face, x, y);
    // a tap that leaves the face in both axes is clamped in y first (oracle definition)
    if ((x < 0 || x >= s) && (y < 0 || y >= s)) y = clampi(y, 0, s - 1);
    float uu = 2.0f * ((float)x + 0.5f) / (float)s - 1.0f;
    float vv = 2.0f * ((float)y + 0.5f) / (float)s - 1.0f;
    V3 d = cube_dir_raw(face, uu, vv);
    uint32_t f2; float u2, v2;
    cube_face_uv(d, f2, u2, v2);
    int x2 = clampi((int)floorf(u2 * (float)s), 0, s - 1);
    int y2 = clampi((int)floorf(v2 * (float)s), 0, s - 1);
    return texel(f2, x2, y2);
}

template <class Texel>
static inline F4 cube_bilinear_uv(int s, uint32_t face, float u, float v, Texel texel) {
    BilinearCoord cx = bilinear_coord(u, s), cy = bilinear_coord(v, s);
    F4 c00 = cube_fetch_seamless(s, face, cx.i0, cy.i0, texel);
    F4 c10 = cube_fetch_seamless(s, face, cx.i1, cy.i0, texel);
    F4 c01 = cube_fetch_seamless(s, face, cx.i0, cy.i1, texel);
    F4 c11 = cube_fetch_seamless(s, face, cx.i1, cy.i1, texel);
    return bilerp(c00, c10, c01, c11, cx.f, cy.f);
}
template <class Texel>
static inline F4 cube_bilinear(int s, V3 dir, Texel texel) {
    uint32_t face; float u, v;
    cube_face_uv(dir, face, u, v);
    return cube_bilinear_uv(s, face, u, v, texel);
}

// the trilinear fetch at given face coordinates (conditioning reports only: the shaders address cubes by direction)
template <class MipTexel>
static inline F4 cube_trilinear_uv(uint32_t size, uint32_t mips, uint32_t face, float u, float v, float lod, MipTexel mt) {
    float maxl = (float)(mips - 1);
    if (!(lod == lod)) lod = 0.0f;
    lod = lod < 0.0f ? 0.0f : (lod > maxl ? maxl : lod);
    lod = snap8(lod);
    float fl = floorf(lod);
    uint32_t l0 = (uint32_t)fl;
    uint32_t l1 = l0 + 1 < mips ? l0 + 1 : mips - 1;
    float f = lod - fl;
    F4 a = cube_bilinear_uv((int)(size >> l0), face, u, v, mt(l0));
    if (f == 0.0f || l1 == l0) return a;
    F4 b = cube_bilinear_uv((int)(size >> l1), face, u, v, mt(l1));
    return fma4(b, f, a * (1.0f - f));
}

// TextureCube.SampleLevel(SamplerLinearClamp (MIN_MAG_MIP_LINEAR), dir, lod)
template <class MipTexel>  // MipTexel(mip) -> Texel functor
static inline F4 cube_trilinear(uint32_t size, uint32_t mips, V3 dir, float lod, MipTexel mt) {
    float maxl = (float)(mips - 1);
    if (!(lod == lod)) lod = 0.0f;
    lod = lod < 0.0f ? 0.0f : (lod > maxl ? maxl : lod);
    lod = snap8(lod);   // D3D12_MIP_LOD_FRACTIONAL_BIT_COUNT = 8
    float fl = floorf(lod);
    uint32_t l0 = (uint32_t)fl;
    uint32_t l1 = l0 + 1 < mips ? l0 + 1 : mips - 1;
    float f = lod - fl;
    F4 a = cube_bilinear((int)(size >> l0), dir, mt(l0));
    if (f == 0.0f || l1 == l0) return a;
    F4 b = cube_bilinear((int)(size >> l1), dir, mt(l1));
    return fma4(b, f, a * (1.0f - f));
}

struct CubeF32 {
    const float* data; uint32_t size, mips;
    struct T { const float* base; int s;
        F4 operator()(uint32_t f, int x, int y) const {
            const float* p = base + 4 * (((size_t)f * s + y) * s + x);
            return f4(p[0], p[1], p[2], p[3]); } };
    T operator()(uint32_t mip) const { return T{data + 4 * cube_mip_offset(size, mip), (int)(size >> mip)}; }
};
struct CubeF16 {
    const uint16_t* data; uint32_t size, mips;
    struct T { const uint16_t* base; int s;
        F4 operator()(uint32_t f, int x, int y) const {
            return load_h4(base + 4 * (((size_t)f * s + y) * s + x)); } };
    T operator()(uint32_t mip) const { return T{data + 4 * cube_mip_offset(size, mip), (int)(size >> mip)}; }
};

// ------------------------------------------------------------------ a1: brdf.hlsli:6-67
static inline float distribution_ggx(float NdotH, float roughness) {
    float a = roughness * roughness;
    float t = (NdotH * NdotH) * (a * a - 1.0f) + 1.0f;
    return a * a / fmaxf(PI_F * t * t, EPSILON_F);
}
static inline V3 fresnel(float NdotL, V3 F0) {
    float p = powf(fmaxf(1.0f - NdotL, EPSILON_F), 5.0f);
    return F0 + (v3(1.0f, 1.0f, 1.0f) - F0) * p;
}
static inline float geometry_schlick_ggx(float NdotV, float k) {
    return NdotV / fmaxf(NdotV * (1.0f - k) + k, EPSILON_F);
}
static inline float geometry_smith(float NdotL, float NdotV, float k) {
    float ggx1 = geometry_schlick_ggx(NdotV, k);
    float ggx2 = geometry_schlick_ggx(NdotL, k);
    return ggx1 * ggx2;
}
static inline V3 compute_F0(V3 albedo, float metallic) {
    return v3(lerpf(0.04f, albedo.x, metallic), lerpf(0.04f, albedo.y, metallic), lerpf(0.04f, albedo.z, metallic));
}
static inline V3 brdf(float metallic, float roughness, V3 albedo, V3 N, V3 V, V3 L) {
    V3 H = normalize3(L + V);
    float NdotL = fmaxf(dot3(N, L), 0.0f);
    float NdotV = fmaxf(dot3(N, V), 0.0f);
    float NdotH = fmaxf(dot3(N, H), 0.0f);
    V3 F0 = compute_F0(albedo, metallic);
    V3 F = fresnel(NdotL, F0);   // Q3: Schlick on NdotL
    float D = distribution_ggx(NdotH, roughness);
    float k = (roughness + 1.0f) * (roughness + 1.0f) / 8.0f;
    float G = geometry_smith(NdotL, NdotV, k);
    V3 Ks = F;
    V3 Kd = (v3(1.0f, 1.0f, 1.0f) - F) * (1.0f - metallic);
    float denom = fmaxf(4.0f * NdotL * NdotV, 0.0001f);
    // Kd * Albedo * INV_PI + Ks * D * G / max(...)
    V3 diffuse = (Kd * albedo) * INV_PI_F;
    V3 spec = ((Ks * D) * G) / denom;
    return diffuse + spec;
}

// ------------------------------------------------------------------ a2: brdf.hlsli:71-114
static inline float radical_inverse_vdc(uint32_t bits) {
    bits = (bits << 16u) | (bits >> 16u);
    bits = ((bits & 0x55555555u) << 1u) | ((bits & 0xAAAAAAAAu) >> 1u);
    bits = ((bits & 0x33333333u) << 2u) | ((bits & 0xCCCCCCCCu) >> 2u);
    bits = ((bits & 0x0F0F0F0Fu) << 4u) | ((bits & 0xF0F0F0F0u) >> 4u);
    bits = ((bits & 0x00FF00FFu) << 8u) | ((bits & 0xFF00FF00u) >> 8u);
    return (float)bits * 2.3283064365386963e-10f;
}
static inline V3 ggx_important_sample(float roughness, V3 normal, float xi_x, float xi_y) {
    float a = roughness * roughness;
    float phi = TWO_PI_F * xi_x;
    float cos_theta = sqrtf((1.0f - xi_y) / (1.0f + (a * a - 1.0f) * xi_y));
    float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
    V3 h = v3(sin_theta * cosf(phi), sin_theta * sinf(phi), cos_theta);
    V3 up = fabsf(normal.z) < 0.999f ? v3(0.0f, 0.0f, 1.0f) : v3(1.0f, 0.0f, 0.0f);
    V3 tangent = normalize3(cross3(normal, up));
    V3 bitangent = cross3(normal, tangent);
    return normalize3((tangent * h.x + bitangent * h.y) + normal * h.z);
}

// ------------------------------------------------------------------ global.hlsli:85-143
static inline float sign_custom(float x) { return x < 0.0f ? -1.0f : 1.0f; }  // Q22
static inline V3 decode_octahedron(float u, float v) {
    V3 d = v3(u * 2.0f - 1.0f, v * 2.0f - 1.0f, 0.0f);
    d.z = 1.0f - fabsf(d.x) - fabsf(d.y);
    if (d.z < 0.0f) {
        float nx = sign_custom(d.x) * (1.0f - fabsf(d.y));
        float ny = sign_custom(d.y) * (1.0f - fabsf(d.x));
        d.x = nx; d.y = ny;
    }
    return d;
}
static inline float luminance(V3 c) { return dot3(c, v3(0.2126f, 0.7152f, 0.0722f)); }

// ------------------------------------------------------------------ matrices (row-major, M*v)
static inline V3 mul_m4_dir(const float* M, V3 v) {   // mul(M, float4(v,0)).xyz
    return v3((M[0] * v.x + M[1] * v.y) + M[2] * v.z,
              (M[4] * v.x + M[5] * v.y) + M[6] * v.z,
              (M[8] * v.x + M[9] * v.y) + M[10] * v.z);
}
static inline V3 mul_m4_pos(const float* M, V3 v) {   // mul(M, float4(v,1)).xyz
    return v3(((M[0] * v.x + M[1] * v.y) + M[2] * v.z) + M[3],
              ((M[4] * v.x + M[5] * v.y) + M[6] * v.z) + M[7],
              ((M[8] * v.x + M[9] * v.y) + M[10] * v.z) + M[11]);
}

// deferred_shading.hlsl:74-77
static inline float view_space_depth(const pbr_global* g, float ndc) {
    return g->Near * g->Far / (g->Far - ndc * (g->Far - g->Near));
}
// clustered.hlsli:40-60
static inline int cluster_index3(int x, int y, int z) {
    return z + x * PBR_CLUSTER_Z + y * PBR_CLUSTER_X * PBR_CLUSTER_Z;
}
static inline int cluster_index_uv(const pbr_global* g, float u, float v, float z) {
    int sx = (int)floorf(u * (float)PBR_CLUSTER_X);
    int sy = (int)floorf((1.0f - v) * (float)PBR_CLUSTER_Y);
    float zc = fminf(fmaxf(z, g->Near), g->Far);
    int sz = (int)((float)PBR_CLUSTER_Z * logf(zc / g->Near) / logf(g->Far / g->Near));
    return cluster_index3(clampi(sx, 0, PBR_CLUSTER_X - 1), clampi(sy, 0, PBR_CLUSTER_Y - 1),
                          clampi(sz, 0, PBR_CLUSTER_Z - 1));
}
// deferred_shading.hlsl:86-89
static inline float attenuation(float d, float c0, float c1, float c2) {
    return 1.0f / fmaxf((c0 + c1 * d) + (c2 * d) * d, EPSILON_F);
}

// deferred_shading.hlsl:23-54
static inline V3 environment_diffuse(const pbr_sh_pack* sh, V3 base, float metallic, V3 n) {
    float a[4] = {n.x, n.y, n.z, 1.0f};
    float b[4] = {n.x * n.y, n.y * n.z, n.z * n.z, n.z * n.x};
    float c = n.x * n.x - n.y * n.y;
    auto dot4 = [](const float* p, const float* q) { return ((p[0] * q[0] + p[1] * q[1]) + p[2] * q[2]) + p[3] * q[3]; };
    V3 L0L1 = v3(dot4(sh->sha_r, a), dot4(sh->sha_g, a), dot4(sh->sha_b, a));
    V3 L2 = v3(dot4(sh->shb_r, b), dot4(sh->shb_g, b), dot4(sh->shb_b, b));
    L2 = L2 + v3(sh->shc[0], sh->shc[1], sh->shc[2]) * c;
    V3 irradiance = L0L1 + L2;
    V3 kd = (base * (1.0f - metallic)) * INV_PI_F;
    return kd * irradiance;
}

// hdr_tone_mapping.hlsl:27-36
static inline float aces1(float x) {
    const float a = 2.51f, b = 0.03f, c = 2.43f, d = 0.59f, e = 0.14f;
    return saturate((x * (a * x + b)) / (x * (c * x + d) + e));
}
// hdr_luminance_histogram.hlsl:23-35
static inline uint32_t luminance_bin(float lum, float min_log, float inv_range) {
    if (lum < EPSILON_F) return 0u;
    float l = saturate((log2f(lum) - min_log) * inv_range);
    return (uint32_t)floorf(l * 254.0f + 1.0f);
}

static inline uint32_t unorm8(float x) { return (uint32_t)floorf(saturate(x) * 255.0f + 0.5f); }

// blur.hlsli:17
const float GAUSS_WEIGHT[9] = {0.0148f, 0.0459f, 0.1050f, 0.1941f, 0.2803f, 0.1941f, 0.1050f, 0.0459f, 0.0148f};

}  // namespace

extern "C" {

int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
void orc_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

uint16_t orc_f32_to_f16(float f) { return f32_to_f16(f); }
float orc_f16_to_f32(uint16_t h) { return f16_to_f32(h); }

float orc_radical_inverse(uint32_t bits) { return radical_inverse_vdc(bits); }
void orc_ggx_sample(float roughness, const float n[3], float xi_x, float xi_y, float out_h[3]) {
    V3 h = ggx_important_sample(roughness, v3(n[0], n[1], n[2]), xi_x, xi_y);
    out_h[0] = h.x; out_h[1] = h.y; out_h[2] = h.z;
}
void orc_brdf(float metallic, float roughness, const float albedo[3], const float n[3],
              const float v[3], const float l[3], float out[3]) {
    V3 r = brdf(metallic, roughness, v3(albedo[0], albedo[1], albedo[2]), v3(n[0], n[1], n[2]),
                v3(v[0], v[1], v[2]), v3(l[0], l[1], l[2]));
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void orc_octa_decode(float u, float v, float out_n[3]) {
    V3 n = normalize3(decode_octahedron(u, v));
    out_n[0] = n.x; out_n[1] = n.y; out_n[2] = n.z;
}
void orc_octa_encode(const float n[3], float out_uv[2]) {  // global.hlsli:117-128
    float sum = fabsf(n[0]) + fabsf(n[1]) + fabsf(n[2]);
    V3 d = v3(n[0] / sum, n[1] / sum, n[2] / sum);
    if (d.z < 0.0f) {
        float nx = sign_custom(d.x) * (1.0f - fabsf(d.y));
        float ny = sign_custom(d.y) * (1.0f - fabsf(d.x));
        d.x = nx; d.y = ny;
    }
    out_uv[0] = d.x * 0.5f + 0.5f;
    out_uv[1] = d.y * 0.5f + 0.5f;
}
float orc_view_space_depth(const pbr_global* g, float ndc_depth) { return view_space_depth(g, ndc_depth); }
int orc_cluster_index(const pbr_global* g, float u, float v, float z_vs) { return cluster_index_uv(g, u, v, z_vs); }
float orc_attenuation(float d, float c0, float c1, float c2) { return attenuation(d, c0, c1, c2); }
void orc_aces(const float x[3], float out[3]) { out[0] = aces1(x[0]); out[1] = aces1(x[1]); out[2] = aces1(x[2]); }
uint32_t orc_luminance_bin(float lum, float min_log, float inv_range) { return luminance_bin(lum, min_log, inv_range); }
void orc_env_diffuse(const pbr_sh_pack* sh, const float albedo[3], float metallic, const float n[3], float out[3]) {
    V3 r = environment_diffuse(sh, v3(albedo[0], albedo[1], albedo[2]), metallic, v3(n[0], n[1], n[2]));
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void orc_cube_dir(uint32_t face, float u, float v, float out[3]) {
    V3 d = normalize3(cube_dir_raw(face, 2.0f * u - 1.0f, 2.0f * v - 1.0f));
    out[0] = d.x; out[1] = d.y; out[2] = d.z;
}
void orc_sample_cube_f32(const float* data, uint32_t size, uint32_t mips, const float dir[3], float lod, float out[4]) {
    F4 c = cube_trilinear(size, mips, v3(dir[0], dir[1], dir[2]), lod, CubeF32{data, size, mips});
    out[0] = c.x; out[1] = c.y; out[2] = c.z; out[3] = c.w;
}
void orc_sample_cube_f16(const uint16_t* data, uint32_t size, uint32_t mips, const float dir[3], float lod, float out[4]) {
    F4 c = cube_trilinear(size, mips, v3(dir[0], dir[1], dir[2]), lod, CubeF16{data, size, mips});
    out[0] = c.x; out[1] = c.y; out[2] = c.z; out[3] = c.w;
}
void orc_sample_2d_f16x4(const uint16_t* img, uint32_t w, uint32_t h, uint32_t pitch, float u, float v, float out[4]) {
    F4 c = sample_2d_h4(img, (int)w, (int)h, (int)pitch, u, v);
    out[0] = c.x; out[1] = c.y; out[2] = c.z; out[3] = c.w;
}

// ==================================================================== a3: precompute_brdf.hlsl:20-62
int orc_brdf_lut_rows(uint32_t res, uint32_t y0, uint32_t rows, uint16_t* out_rg) {
    if (!out_rg || res < 2 || y0 + rows > res) return PBR_ERR_INVALID;
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t yy = 0; yy < (int64_t)rows; yy++) {
        uint32_t y = y0 + (uint32_t)yy;
        for (uint32_t x = 0; x < res; x++) {
            float roughness = (float)x / (float)(res - 1);
            float NdotV = (float)(y + 1) / (float)res;
            V3 V = v3(sqrtf(1.0f - NdotV * NdotV), 0.0f, NdotV);
            V3 N = v3(0.0f, 0.0f, 1.0f);
            float A = 0.0f, B = 0.0f;
            for (uint32_t i = 0; i < PBR_SAMPLE_COUNT; i++) {
                float xi_x = (float)i / (float)PBR_SAMPLE_COUNT;
                float xi_y = radical_inverse_vdc(i);
                V3 H = ggx_important_sample(roughness, N, xi_x, xi_y);
                float VdH = dot3(V, H);
                V3 L = normalize3(H * (2.0f * VdH) - V);
                float NdotL = fmaxf(L.z, 0.0f);
                float NdotH = fmaxf(H.z, 0.0f);
                float VdotH = fmaxf(VdH, 0.0f);
                if (NdotL > 0.0f) {
                    float Fc = powf(1.0f - VdotH, 5.0f);
                    float k = roughness * roughness / 2.0f;   // Q6
                    float G = geometry_smith(NdotL, NdotV, k);
                    float G_Vis = (G * VdotH) / fmaxf(NdotH * NdotV, 0.0001f);
                    A += (1.0f - Fc) * G_Vis;
                    B += Fc * G_Vis;
                }
            }
            A = A / (float)PBR_SAMPLE_COUNT;
            B = B / (float)PBR_SAMPLE_COUNT;
            out_rg[2 * ((size_t)yy * res + x) + 0] = f32_to_f16(A);
            out_rg[2 * ((size_t)yy * res + x) + 1] = f32_to_f16(B);
        }
    }
    return PBR_OK;
}
int orc_brdf_lut(uint32_t res, uint16_t* out_rg) { return orc_brdf_lut_rows(res, 0, res, out_rg); }

// ==================================================================== source-cube mips: 2x2 box per face
int orc_cube_gen_mips(float* cube, uint32_t size, uint32_t mips) {
    if (!cube || size == 0) return PBR_ERR_INVALID;
    for (uint32_t m = 1; m < mips; m++) {
        uint32_t s = size >> m, sp = size >> (m - 1);
        if (s == 0) return PBR_ERR_INVALID;
        const float* src = cube + 4 * cube_mip_offset(size, m - 1);
        float* dst = cube + 4 * cube_mip_offset(size, m);
        for (uint32_t f = 0; f < 6; f++)
            for (uint32_t y = 0; y < s; y++)
                for (uint32_t x = 0; x < s; x++)
                    for (int c = 0; c < 4; c++) {
                        auto at = [&](uint32_t xx, uint32_t yy) { return src[4 * (((size_t)f * sp + yy) * sp + xx) + c]; };
                        float v = ((at(2 * x, 2 * y) + at(2 * x + 1, 2 * y)) + (at(2 * x, 2 * y + 1) + at(2 * x + 1, 2 * y + 1))) * 0.25f;
                        dst[4 * (((size_t)f * s + y) * s + x) + c] = v;
                    }
    }
    return PBR_OK;
}

// ==================================================================== a4: env_map_gen.hlsl:50-105
namespace {
// one output texel (index t = (face * s + y) * s + x of mip `mip`) of env_map_gen.hlsl::cs_main
static inline void prefilter_texel(const CubeF32& cube, uint32_t sky_size, uint32_t sky_mips, uint32_t size, uint32_t s,
                                   float Roughness, int64_t t, uint16_t* out4) {
    uint32_t face = (uint32_t)(t / ((int64_t)s * s));
    uint32_t y = (uint32_t)((t / s) % s), x = (uint32_t)(t % s);
    // xy = dispatch_thread_id.xy / texture_size  (texel CORNER, Q8)
    float u = (float)x / (float)s, v = (float)y / (float)s;
    V3 R = normalize3(cube_dir_raw(face, 2.0f * u - 1.0f, 2.0f * v - 1.0f));
    V3 N = R, V = R;
    V3 total = v3(0, 0, 0);
    float total_w = 0.0f;
    for (uint32_t i = 0; i < PBR_SAMPLE_COUNT; i++) {
        float xi_x = (float)i / (float)PBR_SAMPLE_COUNT;
        float xi_y = radical_inverse_vdc(i);
        V3 H = ggx_important_sample(Roughness, N, xi_x, xi_y);
        V3 L = normalize3(H * (2.0f * dot3(V, H)) - V);
        float NdotL = fmaxf(dot3(N, L), 0.0f);
        if (NdotL > 0.0f) {
            float NdotH = fmaxf(dot3(N, H), 0.0f);
            float HdotV = fmaxf(dot3(H, V), 0.0f);
            float D = distribution_ggx(NdotH, Roughness);
            float pdf = D * NdotH / (4.0f * HdotV + 0.0001f);
            // texel_sa uses the base size for every mip (Q8)
            float texel_sa = 4.0f * PI_F / ((float)(6u * size * size));
            float sample_sa = 1.0f / ((float)PBR_SAMPLE_COUNT * pdf + 0.0001f);
            float lod = Roughness == 0.0f ? 0.0f : 0.5f * log2f(sample_sa / texel_sa);
            F4 c = cube_trilinear(sky_size, sky_mips, L, lod, cube);
            total = total + v3(c.x, c.y, c.z) * NdotL;
            total_w += NdotL;
        }
    }
    total = total / total_w;
    store_h4(out4, f4(total.x, total.y, total.z, 1.0f));
}
}  // namespace

int orc_prefilter_env_mip(const float* sky, uint32_t sky_size, uint32_t sky_mips,
                          uint32_t size, uint32_t mips, uint32_t mip, uint16_t* out) {
    if (!sky || !out || mips < 1 || mip >= mips || (size >> mip) == 0) return PBR_ERR_INVALID;
    CubeF32 cube{sky, sky_size, sky_mips};
    const uint32_t s = size >> mip;
    // DeferredPipeline.cpp:99: Roughness = i / (mips - 1)
    const float Roughness = mips > 1 ? (float)mip / (float)(mips - 1) : 0.0f;
    const int64_t n = (int64_t)6 * s * s;
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t t = 0; t < n; t++) prefilter_texel(cube, sky_size, sky_mips, size, s, Roughness, t, out + 4 * t);
    return PBR_OK;
}
// the same dispatch evaluated on `count` chosen texels of mip `mip` only (texel index = (face * s + y) * s + x):
// what makes a 512^2 x 1024-sample chain checkable on the CPU in seconds
int orc_prefilter_env_texels(const float* sky, uint32_t sky_size, uint32_t sky_mips, uint32_t size, uint32_t mips,
                             uint32_t mip, const uint32_t* texels, uint32_t count, uint16_t* out) {
    if (!sky || !out || !texels || mips < 1 || mip >= mips || (size >> mip) == 0) return PBR_ERR_INVALID;
    CubeF32 cube{sky, sky_size, sky_mips};
    const uint32_t s = size >> mip;
    const float Roughness = mips > 1 ? (float)mip / (float)(mips - 1) : 0.0f;
    for (uint32_t k = 0; k < count; k++)
        if (texels[k] >= 6u * s * s) return PBR_ERR_INVALID;
#pragma omp parallel for schedule(dynamic, 8)
    for (int64_t k = 0; k < (int64_t)count; k++) prefilter_texel(cube, sky_size, sky_mips, size, s, Roughness, (int64_t)texels[k], out + 4 * k);
    return PBR_OK;
}
int orc_prefilter_env(const float* sky, uint32_t sky_size, uint32_t sky_mips,
                      uint32_t size, uint32_t mips, uint16_t* out) {
    for (uint32_t m = 0; m < mips; m++) {
        int r = orc_prefilter_env_mip(sky, sky_size, sky_mips, size, mips, m, out + 4 * cube_mip_offset(size, m));
        if (r) return r;
    }
    return PBR_OK;
}

// ==================================================================== a5: SH.cpp
namespace {
// SH.cpp:6-37
static inline void sh_basis(V3 d, float y[9]) {
    y[0] = 0.282095f;
    y[1] = 0.488603f * d.y;
    y[2] = 0.488603f * d.z;
    y[3] = 0.488603f * d.x;
    y[4] = 1.092548f * d.x * d.y;
    y[5] = 1.092548f * d.y * d.z;
    y[6] = 0.315392f * (3 * d.z * d.z - 1);
    y[7] = 1.092548f * d.x * d.z;
    y[8] = 0.546274f * (d.x * d.x - d.y * d.y);
}
const float SH_BASIS_COEF[9] = {0.282095f, 0.488603f, 0.488603f, 0.488603f, 1.092548f,
                                1.092548f, 0.315392f, 1.092548f, 0.546274f};  // SH.cpp:39-68
// SH.cpp:70-85 (SqrtPI = sqrt(PI))
static inline float cosine_sh(int l) {
    switch (l) {
        case 0: return sqrtf(PI_F) / 2.0f;
        case 1: return sqrtf(PI_F / 3.0f);
        case 2: return sqrtf(5.0f * PI_F) / 8.0f;
        default: return 0.0f;
    }
}
// SH.cpp:135-151 then PackCubeMapSHCoefficient SH.cpp:201-222
static void sh_finish(const double Lrgb[3][9], float out_pack[28]) {
    float c[3][9];
    for (int ch = 0; ch < 3; ch++) {
        for (int l = 0; l <= 2; l++)
            for (int m = -l; m <= l; m++) {
                int n = l * l + m + l;
                float K = sqrtf(4 * PI_F / (float)(2 * l + 1));
                float L = (float)Lrgb[ch][n];
                float A = cosine_sh(l);
                c[ch][n] = INV_PI_F * K * A * L;   // Q17
            }
        for (int i = 0; i < 9; i++) c[ch][i] *= SH_BASIS_COEF[i];
    }
    pbr_sh_pack p;
    float* sha[3] = {p.sha_r, p.sha_g, p.sha_b};
    float* shb[3] = {p.shb_r, p.shb_g, p.shb_b};
    for (int ch = 0; ch < 3; ch++) {
        sha[ch][0] = c[ch][3]; sha[ch][1] = c[ch][1]; sha[ch][2] = c[ch][2]; sha[ch][3] = c[ch][0];
        shb[ch][0] = c[ch][4]; shb[ch][1] = c[ch][5]; shb[ch][2] = c[ch][6] * 3; shb[ch][3] = c[ch][7];  // Q16
    }
    p.shc[0] = c[0][8]; p.shc[1] = c[1][8]; p.shc[2] = c[2][8]; p.shc[3] = 0.0f;
    std::memcpy(out_pack, &p, sizeof(p));
}
}  // namespace

// Deterministic quadrature: L_n = sum_texels colour * Y_n(dir) * dOmega(texel); this is the
// expectation of the reference's nearest-texel Monte-Carlo estimator (SH.cpp:98-133).
int orc_sh9_project(const float* sky, uint32_t size, float out_pack[28]) {
    if (!sky || !out_pack || size == 0) return PBR_ERR_INVALID;
    double L[3][9] = {};
    for (uint32_t f = 0; f < 6; f++)
        for (uint32_t y = 0; y < size; y++)
            for (uint32_t x = 0; x < size; x++) {
                float u = 2.0f * ((float)x + 0.5f) / (float)size - 1.0f;
                float v = 2.0f * ((float)y + 0.5f) / (float)size - 1.0f;
                V3 raw = cube_dir_raw(f, u, v);
                float r2 = dot3(raw, raw);
                V3 d = raw * (1.0f / sqrtf(r2));
                // solid angle of the texel: (2/size)^2 / |raw|^3
                double dw = (4.0 / ((double)size * size)) / ((double)r2 * std::sqrt((double)r2));
                float Y[9];
                sh_basis(d, Y);
                const float* p = sky + 4 * (((size_t)f * size + y) * size + x);
                for (int ch = 0; ch < 3; ch++)
                    for (int n = 0; n < 9; n++) L[ch][n] += (double)p[ch] * (double)Y[n] * dw;
            }
    sh_finish(L, out_pack);
    return PBR_OK;
}

// Seeded restatement of SHBaker::ProjectEnvironmentMap (SH.cpp:87-153) incl. the nearest-texel
// lookup CubeMapTextureData::Sample -> CalcCubeMapCoordinate -> TextureData::Sample
// (BasicStorage.cpp:126-142,191-199; MathLib.cpp:73-136; MathLib.h:1114-1118).
int orc_sh9_project_mc(const float* sky, uint32_t size, uint32_t seed, uint32_t samples, float out_pack[28]) {
    if (!sky || !out_pack || size == 0 || samples == 0) return PBR_ERR_INVALID;
    std::mt19937 gen(seed);
    std::uniform_real_distribution<float> rng(0.0f, 1.0f);
    double Lout[3][9];
    for (int ch = 0; ch < 3; ch++) {
        float c[9] = {};
        for (uint32_t i = 0; i < samples; i++) {
            float phi = 2 * PI_F * rng(gen);
            float theta = acosf(1 - 2 * rng(gen));
            float st = sinf(theta);
            V3 dir = v3(st * cosf(phi), st * sinf(phi), cosf(theta));
            V3 dn = normalize3(dir);
            // CalcCubeMapCoordinate: strict '>' comparisons, ties fall through to face 0 / tc (0,0)
            float ax = fabsf(dn.x), ay = fabsf(dn.y), az = fabsf(dn.z);
            uint32_t face = 0; float tx = 0.0f, ty = 0.0f;
            if (ax > ay && ax > az) {
                if (dn.x > 0) { tx = -dn.z / ax; ty = -dn.y / ax; face = 0; }
                else          { tx = dn.z / ax;  ty = -dn.y / ax; face = 1; }
            } else if (ay > ax && ay > az) {
                if (dn.y > 0) { tx = dn.x / ay; ty = dn.z / ay;  face = 2; }
                else          { tx = dn.x / ay; ty = -dn.z / ay; face = 3; }
            } else if (az > ax && az > ay) {
                if (dn.z > 0) { tx = dn.x / az;  ty = -dn.y / az; face = 4; }
                else          { tx = -dn.x / az; ty = -dn.y / az; face = 5; }
            }
            tx = (tx + 1) * 0.5f; ty = (ty + 1) * 0.5f;
            uint32_t row = std::min<uint32_t>((uint32_t)(tx * size), size - 1);
            uint32_t col = std::min<uint32_t>((uint32_t)(ty * size), size - 1);
            const float* p = sky + 4 * (((size_t)face * size + col) * size + row);
            float Y[9];
            sh_basis(dir, Y);
            for (int n = 0; n < 9; n++) c[n] += p[ch] * Y[n];
        }
        for (int n = 0; n < 9; n++) { c[n] *= 4 * PI_F / (float)samples; Lout[ch][n] = c[n]; }
    }
    sh_finish(Lout, out_pack);
    return PBR_OK;
}

// ==================================================================== a13: clustered_compute.hlsl:8-42
int orc_cluster_build(const pbr_global* g, pbr_cluster* clusters) {
    if (!g || !clusters) return PBR_ERR_INVALID;
    const float htan = tanf(g->Fov / 2);
    auto zplane = [&](float nx, float ny, float view_z) {
        V3 ray = v3(nx * g->Ratio * htan, ny * htan, 1.0f) * g->Near;
        float t = view_z / ray.z;
        return ray * t;
    };
    for (int ty = 0; ty < PBR_CLUSTER_Y; ty++)
        for (int tx = 0; tx < PBR_CLUSTER_X; tx++)
            for (int z = 0; z < PBR_CLUSTER_Z; z++) {
                int idx = cluster_index3(tx, ty, z);
                float znear = g->Near * powf(g->Far / g->Near, (float)z / (float)PBR_CLUSTER_Z);
                float zfar = g->Near * powf(g->Far / g->Near, (float)(z + 1) / (float)PBR_CLUSTER_Z);
                float minx = 2 * (float)tx / (float)PBR_CLUSTER_X - 1, miny = 2 * (float)ty / (float)PBR_CLUSTER_Y - 1;
                float maxx = 2 * (float)(tx + 1) / (float)PBR_CLUSTER_X - 1, maxy = 2 * (float)(ty + 1) / (float)PBR_CLUSTER_Y - 1;
                V3 min_near = zplane(minx, miny, znear), min_far = zplane(minx, miny, zfar);
                V3 max_near = zplane(maxx, maxy, znear), max_far = zplane(maxx, maxy, zfar);
                pbr_cluster& c = clusters[idx];
                c.MinBound[0] = fminf(min_near.x, min_far.x); c.MinBound[1] = fminf(min_near.y, min_far.y); c.MinBound[2] = fminf(min_near.z, min_far.z);
                c.MaxBound[0] = fmaxf(max_near.x, max_far.x); c.MaxBound[1] = fmaxf(max_near.y, max_far.y); c.MaxBound[2] = fmaxf(max_near.z, max_far.z);
                c.NumLights = 0;
            }
    return PBR_OK;
}

// clustered_culling.hlsl:11-41
int orc_cluster_cull(const pbr_global* g, const pbr_light* lights, int n, pbr_cluster* clusters) {
    if (!g || !clusters || n < 0 || n > PBR_MAX_SCENE_LIGHTS || (n > 0 && !lights)) return PBR_ERR_INVALID;
    for (int ci = 0; ci < PBR_NUM_CLUSTERS; ci++) {
        pbr_cluster& c = clusters[ci];
        for (int i = 0; i < n && c.NumLights < PBR_MAX_LIGHTS_PER_CLUSTER; i++) {
            const pbr_light& l = lights[i];
            V3 pv = mul_m4_pos(g->View, v3(l.Position[0], l.Position[1], l.Position[2]));
            float radius = l.Radius * 1.814f * sqrtf(l.Intensity);   // Q19
            V3 closest = v3(fminf(fmaxf(pv.x, c.MinBound[0]), c.MaxBound[0]),
                            fminf(fmaxf(pv.y, c.MinBound[1]), c.MaxBound[1]),
                            fminf(fmaxf(pv.z, c.MinBound[2]), c.MaxBound[2]));
            V3 d = pv - closest;
            if (dot3(d, d) < radius * radius) {
                int li = c.NumLights++;
                c.LightIndex[li] = i;
            }
        }
    }
    return PBR_OK;
}

// ==================================================================== a8-a12: deferred_shading.hlsl
int orc_deferred_shade(const pbr_global* g, const pbr_tile* tile, const pbr_gbuffer* gb,
                       const uint16_t* lut, uint32_t lut_res,
                       const uint16_t* env, uint32_t env_size, uint32_t env_mips,
                       const pbr_cluster* clusters, const pbr_light* lights,
                       uint16_t* hdr, uint32_t hdr_pitch, float* hdr_f32) {
    return orc_deferred_shade_sens(g, tile, gb, lut, lut_res, env, env_size, env_mips, clusters, lights, hdr, hdr_pitch, hdr_f32, nullptr, nullptr);
}

// The same pass, optionally reporting how ILL-CONDITIONED each pixel's colour is in fp32 (sens_rgb, 3 floats per
// pixel, may be null).  distribution_ggx computes t = NdotH^2 (a^4 - 1) + 1: near a highlight (NdotH -> 1) this cancels
// down to ~a^4 = roughness^8, so a rounding error e in NdotH changes D = a^4 / (pi t^2) by the factor
// 4 NdotH (1 - a^4) / t * e — up to 4 / roughness^8 (2.5e6 at roughness 0.19).  sens = sum over the pixel's lights of
// |specular contribution| * 4 NdotH (1 - a^4) / t / min(1, |L + V|): first-order change of the colour per unit rounding
// error in the components of H (at grazing incidence L ~ -V the sum L + V cancels before it is normalised).  Any two
// fp32 evaluations of the shader (this one, a GPU's, the reference's own on another driver) differ by a few 2^-24 in
// NdotH, i.e. by a few 2^-24 * sens in the colour; parity tests allow exactly that on top of their relative bound.
//
// flip_rgb (3 floats per pixel, may be null): the sampler model snaps texel coordinates to x.8 fixed point, so the
// filtered value is a STEP function of the coordinate with steps of 1/256 texel.  A coordinate that sits within a
// rounding error of a step edge lands on either side depending on who evaluates the reflection vector (IEEE divide vs
// reciprocal, contraction, ...).  flip = the largest change of the IBL specular term when the env sample moves by one
// step of its coarser mip in u or v, or the LUT sample by one step in N.V: what ONE such flip can cost this pixel.
int orc_deferred_shade_sens(const pbr_global* g, const pbr_tile* tile, const pbr_gbuffer* gb,
                            const uint16_t* lut, uint32_t lut_res,
                            const uint16_t* env, uint32_t env_size, uint32_t env_mips,
                            const pbr_cluster* clusters, const pbr_light* lights,
                            uint16_t* hdr, uint32_t hdr_pitch, float* hdr_f32, float* sens_rgb, float* flip_rgb) {
    if (!g || !tile || !gb || !lut || !env || !clusters || !hdr) return PBR_ERR_INVALID;
    CubeF16 cube{env, env_size, env_mips};
    // vs_main, deferred_shading.hlsl:91-121
    const float near_height = 2 * g->Near * tanf(g->Fov / 2);
    const float near_width = near_height * g->Ratio;
    const V3 cam = v3(g->CameraPos[0], g->CameraPos[1], g->CameraPos[2]);
#pragma omp parallel for schedule(dynamic, 4)
    for (int64_t py = 0; py < (int64_t)tile->h; py++) {
        for (uint32_t px = 0; px < tile->w; px++) {
            size_t gi = (size_t)py * gb->pitch + px;
            if (gb->stencil[gi] == 0) continue;   // DeferredPipeline.h:176-181: stencil ref 0 < value
            float u = ((float)(tile->x0 + px) + 0.5f) / (float)tile->full_w;
            float v = ((float)(tile->y0 + (uint32_t)py) + 0.5f) / (float)tile->full_h;
            float ndc_x = 2.0f * u - 1.0f, ndc_y = 1.0f - 2.0f * v;
            // interpolated camera_vec: (ndc/2) * (near_width, near_height), z = Near (D3D12Device.cpp:167-176)
            V3 cv_view = v3(ndc_x * 0.5f * near_width, ndc_y * 0.5f * near_height, g->Near);
            V3 camera_vec = mul_m4_dir(g->InvView, cv_view);

            uint32_t a = gb->A[gi], b = gb->B[gi], c = gb->C[gi];
            V3 albedo = v3((float)(a & 255u) / 255.0f, (float)((a >> 8) & 255u) / 255.0f, (float)((a >> 16) & 255u) / 255.0f);
            float emission = (float)(a >> 24) / 255.0f;
            float roughness = (float)(c & 255u) / 255.0f;
            float metallic = (float)((c >> 8) & 255u) / 255.0f;
            V3 n = normalize3(decode_octahedron((float)(b & 255u) / 255.0f, (float)((b >> 8) & 255u) / 255.0f));

            float depth_ndc = gb->depth[gi];
            float z_vs = view_space_depth(g, depth_ndc);
            V3 pos = cam + camera_vec * z_vs / g->Near;     // ReconstructWorldPosition :79-83
            V3 view = normalize3(cam - pos);

            V3 env_diffuse = environment_diffuse(&g->SkyBoxSH, albedo, metallic, n);

            // EnvironmentSpecular :56-70
            V3 F0 = compute_F0(albedo, metallic);
            float NdV = dot3(n, view);
            float NdotV = fmaxf(NdV, 0.0f);
            V3 R = normalize3(n * (2.0f * NdV) - view);
            F4 envc = cube_trilinear(env_size, env_mips, R, roughness * (float)PBR_ENV_MIPS, cube);   // Q4
            // LUT: Texture2D<half2>.Sample(LinearClamp, (roughness, NdotV)), Q5
            BilinearCoord cx = bilinear_coord(roughness, (int)lut_res), cy = bilinear_coord(NdotV, (int)lut_res);
            int x0 = clampi(cx.i0, 0, (int)lut_res - 1), x1 = clampi(cx.i1, 0, (int)lut_res - 1);
            int y0 = clampi(cy.i0, 0, (int)lut_res - 1), y1 = clampi(cy.i1, 0, (int)lut_res - 1);
            auto lutat = [&](int xx, int yy) {
                const uint16_t* p = lut + 2 * ((size_t)yy * lut_res + xx);
                return f4(f16_to_f32(p[0]), f16_to_f32(p[1]), 0.0f, 0.0f); };
            F4 lb = bilerp(lutat(x0, y0), lutat(x1, y0), lutat(x0, y1), lutat(x1, y1), cx.f, cy.f);
            V3 env_specular = v3(envc.x, envc.y, envc.z) * (F0 * lb.x + v3(lb.y, lb.y, lb.y));
            V3 flip = v3(0, 0, 0);
            if (flip_rgb) {   // conditioning report only
                auto amax = [](V3 a, V3 b) { return v3(fmaxf(a.x, fabsf(b.x)), fmaxf(a.y, fabsf(b.y)), fmaxf(a.z, fabsf(b.z))); };
                uint32_t face; float cu, cv;
                cube_face_uv(R, face, cu, cv);
                float lod = roughness * (float)PBR_ENV_MIPS;
                float lc = lod < 0.0f ? 0.0f : (lod > (float)(env_mips - 1) ? (float)(env_mips - 1) : lod);
                uint32_t l1 = (uint32_t)floorf(snap8(lc)) + 1;
                if (l1 > env_mips - 1) l1 = env_mips - 1;
                float step = 1.0f / (256.0f * (float)(env_size >> l1));
                V3 spec_w = F0 * lb.x + v3(lb.y, lb.y, lb.y);
                V3 e0 = v3(envc.x, envc.y, envc.z);
                const float du[4] = {step, -step, 0.0f, 0.0f}, dv[4] = {0.0f, 0.0f, step, -step};
                for (int k = 0; k < 4; k++) {
                    F4 e = cube_trilinear_uv(env_size, env_mips, face, cu + du[k], cv + dv[k], lod, cube);
                    flip = amax(flip, (v3(e.x, e.y, e.z) - e0) * spec_w);
                }
                // a near-tie between the two largest components of R picks the face; the model's seamless rule is only
                // approximately continuous across a face edge (coarse mips!), so the other face may give another value
                {
                    float ab[3] = {fabsf(R.x), fabsf(R.y), fabsf(R.z)};
                    int major = face >> 1;
                    for (int ax = 0; ax < 3; ax++) {
                        if (ax == major || ab[ax] < ab[major] * (1.0f - 1e-5f)) continue;
                        uint32_t f2; float u2, v2;
                        cube_face_uv_axis(R, ax, f2, u2, v2);
                        F4 e = cube_trilinear_uv(env_size, env_mips, f2, u2, v2, lod, cube);
                        flip = amax(flip, (v3(e.x, e.y, e.z) - e0) * spec_w);
                    }
                }
                float ls = 1.0f / (256.0f * (float)lut_res);
                for (int k = -1; k <= 1; k += 2) {
                    BilinearCoord cy2 = bilinear_coord(NdotV + (float)k * ls, (int)lut_res);
                    int yy0 = clampi(cy2.i0, 0, (int)lut_res - 1), yy1 = clampi(cy2.i1, 0, (int)lut_res - 1);
                    F4 l2 = bilerp(lutat(x0, yy0), lutat(x1, yy0), lutat(x0, yy1), lutat(x1, yy1), cx.f, cy2.f);
                    flip = amax(flip, e0 * (F0 * (l2.x - lb.x) + v3(l2.y - lb.y, l2.y - lb.y, l2.y - lb.y)));
                }
            }

            // point lights :159-186
            int ci = cluster_index_uv(g, u, v, z_vs);
            const pbr_cluster& cl = clusters[ci];
            V3 pl = v3(0, 0, 0);
            V3 sens = v3(0, 0, 0);
            for (int i = 0; i < cl.NumLights; i++) {
                const pbr_light& lt = lights[cl.LightIndex[i]];
                V3 dir = v3(lt.Position[0], lt.Position[1], lt.Position[2]) - pos;
                float dist = sqrtf(dot3(dir, dir));
                dir = dir / dist;
                float NdotL = fmaxf(dot3(n, dir), 0.0f);
                float att = attenuation(dist, lt.C0, lt.C1, lt.C2);
                V3 f = brdf(metallic, roughness, albedo, n, view, dir);
                V3 col = v3(lt.Color[0], lt.Color[1], lt.Color[2]);
                pl = pl + (((f * col) * lt.Intensity) * att) * NdotL;
                if (sens_rgb) {   // conditioning report only: does not feed the colour
                    V3 H = normalize3(dir + view);
                    float NdotH = fmaxf(dot3(n, H), 0.0f);
                    float a = roughness * roughness, a4 = a * a;
                    float t = (NdotH * NdotH) * (a4 - 1.0f) + 1.0f;
                    if (PI_F * t * t > EPSILON_F) {
                        V3 diffuse = ((v3(1.0f, 1.0f, 1.0f) - fresnel(NdotL, F0)) * (1.0f - metallic) * albedo) * INV_PI_F;
                        V3 spec = f - diffuse;
                        float amp = 4.0f * NdotH * (1.0f - a4) / fabsf(t);
                        // H = normalize(L + V): at grazing incidence |L + V| << 1 and the components of L + V (absolute
                        // rounding error ~2^-24 each) are relatively that much less accurate
                        V3 wv = dir + view;
                        float wl = sqrtf(dot3(wv, wv));
                        if (wl < 1.0f) amp /= fmaxf(wl, 1e-6f);
                        V3 c = (((spec * col) * lt.Intensity) * att) * NdotL;
                        sens = sens + v3(fabsf(c.x), fabsf(c.y), fabsf(c.z)) * amp;
                    }
                }
            }
            V3 emission_l = albedo * emission;
            V3 out = ((env_diffuse + env_specular) + pl) + emission_l;   // Q1: directional light dropped
            size_t oi = (size_t)py * hdr_pitch + px;
            store_h4(hdr + 4 * oi, f4(out.x, out.y, out.z, 1.0f));
            if (hdr_f32) { hdr_f32[4 * oi] = out.x; hdr_f32[4 * oi + 1] = out.y; hdr_f32[4 * oi + 2] = out.z; hdr_f32[4 * oi + 3] = 1.0f; }
            if (sens_rgb) { sens_rgb[3 * oi] = sens.x; sens_rgb[3 * oi + 1] = sens.y; sens_rgb[3 * oi + 2] = sens.z; }
            if (flip_rgb) { flip_rgb[3 * oi] = flip.x; flip_rgb[3 * oi + 1] = flip.y; flip_rgb[3 * oi + 2] = flip.z; }
        }
    }
    return PBR_OK;
}

// ==================================================================== 8f-1: skybox.hlsl:12-28
namespace {
// face coordinates (sc/ma, tc/ma in [-1,1]) of `d` on a GIVEN face (no major-axis test)
static inline void cube_project_on_face(V3 d, uint32_t face, float& u, float& v) {
    float sc, tc, ma;
    switch (face) {
        case 0: ma = d.x;  sc = -d.z; tc = -d.y; break;
        case 1: ma = -d.x; sc = d.z;  tc = -d.y; break;
        case 2: ma = d.y;  sc = d.x;  tc = d.z;  break;
        case 3: ma = -d.y; sc = d.x;  tc = -d.z; break;
        case 4: ma = d.z;  sc = d.x;  tc = -d.y; break;
        default: ma = -d.z; sc = -d.x; tc = -d.y; break;
    }
    u = sc / ma;
    v = tc / ma;
}
}  // namespace

int orc_skybox(const pbr_global* g, const pbr_tile* tile, const float* sky, uint32_t sky_size, uint32_t sky_mips,
               const uint8_t* stencil, uint32_t pitch, uint16_t* hdr, uint32_t hdr_pitch) {
    if (!g || !tile || !sky || !stencil || !hdr) return PBR_ERR_INVALID;
    CubeF32 cube{sky, sky_size, sky_mips};
    const float near_height = 2 * g->Near * tanf(g->Fov / 2);
    const float near_width = near_height * g->Ratio;
    auto ray = [&](float gx, float gy) {   // world-space camera ray through the centre of global pixel (gx, gy)
        float u = (gx + 0.5f) / (float)tile->full_w, v = (gy + 0.5f) / (float)tile->full_h;
        float ndc_x = 2.0f * u - 1.0f, ndc_y = 1.0f - 2.0f * v;
        return mul_m4_dir(g->InvView, v3(ndc_x * 0.5f * near_width, ndc_y * 0.5f * near_height, g->Near));
    };
#pragma omp parallel for schedule(static)
    for (int64_t py = 0; py < (int64_t)tile->h; py++)
        for (uint32_t px = 0; px < tile->w; px++) {
            if (stencil[(size_t)py * pitch + px] != 0) continue;
            const float gx = (float)(tile->x0 + px), gy = (float)(tile->y0 + (uint32_t)py);
            V3 d = ray(gx, gy);
            uint32_t face; float fu, fv;
            cube_face_uv(d, face, fu, fv);
            float u0, v0, ux, vx, uy, vy;
            cube_project_on_face(d, face, u0, v0);
            cube_project_on_face(ray(gx + 1.0f, gy), face, ux, vx);
            cube_project_on_face(ray(gx, gy + 1.0f), face, uy, vy);
            const float half_size = 0.5f * (float)sky_size;
            float rx = half_size * sqrtf((ux - u0) * (ux - u0) + (vx - v0) * (vx - v0));
            float ry = half_size * sqrtf((uy - u0) * (uy - u0) + (vy - v0) * (vy - v0));
            float lod = log2f(fmaxf(rx, ry));
            F4 c = cube_trilinear(sky_size, sky_mips, d, lod, cube);
            store_h4(hdr + 4 * ((size_t)py * hdr_pitch + px), f4(c.x, c.y, c.z, 1.0f));
        }
    return PBR_OK;
}

// ==================================================================== 8f-2: gbuffer.hlsl::ps_main :88-149
int orc_gbuffer_encode(const float* m0, const float* m1, const float* m2, uint32_t w, uint32_t h, uint32_t pitch,
                       uint32_t* A, uint32_t* B, uint32_t* C) {
    if (!m0 || !m1 || !m2 || !A || !B || !C) return PBR_ERR_INVALID;
    for (uint32_t y = 0; y < h; y++)
        for (uint32_t x = 0; x < w; x++) {
            const size_t i = (size_t)y * pitch + x;
            const float* a = m0 + 4 * i; const float* b = m1 + 4 * i; const float* c = m2 + 4 * i;
            // decode_gamma (global.hlsli:73-77): pow(c, 2.2) per channel
            uint32_t pa = unorm8(powf(a[0], 2.2f)) | (unorm8(powf(a[1], 2.2f)) << 8) | (unorm8(powf(a[2], 2.2f)) << 16) | (unorm8(a[3]) << 24);
            V3 n = normalize3(v3(b[0], b[1], b[2]));
            float uv[2];
            orc_octa_encode(&n.x, uv);
            uint32_t pb = unorm8(uv[0]) | (unorm8(uv[1]) << 8) | (255u << 16);   // (pack_normal, 1, 0)
            uint32_t pc = unorm8(b[3]) | (unorm8(c[0]) << 8) | (unorm8(c[1]) << 16);
            A[i] = pa; B[i] = pb; C[i] = pc;
        }
    return PBR_OK;
}

// ==================================================================== 8f-3: Radiance RGBE texel decode
// ResourceLoader.cpp:381-406 calls DirectX::LoadFromHDRFile (DirectXTex: vcpkg dependency, absent, unpinned);
// restated from the published format (Ward, "Real Pixels"): e == 0 -> 0, else mantissa * 2^(e - (128 + 8)).
int orc_rgbe_decode(const uint8_t* rgbe, size_t texels, float* out) {
    if (!rgbe || !out) return PBR_ERR_INVALID;
    for (size_t i = 0; i < texels; i++) {
        const uint8_t* p = rgbe + 4 * i;
        float f = p[3] ? ldexpf(1.0f, (int)p[3] - 136) : 0.0f;
        out[4 * i + 0] = (float)p[0] * f;
        out[4 * i + 1] = (float)p[1] * f;
        out[4 * i + 2] = (float)p[2] * f;
        out[4 * i + 3] = 1.0f;
    }
    return PBR_OK;
}

// ==================================================================== a14: bloom_prefilter.hlsl:17-60
int orc_bloom_prefilter(const uint16_t* hdr, uint32_t w, uint32_t h, uint32_t pitch,
                        uint16_t* out, float threshold, float knee) {
    if (!hdr || !out || (w >> 1) == 0 || (h >> 1) == 0) return PBR_ERR_INVALID;
    const uint32_t ow = w >> 1, oh = h >> 1;
    const float tx = 1.0f / (float)ow, ty = 1.0f / (float)oh;   // DeferredPipeline.cpp:418
    static const float offs[5][2] = {{0, 0}, {-1, -1}, {-1, 1}, {1, -1}, {1, 1}};
#pragma omp parallel for schedule(static)
    for (int64_t y = 0; y < (int64_t)oh; y++)
        for (uint32_t x = 0; x < ow; x++) {
            float u = (float)x * tx, v = (float)y * ty;   // no +0.5 (Q9)
            V3 total = v3(0, 0, 0);
            float total_w = 0.0f;
            for (int i = 0; i < 5; i++) {
                F4 c = sample_2d_h4(hdr, (int)w, (int)h, (int)pitch, u + offs[i][0] * tx, v + offs[i][1] * ty);
                float brightness = fmaxf(c.x, fmaxf(c.y, c.z));
                float soft = fminf(fmaxf(brightness - threshold + threshold * knee, 0.0f), 2 * threshold * knee);
                soft /= 4 * threshold * knee + 0.00001f;
                float contribution = fmaxf(soft, brightness - threshold) / fmaxf(brightness, 0.00001f);
                V3 col = v3(c.x, c.y, c.z) * contribution;
                float wgt = 1.0f / (luminance(col) + 1.0f);
                total = total + col * wgt;
                total_w += wgt;
            }
            if (total_w > 0.0f) total = total / total_w;
            store_h4(out + 4 * ((size_t)y * ow + x), f4(total.x, total.y, total.z, 1.0f));
        }
    return PBR_OK;
}

// ==================================================================== a15: blur.hlsli:24-89
namespace {
// One 256-thread group of blur_horizontal (blur.hlsli:24-55): fills Cache[264] for the
// group starting at output pixel gx0 of row y and returns the 9-tap sums for thread t.
static void blur_h_group(const uint16_t* in, int iw, int ih, float tx, float ty, uint32_t gx0, uint32_t y, F4 cache[264]) {
    for (uint32_t t = 0; t < 256; t++) {
        float uvx = ((float)(gx0 + t) + 0.5f) * tx;
        float uvy = ((float)y + 0.5f) * ty;
        if (t < 4) {
            float x = fmaxf(uvx - 4.0f * tx, 0.0f);
            cache[t] = sample_2d_h4(in, iw, ih, iw, x, uvy);
        }
        if (t >= 252) {
            float x = fminf(uvx + 4.0f * tx, 1.0f);
            cache[t + 8] = sample_2d_h4(in, iw, ih, iw, x, uvy);
        }
        cache[t + 4] = sample_2d_h4(in, iw, ih, iw, uvx, uvy);
    }
}
static void blur_v_group(const uint16_t* in, int iw, int ih, float tx, float ty, uint32_t x, uint32_t gy0, F4 cache[264]) {
    for (uint32_t t = 0; t < 256; t++) {
        float uvx = ((float)x + 0.5f) * tx;
        float uvy = ((float)(gy0 + t) + 0.5f) * ty;
        if (t < 4) {
            float yy = fmaxf(uvy - 4.0f * ty, 0.0f);
            cache[t] = sample_2d_h4(in, iw, ih, iw, uvx, yy);
        }
        if (t >= 252) {
            float yy = fminf(uvy + 4.0f * ty, 1.0f);
            cache[t + 8] = sample_2d_h4(in, iw, ih, iw, uvx, yy);
        }
        cache[t + 4] = sample_2d_h4(in, iw, ih, iw, uvx, uvy);
    }
}
static inline F4 gauss9(const F4* cache_at_t) {   // cache_at_t = &Cache[gtid] (tap -4)
    F4 v = f4(0, 0, 0, 0);
    for (int i = 0; i < 9; i++) v = fma4(cache_at_t[i], GAUSS_WEIGHT[i], v);   // value += pixel * weight as a fused mad
    return v;
}
}  // namespace

int orc_blur_h(const uint16_t* in, uint32_t iw, uint32_t ih, uint16_t* out, uint32_t ow, uint32_t oh) {
    if (!in || !out || !iw || !ih || !ow || !oh) return PBR_ERR_INVALID;
    const float tx = 1.0f / (float)ow, ty = 1.0f / (float)oh;
#pragma omp parallel for schedule(static)
    for (int64_t y = 0; y < (int64_t)oh; y++) {
        F4 cache[264];
        for (uint32_t gx0 = 0; gx0 < ow; gx0 += 256) {
            blur_h_group(in, (int)iw, (int)ih, tx, ty, gx0, (uint32_t)y, cache);
            for (uint32_t t = 0; t < 256 && gx0 + t < ow; t++)
                store_h4(out + 4 * ((size_t)y * ow + gx0 + t), gauss9(cache + t));
        }
    }
    return PBR_OK;
}
int orc_blur_v(const uint16_t* in, uint32_t iw, uint32_t ih, uint16_t* out, uint32_t ow, uint32_t oh) {
    if (!in || !out || !iw || !ih || !ow || !oh) return PBR_ERR_INVALID;
    const float tx = 1.0f / (float)ow, ty = 1.0f / (float)oh;
#pragma omp parallel for schedule(static)
    for (int64_t x = 0; x < (int64_t)ow; x++) {
        F4 cache[264];
        for (uint32_t gy0 = 0; gy0 < oh; gy0 += 256) {
            blur_v_group(in, (int)iw, (int)ih, tx, ty, (uint32_t)x, gy0, cache);
            for (uint32_t t = 0; t < 256 && gy0 + t < oh; t++)
                store_h4(out + 4 * ((size_t)(gy0 + t) * ow + x), gauss9(cache + t));
        }
    }
    return PBR_OK;
}
// bloom_upsample_add.hlsl:13-25: lower first, then upper, then add
int orc_bloom_upsample_add(const uint16_t* upper, uint32_t uw, uint32_t uh,
                           const uint16_t* lower, uint32_t lw, uint32_t lh, uint16_t* out) {
    if (!upper || !lower || !out || !uw || !uh || !lw || !lh) return PBR_ERR_INVALID;
    const float tx = 1.0f / (float)uw, ty = 1.0f / (float)uh;
#pragma omp parallel for schedule(static)
    for (int64_t y = 0; y < (int64_t)uh; y++) {
        F4 cl[264], cu[264];
        for (uint32_t gx0 = 0; gx0 < uw; gx0 += 256) {
            blur_h_group(lower, (int)lw, (int)lh, tx, ty, gx0, (uint32_t)y, cl);
            blur_h_group(upper, (int)uw, (int)uh, tx, ty, gx0, (uint32_t)y, cu);
            for (uint32_t t = 0; t < 256 && gx0 + t < uw; t++)
                store_h4(out + 4 * ((size_t)y * uw + gx0 + t), gauss9(cl + t) + gauss9(cu + t));
        }
    }
    return PBR_OK;
}
// bloom_merge.hlsl:7-11
int orc_bloom_merge(uint16_t* hdr, uint32_t pitch, const uint16_t* in, uint32_t w, uint32_t h) {
    if (!hdr || !in) return PBR_ERR_INVALID;
#pragma omp parallel for schedule(static)
    for (int64_t y = 0; y < (int64_t)h; y++)
        for (uint32_t x = 0; x < w; x++) {
            uint16_t* p = hdr + 4 * ((size_t)y * pitch + x);
            store_h4(p, load_h4(p) + load_h4(in + 4 * ((size_t)y * w + x)));
        }
    return PBR_OK;
}

static size_t bloom_level_offset(uint32_t w, uint32_t h, uint32_t level) {
    size_t off = 0;
    for (uint32_t l = 0; l < level; l++) off += (size_t)(w >> l) * (h >> l);
    return off;
}
// BloomPass::Execute, DeferredPipeline.cpp:400-570 (schedule comment :379-399)
int orc_bloom(uint16_t* hdr, uint32_t w, uint32_t h, uint32_t pitch,
              uint16_t* A, uint16_t* B, float threshold, float knee) {
    if (!hdr || !A || !B || (w >> (PBR_BLOOM_MIPS - 1)) == 0 || (h >> (PBR_BLOOM_MIPS - 1)) == 0) return PBR_ERR_INVALID;
    auto a = [&](uint32_t l) { return A + 4 * bloom_level_offset(w, h, l); };
    auto b = [&](uint32_t l) { return B + 4 * bloom_level_offset(w, h, l); };
    auto W = [&](uint32_t l) { return w >> l; };
    auto H = [&](uint32_t l) { return h >> l; };
    int r = orc_bloom_prefilter(hdr, w, h, pitch, a(1), threshold, knee);
    if (r) return r;
    for (uint32_t i = 0; i < PBR_BLOOM_STEP; i++) {           // downsample
        uint32_t up = i + 1, lo = i + 2;
        if ((r = orc_blur_h(a(up), W(up), H(up), b(lo), W(lo), H(lo)))) return r;
        if ((r = orc_blur_v(b(lo), W(lo), H(lo), a(lo), W(lo), H(lo)))) return r;
    }
    for (int i = PBR_BLOOM_STEP - 1; i >= 0; i--) {           // upsample
        uint32_t up = (uint32_t)i + 1;
        if ((r = orc_bloom_upsample_add(a(up), W(up), H(up), a(up + 1), W(up + 1), H(up + 1), b(up)))) return r;
        if ((r = orc_blur_v(b(up), W(up), H(up), a(up), W(up), H(up)))) return r;
    }
    if ((r = orc_blur_h(a(1), W(1), H(1), b(0), w, h))) return r;   // merge
    if ((r = orc_blur_v(b(0), w, h, a(0), w, h))) return r;
    return orc_bloom_merge(hdr, pitch, a(0), w, h);
}

// ==================================================================== a16
int orc_lum_histogram(const uint16_t* hdr, uint32_t w, uint32_t h, uint32_t pitch,
                      float min_log, float inv_range, uint32_t* hist) {
    if (!hdr || !hist) return PBR_ERR_INVALID;
    // integer counts: a per-thread histogram folded at the end gives the same result in any order
#pragma omp parallel
    {
        uint32_t local[PBR_HISTOGRAM_BINS] = {0};
#pragma omp for schedule(static) nowait
        for (int64_t y = 0; y < (int64_t)h; y++)
            for (uint32_t x = 0; x < w; x++) {
                F4 c = load_h4(hdr + 4 * ((size_t)y * pitch + x));
                local[luminance_bin(luminance(v3(c.x, c.y, c.z)), min_log, inv_range)]++;
            }
#pragma omp critical
        for (int i = 0; i < PBR_HISTOGRAM_BINS; i++) hist[i] += local[i];
    }
    return PBR_OK;
}

// ==================================================================== a17: hdr_average_histogram.hlsl:26-73
namespace {
static float average_bin(const uint32_t* hist, uint32_t pixel_count) {
    float s[256];
    for (uint32_t i = 0; i < 256; i++) s[i] = (float)(uint32_t)(hist[i] * i);   // uint32 product (Q15)
    for (uint32_t step = 128; step > 0; step >>= 1)
        for (uint32_t i = 0; i < step; i++) s[i] += s[i + step];
    return s[0] / (float)(pixel_count - hist[0]);   // Q14: 0/0 on an all-black frame
}
}  // namespace
float orc_lum_average_bin(const uint32_t* hist, uint32_t pixel_count) { return average_bin(hist, pixel_count); }
int orc_lum_average(uint32_t* hist, uint32_t pixel_count, float min_log, float range, float dt, float* avg) {
    if (!hist || !avg) return PBR_ERR_INVALID;
    float ab = average_bin(hist, pixel_count);
    // BinIndexToLuminance(uint bin_index): float -> uint truncation (Q13); NaN / negative -> 0
    uint32_t bin = (ab == ab && ab > 0.0f) ? (ab >= 4294967296.0f ? 0xFFFFFFFFu : (uint32_t)ab) : 0u;
    float log_l = ((float)bin - 1.0f) / 254.0f;
    float lum = exp2f(log_l * range + min_log);
    float prev = *avg;
    *avg = lerpf(prev, lum, saturate(1.0f - expf(-dt * 1.6f)));
    for (int i = 0; i < 256; i++) hist[i] = 0;
    return PBR_OK;
}

// ==================================================================== a18: hdr_tone_mapping.hlsl:9-52
int orc_tonemap(const uint16_t* hdr, uint32_t w, uint32_t h, uint32_t pitch, const float* avg,
                uint32_t* rgba8, uint32_t out_pitch) {
    if (!hdr || !avg || !rgba8) return PBR_ERR_INVALID;
    const float l_max = 9.6f * (*avg);
#pragma omp parallel for schedule(static)
    for (int64_t y = 0; y < (int64_t)h; y++)
        for (uint32_t x = 0; x < w; x++) {
            F4 c = load_h4(hdr + 4 * ((size_t)y * pitch + x));
            float e[3] = {c.x / (l_max + 0.001f), c.y / (l_max + 0.001f), c.z / (l_max + 0.001f)};
            uint32_t px = 0xFF000000u;
            for (int k = 0; k < 3; k++) {
                float m = aces1(e[k]);
                float gcorr = powf(m, 0.454545f);   // encode_gamma, global.hlsli:79-83
                px |= unorm8(gcorr) << (8 * k);
            }
            rgba8[(size_t)y * out_pitch + x] = px;
        }
    return PBR_OK;
}

}  // extern "C"
