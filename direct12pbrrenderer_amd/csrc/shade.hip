// shade.hip — deferred Cook-Torrance shade of a G-buffer tile (deferred_shading.hlsl:91-192).
//
// Replaces the stencil-masked full-screen draw of DeferredShadingPass::Execute
// (DeferredPipeline.cpp:187-206).  MI355X mapping:
//  * one lane per pixel, a wave = 64 consecutive pixels of a row: every G-buffer plane is read as
//    one 256-byte (RGBA8 / depth) or 64-byte (stencil) coalesced segment per wave, the half4 output
//    is one 512-byte store; a block walks SHADE_ROWS rows after staging its tables once;
//  * the light table is staged in LDS as nine structure-of-arrays planes (position, colour * intensity, attenuation):
//    the same component of two lights lands in an adjacent VGPR pair straight from two ds_read_b32;
//  * the light lists of the clusters the block can touch are staged in LDS as LDS byte addresses of the lights (one
//    dword each: a trip's two entries are ONE ds_read_b64, each address in its own register), so the divergent per-lane
//    list walk is an LDS read, not a global gather (blocks that span too many cluster tiles — tiny render targets —
//    fall back to the global list);
//  * the env chain is sampled from its padded layout (pbr_env_pad): no seam branches, each bilinear row is one
//    16-byte load;
//  * with 256 lights the kernel is FP32-VALU-issue-bound (46 packed + 4 transcendental + 2 plain instructions per pair
//    of lights, ~500 per pixel around the loop), not HBM-bound.
#include <type_traits>
#include "pbr_internal.hpp"
#include "pbr_device.hpp"

using namespace pbr;

struct ShadeParams {
    pbr_sh_pack sh;
    float InvView[9];      // 3x3 part, row-major
    float CameraPos[3];
    float Near, Far;
    float near_width, near_height;   // 2 Near tan(Fov/2) [* Ratio]  (vs_main :94-95), host libm
    float log_far_near;              // log(Far/Near) of ClusterIndex (clustered.hlsli:53), host libm
    float inv_near, slice_k;         // 1 / Near and PBR_CLUSTER_Z / log2(Far/Near) (host, from double): the slice index's quick estimate
    uint32_t x0, y0, w, h, full_w, full_h;
    const uint32_t* A;
    const uint32_t* B;
    const uint32_t* C;
    const float* depth;
    const uint8_t* stencil;
    uint32_t pitch;
    const pbr_half* lut;
    uint32_t lut_res;
    const pbr_half* env;   // padded layout
    uint32_t env_size, env_mips;
    uint32_t env_mip_off[16];   // texel offset of each padded mip (host-computed)
    const pbr_cluster* clusters;
    const pbr_light* lights;
    pbr_half* hdr;
    uint32_t hdr_pitch;
    float* hdr_f32;        // F32OUT instantiations only (pbr_deferred_shade_f32): the colour BEFORE the fp16 store
};

// ViewSpaceDepth (deferred_shading.hlsl:74-77): Near Far / (Far - d (Far - Near)).  The denominator cancels (d -> 1:
// Far - d (Far - Near) is ~Far / z_vs), so ONE rounding more or less in it moves z_vs by ~z_vs / Near ulps — enough to
// push a pixel across a cluster-slice boundary and hand it another light list.  Evaluated operation by operation (no
// fused multiply-add), with IEEE division, exactly as the oracle: the slice index is then a function of the depth texel.
__device__ __forceinline__ float view_space_depth(float d, float near_z, float far_z) {
#pragma clang fp contract(off)
    const float range = far_z - near_z;
    const float prod = d * range;
    const float den = far_z - prod;
    const float num = near_z * far_z;
    return num / den;
}

// the same expression with v_rcp for the division (<= 2 ulp from the IEEE result): for consumers that are continuous in z_vs
__device__ __forceinline__ float view_space_depth_quick(float d, float near_z, float far_z) {
#pragma clang fp contract(off)
    const float range = far_z - near_z;
    const float prod = d * range;
    const float den = far_z - prod;
    return (near_z * far_z) * rcp(den);
}

__device__ __forceinline__ float sign_custom(float x) { return x < 0.0f ? -1.0f : 1.0f; }   // global.hlsli:85-88 (Q22)

// global.hlsli:101-115
__device__ __forceinline__ V3 decode_octahedron(float u, float v) {
    V3 d = v3(u * 2.0f - 1.0f, v * 2.0f - 1.0f, 0.0f);
    d.z = 1.0f - fabsf(d.x) - fabsf(d.y);
    if (d.z < 0.0f) {
        float nx = sign_custom(d.x) * (1.0f - fabsf(d.y));
        float ny = sign_custom(d.y) * (1.0f - fabsf(d.x));
        d.x = nx; d.y = ny;
    }
    return d;
}

// ------------------------------------------------------------------------------------------------
// "Padded" env chain = the footprint layout of pbr_device.hpp::env_padded_mip_offset: entry (face, yq, xq), xq, yq in
// [0, s], holds the four texels of the bilinear footprint whose origin is tap (xq-1, yq-1), each resolved by the
// seamless-cube rule.  One thread per stored texel.
__global__ __launch_bounds__(256) void k_env_pad(const pbr_half* __restrict__ src, pbr_half* __restrict__ dst, int s) {
    const int sq = s + 1;
    const size_t n = (size_t)6 * sq * sq * 4;
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    const int k = (int)(t & 3), xq = (int)((t >> 2) % sq), yq = (int)(((t >> 2) / sq) % sq);
    uint32_t face = (uint32_t)((t >> 2) / ((size_t)sq * sq));
    int x = xq - 1 + (k & 1), y = yq - 1 + (k >> 1);
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
    reinterpret_cast<H4*>(dst)[t] = reinterpret_cast<const H4*>(src)[((size_t)face * s + y) * s + x];
}

// Light table in LDS, structure-of-arrays: 9 planes of LSTRIDE floats
//   0..2 position, 3..5 color*intensity, 6..8 attenuation C0,C1,C2
// SoA (not 48-byte records) so that the SAME component of two different lights lands in an adjacent
// VGPR pair straight from two ds_read_b32 — the operand shape v_pk_*_f32 wants — with no repacking
// moves; the plane stride is a compile-time constant so the plane offset rides in the DS offset field.
constexpr int LIGHT_PLANES = 9;

// Which pixels of the tile a launch shades: up to SHADE_MAX_RECTS rectangles (tile-local), walked by ONE 1-D grid — a
// whole tile is one rectangle; the overlapped multi-GPU frame shades the tile's border ring (<= 4 rectangles) in one
// launch and its core in another.  Per rectangle the two-zone schedule of the kernel applies.
constexpr int SHADE_MAX_RECTS = 5;
struct ShadeRects {
    uint32_t n, rows_big, rows_small;   // rows of a long / of a short block (<= SHADE_ROWS)
    uint32_t x0[SHADE_MAX_RECTS], y0[SHADE_MAX_RECTS], w[SHADE_MAX_RECTS], h[SHADE_MAX_RECTS];
    uint32_t cols[SHADE_MAX_RECTS], nb_big[SHADE_MAX_RECTS], first[SHADE_MAX_RECTS + 1];   // first block of rect r; [n] = total
};

constexpr int SHADE_BLOCK = 256;
constexpr int SHADE_ROWS = 8;          // most rows of 256 pixels one block walks after staging its tables (12: no change, 16: +3 %, round 5)
constexpr int MAX_STAGED_TILES = 12;   // cluster (x,y) tiles whose 8 z-slices may be staged per block
// staged list: count, pad, 32 u16 indices = 34 halfwords (68 B) per cluster
constexpr int LIST_STRIDE = 34;         // dwords per staged cluster list: count, pad, 32 entries (8-byte aligned pairs)

typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 f2s(float a) { return f2{a, a}; }
__device__ __forceinline__ f2 max2(f2 a, f2 b) { return f2{fmaxf(a.x, b.x), fmaxf(a.y, b.y)}; }
typedef __attribute__((address_space(3))) const float lds_cf;   // a float in LDS: 32-bit addresses, ds_read with an immediate plane offset
__device__ __forceinline__ f2 rsq2(f2 a) { return f2{rsq(a.x), rsq(a.y)}; }
__device__ __forceinline__ f2 rcp2(f2 a) { return f2{rcp(a.x), rcp(a.y)}; }

// v_pk_mul_f32 with the clamp output modifier: clamp(a * b, 0, 1) on both halves, free of charge.  Used for the
// cosines N.L and N.H, which the shader clamps from below with max(., 0) and which cannot exceed 1 except by a
// rounding error of normalised vectors (1 + 1e-7 becomes 1).
// The s_nop on either side are REQUIRED: gfx950 needs one wait state between a transcendental (v_rsq / v_rcp) or a
// packed-fp32 instruction and a VALU instruction that reads its result.  The compiler inserts those wait states for
// the code it schedules, but it cannot see into an asm statement: without them this instruction consumed a v_rsq result
// issued immediately before it and read the register's OLD value in the high half (second light of the pair) — found
// as a parity failure that came and went with instruction scheduling.
__device__ __forceinline__ f2 mul2_sat(f2 a, f2 b) {
    f2 r;
    asm("s_nop 0\n\tv_pk_mul_f32 %0, %1, %2 clamp\n\ts_nop 0" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// {0, 0} in a register pair with one instruction.  (s_nop: the wait state a packed-fp32 consumer of the result needs, which the
// compiler cannot insert around asm — see mul2_sat.)
__device__ __forceinline__ f2 zero2() {
    f2 r;
    asm("v_pk_mov_b32 %0, 0, 0\n\ts_nop 0" : "=v"(r));
    return r;
}

struct alignas(8) H4x2 { H4 a, b; };   // two x-adjacent half4 texels (16 bytes, 8-byte aligned)
struct alignas(4) H2x2 { H2 a, b; };   // two x-adjacent LUT texels (8 bytes, 4-byte aligned)

// One pixel.  The kernel is bound by VALU ISSUE at 256 lights (SQ_ACTIVE_INST_VALU ~90 % of the launch), so the design
// constraints are instruction count and instruction class; measured issue costs on gfx950 at this kernel's occupancy of
// 5 waves per SIMD (tools/valu_rate3.hip -> profiles/r02_valu_rate3.txt, shader-clock cycles per wave-instruction per
// SIMD): v_fma_f32 2.6, v_mul/v_add 3.0, v_pk_{fma,mul,add}_f32 4.7, v_max/v_min 4.6, v_rcp/v_rsq 8.5; with one wave
// alone every plain op costs 7.5.  96 VGPRs (5 waves) is the measured optimum: 64 VGPRs (8 waves) spills heavily, 80 and
// 116 VGPRs time the same.  The pixel is shaded in phases that keep the live set small:
//   1. geometry: position, normal, view vector, roughness terms;
//   2. light loop: accumulates nine light-colour-weighted sums that do not depend on albedo/F0:
//        pl_c = Kdiff_c * S1_c + F0_c * spec_pix * S2_c + (1-F0_c) * spec_pix * S3_c
//        S1_c = sum col_c X (1-f5), S2_c = sum col_c X s, S3_c = sum col_c X s f5,   s = NdotL/(T A B)
//   3. material: re-reads the A/C planes (L2 hits) and folds the sums;
//   4. IBL: SH diffuse + split-sum specular from the padded env chain and the LUT.
template <bool STAGED_LISTS, int LSTRIDE, bool F32OUT>
__device__ __forceinline__ void shade_pixel(const ShadeParams& p, const float* llds, const uint32_t* lists, const uint32_t* mip_off,
                                            int tile_x0, int tile_y0, int tiles_x, int n_lights, int q_safe, uint32_t px, uint32_t py, float4 row) {
    // 32-bit element index (host-checked: pitch * rows * 16 < 2^32): a uniform base + one 32-bit lane offset per access instead of
    // 64-bit address arithmetic for every plane
    const uint32_t gi = __umul24(py, p.pitch) + px;
    auto at = [](const auto* base, uint32_t byte_off) { return *reinterpret_cast<std::remove_reference_t<decltype(*base)>*>(reinterpret_cast<const char*>(base) + byte_off); };
    if (at(p.stencil, gi) == 0) return;   // stencil ref 0 < value (DeferredPipeline.h:176-181)
    const float inv255 = 1.0f / 255.0f;   // UNORM8 -> float

    // ---- phase 1: geometry (vs_main :91-121, screen triangle D3D12Device.cpp:167-176; uv from the GLOBAL pixel)
    const float u = ((float)(p.x0 + px) + 0.5f) / (float)p.full_w;
    // v, cvv.y and the cluster row depend on the pixel ROW only: evaluated once per block row (k_deferred_shade, the same
    // expressions) and handed in as row = {v, cvv.y, cluster row} — an IEEE divide, a floor and their neighbours less per pixel
    V3 pos, view, n;
    float z_vs, roughness, depth_keep;
    {
        const uint32_t b = at(p.B, gi * 4u), c = at(p.C, gi * 4u);
        const float depth_ndc = at(p.depth, gi * 4u);
        const float ndc_x = 2.0f * u - 1.0f;
        const V3 cvv = v3(ndc_x * 0.5f * p.near_width, row.y, p.Near);
        const V3 camera_vec = v3(p.InvView[0] * cvv.x + p.InvView[1] * cvv.y + p.InvView[2] * cvv.z,
                                 p.InvView[3] * cvv.x + p.InvView[4] * cvv.y + p.InvView[5] * cvv.z,
                                 p.InvView[6] * cvv.x + p.InvView[7] * cvv.y + p.InvView[8] * cvv.z);
        roughness = (float)(c & 255u) * inv255;
        n = normalize3(decode_octahedron((float)(b & 255u) * inv255, (float)((b >> 8) & 255u) * inv255));
        // ViewSpaceDepth :74-77, ReconstructWorldPosition :79-83.  Quick form (one v_rcp, <= 2 ulp): the position is continuous in
        // it; the cluster slice, which is not, re-evaluates the exact IEEE sequence wherever it could matter (below)
        z_vs = view_space_depth_quick(depth_ndc, p.Near, p.Far);
        depth_keep = depth_ndc;
        const V3 cam = v3(p.CameraPos[0], p.CameraPos[1], p.CameraPos[2]);
        const float zs = z_vs * p.inv_near;   // not an IEEE divide: the position is continuous in it (the slice index below is not, and keeps its divides)
        pos = v3(cam.x + camera_vec.x * zs, cam.y + camera_vec.y * zs, cam.z + camera_vec.z * zs);
        view = normalize3(cam - pos);
    }
    const float NdV = dot3(n, view);
    const float NdotV = fmaxf(NdV, 0.0f);

    // ---- phase 2: clustered point lights :159-186.  ClusterIndex(uv, z), clustered.hlsli:45-60;
    // logf (not the fast intrinsic): the result is truncated to the slice index.
    float s1x = 0.0f, s1y = 0.0f, s1z = 0.0f, s2x = 0.0f, s2y = 0.0f, s2z = 0.0f, s3x = 0.0f, s3y = 0.0f, s3z = 0.0f;
    // STAGED_LISTS: every staged list holds at least one pair (empty ones: two null lights), so the walk is a do-while with no
    // branch around it — with one, the compiler zeroes the 18 accumulator registers on both sides of every branch (36 moves per pixel)
#ifndef PBR_EXP_NOLOOP   // (PBR_EXP_*: compile-time switches of tools/isa_phase_count.py, which sizes the phases of this function)
    if (STAGED_LISTS || n_lights > 0) {
        int sx = (int)floorf(u * (float)PBR_CLUSTER_X);
        int sy = (int)row.z;
        float zc = fminf(fmaxf(z_vs, p.Near), p.Far);
        // Slice index = (int)(Z * logf(zc / Near) / log(Far / Near)), a discontinuous function of the depth: its value must be the
        // shader's / the oracle's to the bit, which takes two IEEE divides and a full-precision logf (~45 instructions).  A quick
        // estimate t = slice_k * v_log_f32(zc * inv_near) is within 5e-6 of that expression's real value (1-ulp log2 of
        // magnitude <= ~14, two rounded constants), and so is the exact sequence's own result: wherever t is further than 1e-4
        // from an integer both truncate alike.  Only waves with a lane inside that band (2e-4 of the pixels) run the sequence —
        // from the depth texel on: the quick view-space depth above (v_rcp) is within 2 ulp of the IEEE one, i.e. within 3e-7 of
        // its slice coordinate, far inside the band.
        const float t_quick = p.slice_k * __builtin_amdgcn_logf(zc * p.inv_near);
        const float t_frac = t_quick - floorf(t_quick);
        int sz = (int)t_quick;
        if (__any(!(t_frac > 1.0e-4f && t_frac < 1.0f - 1.0e-4f))) {
            zc = fminf(fmaxf(view_space_depth(depth_keep, p.Near, p.Far), p.Near), p.Far);
            sz = (int)((float)PBR_CLUSTER_Z * logf(zc / p.Near) / p.log_far_near);
        }
        sx = clampi(sx, 0, PBR_CLUSTER_X - 1);
        sz = clampi(sz, 0, PBR_CLUSTER_Z - 1);
        // brdf() (brdf.hlsli:47-67) with D, G, the 4 NdotL NdotV denominator AND the attenuation under ONE reciprocal:
        //   D G / max(4 NdotL NdotV, 1e-4) = [a^2/pi * gV] * gl / (T * A),   a = roughness^2 (the shader's `a * a`),
        //   T  = max(t^2, 1e-6/pi), t = NdotH^2 (a^2 - 1) + 1,
        //   A  = NdotL(1-k)+k (>= 1/8: the shader's max(.,1e-6) never binds),
        //   gl = NdotL / max(4 NdotL NdotV, 1e-4) = min(1 / (4 NdotV), 1e4 NdotL)      (1 / max(a,b) = min(1/a, 1/b))
        //      = [1 / (4 NdotV)] * sat(NdotL * 4e4 NdotV): the saturation rides on a packed multiply (clamp modifier) where
        //      the min took two unpacked v_min, and the per-pixel factor 1 / (4 NdotV) moves out of the loop into spec_pix
        //      (where it cancels gV's NdotV: no division by NdotV is left),
        //   attenuation * NdotL = NdotL / Q;  with r = 1 / (Q T A):  1/Q = r T A,  so one v_rcp serves both; and T is
        //   carried as t^2 h2^2 (h2 = |L + V|^2), which removes the normalisation of H: 4 transcendentals per pair of lights.
        const float ra = roughness * roughness;
        const float a4m1 = ra * ra - 1.0f;
        const float k = (roughness + 1.0f) * (roughness + 1.0f) * 0.125f;
        const float one_k = 1.0f - k;
        const float c4 = 4.0e4f * NdotV;   // NdotV = 0: sat(0) = 0 here and spec_pix = 0 below, as gV = 0 makes the shader's term
        const float t_floor = EPSILON_F * INV_PI_F;
        // Instruction budget of the loop, from the measured issue costs (tools/valu_rate3.hip -> profiles/r02_valu_rate3.txt,
        // 5 waves per SIMD, cycles per wave-instruction per SIMD): v_pk_{fma,mul,add}_f32 4.7 (two lights per instruction),
        // plain v_fma 2.6 / v_mul 3.0, v_max / v_min 4.6, v_rsq / v_rcp 8.5.  Two lights per trip in the halves of packed
        // registers therefore buy ~1.2x per flop, not 2x; the SoA light planes put the same component of both lights into
        // an adjacent VGPR pair with no moves.  Per trip: 46 packed + 4 transcendental + the loop's compare and pointer step (+ 2 / 4 v_max on the slow paths).
        // the nine packed sums start at zero: ONE v_pk_mov_b32 per register pair (the compiler writes two v_mov_b32 per pair)
        f2 a1x = zero2(), a1y = zero2(), a1z = zero2(), a2x = zero2(), a2y = zero2(), a2z = zero2(), a3x = zero2(), a3y = zero2(), a3z = zero2();
        const float att_c0 = llds[6 * LSTRIDE], att_c1 = llds[7 * LSTRIDE], att_c2 = llds[8 * LSTRIDE];   // light 0's polynomial (ATT: everyone's)
        auto light2 = [&](auto q_safe, auto t_safe, const lds_cf* la, const lds_cf* lb, auto att_uniform) {
            constexpr bool QSAFE = decltype(q_safe)::value, TSAFE = decltype(t_safe)::value, ATT = decltype(att_uniform)::value;
            auto comp = [&](int c) { return f2{la[c * LSTRIDE], lb[c * LSTRIDE]}; };
            const f2 dx = comp(0) - f2s(pos.x), dy = comp(1) - f2s(pos.y), dz = comp(2) - f2s(pos.z);
            const f2 d2 = dx * dx + dy * dy + dz * dz;
            const f2 invd = rsq2(d2);
            const f2 dn = dx * n.x + dy * n.y + dz * n.z;
            const f2 NdotL = mul2_sat(dn, invd);   // max(N.L, 0)
            // |L + V|^2 is summed from the components of L + V, NOT taken as 2 + 2 L.V: at grazing incidence (L ~ -V,
            // |L + V|^2 ~ 1e-2) the shortcut cancels and its 1e-7 error becomes 1e-5 of N.H, which a GGX highlight at
            // roughness 0.2 (t ~ 2e-3) multiplies by 4 / t: several per cent of D (measured against the oracle on a 4K
            // band: 3 pixels of 122 880).  One packed instruction more than the shortcut.
            const f2 wx = dx * invd + f2s(view.x), wy = dy * invd + f2s(view.y), wz = dz * invd + f2s(view.z);
            // (the 1e-15 keeps h2 > 0 when L = -V exactly, so that the shared reciprocal below stays finite; it is
            //  ten orders of magnitude below any |L + V|^2 that contributes light)
            const f2 h2 = wx * wx + (wy * wy + (wz * wz + f2s(1.0e-15f)));
            // GGX without normalising H at all: with nw = N.(L + V) and h2 = |L + V|^2, N.H^2 = nw^2 / h2 and
            //   t = N.H^2 (a^2 - 1) + 1 = tn / h2,  tn = nw^2 (a^2 - 1) + h2,   so  1 / t^2 = h2^2 / tn^2
            // which joins the one reciprocal of the trip: no v_rsq for H.  nw < 0 (the shader clamps N.H to 0 there)
            // needs no case: N.L > 0 then forces N.V < 0, which zeroes the specular term through spec_pix.
            const f2 nw = dn * invd + f2s(NdV);
            const f2 tn = (nw * nw) * a4m1 + h2;
            const f2 h4 = h2 * h2;
            // TSAFE: every active lane of the wave has a^2 >= 6e-4, hence t >= a^2 - 6e-8 > sqrt(1e-6 / pi) = 5.64e-4 and the
            // shader's max(pi t^2, 1e-6) never binds (decided per wave before the walk)
            f2 Tn = tn * tn;                                 // t^2 h2^2
            if constexpr (!TSAFE) Tn = max2(Tn, h4 * t_floor);
            const f2 A = NdotL * one_k + f2s(k);
            // attenuation(): max(C0 + C1 d + C2 d^2, 1e-6).  QSAFE: every staged light has C0 >= 1e-6 and C1, C2 >= 0,
            // so the floor never binds (checked once per block while the table is staged)
            f2 Q;
            // C0 + C1 d + C2 d^2 with d = d2 / sqrt(d2) never formed: d2 (C1 invd + C2) + C0 — two fused operations
            if constexpr (ATT) Q = d2 * (invd * f2s(att_c1) + f2s(att_c2)) + f2s(att_c0);   // the scene's one polynomial (the null light takes it too: its colour is 0 and Q stays finite)
            else Q = d2 * (invd * comp(7) + comp(8)) + comp(6);
            if constexpr (!QSAFE) Q = max2(Q, f2s(EPSILON_F));
            const f2 TA = Tn * A;
            const f2 r = rcp2(Q * TA);                       // 1 / (Q A t^2 h2^2)
            const f2 q = NdotL * r;
            const f2 X = q * TA;                             // attenuation * NdotL = NdotL / Q
            const f2 gs = mul2_sat(NdotL, f2s(c4));          // gl * 4 NdotV
            // fresnel on NdotL (Q3).  The shader's max(1-NdotL, 1e-6) only matters within 1e-6 of NdotL = 1, where it
            // changes f5 by < 1e-30: dropped.
            const f2 fm = f2s(1.0f) - NdotL;
            const f2 fm2 = fm * fm;
            const f2 f5 = fm2 * fm2 * fm;
            const f2 w2 = (q * gs) * h4;                     // X * gl / (t^2 A) * 4 NdotV
            const f2 w1 = X - X * f5;                        // X * (1 - f5)
            const f2 w3 = w2 * f5;
            const f2 cr = comp(3), cg = comp(4), cb = comp(5);
            a1x += cr * w1; a1y += cg * w1; a1z += cb * w1;
            a2x += cr * w2; a2y += cg * w2; a2z += cb * w2;
            a3x += cr * w3; a3y += cg * w3; a3z += cb * w3;
        };
        const lds_cf* const ltab = (const lds_cf*)llds;      // the staged light planes, as an LDS (address space 3) pointer
        if (STAGED_LISTS) {
            const uint32_t* my = lists + __mul24(__mul24(__mul24(sy - tile_y0, tiles_x) + (sx - tile_x0), PBR_CLUSTER_Z) + sz, LIST_STRIDE);
            // staged lists are padded to an even count with the null light (black, far away): no odd tail.  An entry is
            // the LDS BYTE ADDRESS of the light's first plane (table base + 4 * index), a dword of its own: packed two to a dword
            // the unpacking mask + shift were two more VALU issues per trip
            const int nl = my[0];
            const bool t_ok = __all(ra * ra >= 6.0e-4f) != 0;
            auto walk = [&](auto qs, auto ts, auto au) {
                int i = 0;
                do {   // nl >= 2.  One 8-byte LDS read = the two addresses of the trip, each in its own register
                    const uint2 pair = *reinterpret_cast<const uint2*>(my + 2 + i);   // i even -> 8-byte aligned
                    light2(qs, ts, (const lds_cf*)(uintptr_t)pair.x, (const lds_cf*)(uintptr_t)pair.y, au);
                    i += 2;
                } while (i < nl);
                // (reading the NEXT trip's two entries one trip ahead was measured in round 4: two more live registers push the
                //  kernel's scratch from 12 to 24 bytes per lane at its 96-VGPR budget, in-frame 0.3356 -> 0.3470 ms; removed —
                //  profiles/r04_d_ab_shade_prefetch.txt)
            };
            if (q_safe & 1) {
                if (t_ok) { if (q_safe & 2) walk(std::true_type{}, std::true_type{}, std::true_type{}); else walk(std::true_type{}, std::true_type{}, std::false_type{}); }
                else if (q_safe & 2) walk(std::true_type{}, std::false_type{}, std::true_type{});   // a rough-enough wave is a property of the PIXELS, one polynomial of the SCENE: independent
                else walk(std::true_type{}, std::false_type{}, std::false_type{});
            } else walk(std::false_type{}, std::false_type{}, std::false_type{});
        } else {
            const pbr_cluster* cl = p.clusters + (sz + sx * PBR_CLUSTER_Z + sy * PBR_CLUSTER_X * PBR_CLUSTER_Z);
            const int nl = min(max(cl->NumLights, 0), PBR_MAX_LIGHTS_PER_CLUSTER);
            auto idx = [&](int q) { return q < nl ? min(max(cl->LightIndex[q], 0), n_lights - 1) : n_lights; };   // n_lights = the null light
            for (int i = 0; i < nl; i += 2) light2(std::false_type{}, std::false_type{}, ltab + idx(i), ltab + idx(i + 1), std::false_type{});
        }
        s1x = a1x.x + a1x.y; s1y = a1y.x + a1y.y; s1z = a1z.x + a1z.y;
        s2x = a2x.x + a2x.y; s2y = a2y.x + a2y.y; s2z = a2z.x + a2z.y;
        s3x = a3x.x + a3x.y; s3y = a3y.x + a3y.y; s3z = a3z.x + a3z.y;
    }
#endif

    // ---- phase 3: material terms (planes A and C re-read: L2 hits, keeps them out of the loop's registers)
    const uint32_t a = at(p.A, gi * 4u);
    const float metallic = (float)((at(p.C, gi * 4u) >> 8) & 255u) * inv255;
    const V3 albedo = v3((float)(a & 255u) * inv255, (float)((a >> 8) & 255u) * inv255, (float)((a >> 16) & 255u) * inv255);
    const float emission = (float)(a >> 24) * inv255;
    const V3 F0 = v3(0.04f + metallic * (albedo.x - 0.04f), 0.04f + metallic * (albedo.y - 0.04f), 0.04f + metallic * (albedo.z - 0.04f));
    V3 out;
    {
        const float ra = roughness * roughness;
        const float k = (roughness + 1.0f) * (roughness + 1.0f) * 0.125f;
        // a^2 / pi * gV / (4 NdotV), gV = NdotV / max(NdotV (1-k) + k, 1e-6): the light sums S2, S3 carry gl * 4 NdotV
        const float spec_pix = NdotV > 0.0f ? ra * ra * (0.25f * INV_PI_F) * rcp(fmaxf(NdotV * (1.0f - k) + k, EPSILON_F)) : 0.0f;
        const float kd = (1.0f - metallic) * INV_PI_F;   // Kd*albedo/pi = (1-F0)(1-m) albedo/pi * (1-f5)
        out.x = (1.0f - F0.x) * (kd * albedo.x * s1x + spec_pix * s3x) + F0.x * spec_pix * s2x;
        out.y = (1.0f - F0.y) * (kd * albedo.y * s1y + spec_pix * s3y) + F0.y * spec_pix * s2y;
        out.z = (1.0f - F0.z) * (kd * albedo.z * s1z + spec_pix * s3z) + F0.z * spec_pix * s2z;
        // emission (Q1: the directional light of :144-156 is computed by the reference but never added)
        out = out + albedo * emission;

        // ---- phase 4a: EnvironmentDiffuse :23-54
#ifndef PBR_EXP_NOSH
        const float bx = n.x * n.y, by = n.y * n.z, bz = n.z * n.z, bw = n.z * n.x;
        const float cc = n.x * n.x - n.y * n.y;
        const pbr_sh_pack& s = p.sh;
        const float ir = (s.sha_r[0] * n.x + s.sha_r[1] * n.y + s.sha_r[2] * n.z + s.sha_r[3]) +
                         ((s.shb_r[0] * bx + s.shb_r[1] * by + s.shb_r[2] * bz + s.shb_r[3] * bw) + s.shc[0] * cc);
        const float ig = (s.sha_g[0] * n.x + s.sha_g[1] * n.y + s.sha_g[2] * n.z + s.sha_g[3]) +
                         ((s.shb_g[0] * bx + s.shb_g[1] * by + s.shb_g[2] * bz + s.shb_g[3] * bw) + s.shc[1] * cc);
        const float ib = (s.sha_b[0] * n.x + s.sha_b[1] * n.y + s.sha_b[2] * n.z + s.sha_b[3]) +
                         ((s.shb_b[0] * bx + s.shb_b[1] * by + s.shb_b[2] * bz + s.shb_b[3] * bw) + s.shc[2] * cc);
        out.x += albedo.x * kd * ir;
        out.y += albedo.y * kd * ig;
        out.z += albedo.z * kd * ib;
#endif
    }

    // ---- phase 4b: EnvironmentSpecular :56-70 — padded env chain: each bilinear row is one 16-byte pair
#ifndef PBR_EXP_NOIBL
    {
#ifndef PBR_EXP_NOENV
        const V3 R = normalize3(n * (2.0f * NdV) - view);
        float lod = roughness * (float)PBR_ENV_MIPS;   // Q4: roughness*5 on a 5-mip chain
        lod = snap8(fminf(fmaxf(lod, 0.0f), (float)(p.env_mips - 1)));   // 8-bit LOD fraction
        const float fl = floorf(lod);
        const uint32_t l0 = (uint32_t)fl, l1 = min(l0 + 1, p.env_mips - 1);
        const float env_f = lod - fl;
        // face + uv of R (D3D cube addressing).  v_rcp instead of an IEEE divide: u,v only feed a continuous filter
        uint32_t face;
        float cu, cv;
        {
            // the cube-map coordinate instructions (v_cubeid / v_cubesc / v_cubetc / v_cubema: D3D face order and (sc, tc)
            // table; ma = 2 x the signed major axis): 4 instructions for the three-way compare-and-select.  On an exact tie
            // of two |components| they pick z, then y, then x where the shader's HLSL picks x, then y, then z — the same
            // point on the shared edge of two faces, which the seamless footprints filter alike.
            const float ma = 0.5f * fabsf(__builtin_amdgcn_cubema(R.x, R.y, R.z));
            const float sc = __builtin_amdgcn_cubesc(R.x, R.y, R.z), tc = __builtin_amdgcn_cubetc(R.x, R.y, R.z);
            face = (uint32_t)__builtin_amdgcn_cubeid(R.x, R.y, R.z);
            const float inv = rcp(ma);
            cu = (sc * inv + 1.0f) * 0.5f;
            cv = (tc * inv + 1.0f) * 0.5f;
        }
        // One trilinear sample = eight texels x eight weights.  The shader's nested lerps (x, then y, then level) cost 2 products per
        // lerp and channel (47 instructions for the three channels); as a weighted sum — weights (1-fx | fx)(1-fy | fy)(1-f | f),
        // 10 multiplies once — each texel is ONE v_fma_mix_f32 per channel (34 instructions).  The two forms differ by rounding
        // only (~3 ulp of fp32 on a convex combination of fp16 texels).
        struct Foot { H4x2 r0, r1; float w00, w10, w01, w11; };
        auto foot = [&](uint32_t l, uint32_t mip_off, float wl) {
            const int s = (int)(p.env_size >> l), sq = s + 1;
            // u in [0,1] -> texel coordinate in [-0.5, s-0.5]: no NaN / range guard needed here; x.8 fixed-point snap
            const float fxp = snap8(cu * (float)s) - 0.5f, fyp = snap8(cv * (float)s) - 0.5f;
            const float flx = floorf(fxp), fly = floorf(fyp);
            const float fx = fxp - flx, fy = fyp - fly;
            // footprint layout: the four texels of this tap are 32 contiguous bytes.  24-bit multiplies (every factor < 2^24;
            // v_mul_lo_u32 / v_mad_u64_u32 are multi-pass instructions) and a 32-bit byte offset from the chain's base
            // (host-checked: the padded chain is smaller than 4 GiB)
            const uint32_t o = __umul24(__umul24(face, (uint32_t)sq) + (uint32_t)((int)fly + 1), (uint32_t)sq) + (uint32_t)((int)flx + 1);
            const char* q = reinterpret_cast<const char*>(p.env) + (mip_off + o * 4u) * 8u;
            Foot f;
            f.r0 = *reinterpret_cast<const H4x2*>(q);
            f.r1 = *reinterpret_cast<const H4x2*>(q + 16);
            const float wy1 = fy * wl, wy0 = wl - wy1;         // (1 - fy) wl, fy wl
            f.w10 = fx * wy0; f.w00 = wy0 - f.w10;             // (1 - fx)(1 - fy) wl, fx (1 - fy) wl
            f.w11 = fx * wy1; f.w01 = wy1 - f.w11;
            return f;
        };
        const Foot fa = foot(l0, mip_off[l0], 1.0f - env_f);   // per-lane index: the table sits in LDS (an SGPR array would spill to scratch)
        const Foot fb = foot(l1, mip_off[l1], env_f);
        auto chan = [&](h16 a00, h16 a10, h16 a01, h16 a11, h16 b00, h16 b10, h16 b01, h16 b11) {
            float r = __builtin_fmaf((float)a00, fa.w00, 0.0f);   // fma(x, w, 0): the first product as v_fma_mix_f32 too (no separate v_cvt)
            r = __builtin_fmaf((float)a10, fa.w10, r);
            r = __builtin_fmaf((float)a01, fa.w01, r);
            r = __builtin_fmaf((float)a11, fa.w11, r);
            r = __builtin_fmaf((float)b00, fb.w00, r);
            r = __builtin_fmaf((float)b10, fb.w10, r);
            r = __builtin_fmaf((float)b01, fb.w01, r);
            return __builtin_fmaf((float)b11, fb.w11, r);
        };
        const V3 envc = v3(chan(fa.r0.a.x, fa.r0.b.x, fa.r1.a.x, fa.r1.b.x, fb.r0.a.x, fb.r0.b.x, fb.r1.a.x, fb.r1.b.x),
                           chan(fa.r0.a.y, fa.r0.b.y, fa.r1.a.y, fa.r1.b.y, fb.r0.a.y, fb.r0.b.y, fb.r1.a.y, fb.r1.b.y),
                           chan(fa.r0.a.z, fa.r0.b.z, fa.r1.a.z, fa.r1.b.z, fb.r0.a.z, fb.r0.b.z, fb.r1.a.z, fb.r1.b.z));
#else
        const V3 envc = v3(0.5f, 0.5f, 0.5f);
#endif
#ifndef PBR_EXP_NOLUT
        // LUT bilinear with clamp addressing (Q5): one 8-byte pair per row; at the borders both taps are
        // the same texel, picked out of the pair that stays inside the row
        // Both coordinates lie in [0, 1] (UNORM8 roughness, clamped N.V): no NaN / range guard; x.8 fixed-point snap as in
        // pbr_device.hpp::bilinear_coord.  The row pair (xb, xb + 1) always lies inside the row; where the sampler's clamp makes
        // both taps the SAME texel (the first / last half texel) the weight is moved onto it (0 or 1) instead of selecting texels.
        const int lr = (int)p.lut_res;
        const float lrf = (float)lr;
        const float xs = snap8(roughness * lrf) - 0.5f, ys = snap8(NdotV * lrf) - 0.5f;
        const float xfl = floorf(xs), yfl = floorf(ys);
        const int xi = (int)xfl, yi = (int)yfl;
        const int y0 = clampi(yi, 0, lr - 1), y1 = clampi(yi + 1, 0, lr - 1);
        const int xb = clampi(xi, 0, max(lr - 2, 0));
        const float fx = xi < 0 ? 0.0f : (xi > lr - 2 ? 1.0f : xs - xfl), fy = ys - yfl;
        const H2* lut = reinterpret_cast<const H2*>(p.lut);
        H2x2 lt0, lt1;
        if (lr > 1) {
            const char* lb8 = reinterpret_cast<const char*>(lut);   // 32-bit byte offsets: lr <= 16384 (host-checked)
            lt0 = *reinterpret_cast<const H2x2*>(lb8 + (__umul24((uint32_t)y0, (uint32_t)lr) + (uint32_t)xb) * 4u);
            lt1 = *reinterpret_cast<const H2x2*>(lb8 + (__umul24((uint32_t)y1, (uint32_t)lr) + (uint32_t)xb) * 4u);
        } else {
            lt0.a = lt0.b = lut[0];
            lt1 = lt0;
        }
        const float wx0 = 1.0f - fx, wy0 = 1.0f - fy;
        auto xl = [&](h16 a, h16 b) { return __builtin_fmaf((float)b, fx, __builtin_fmaf((float)a, wx0, 0.0f)); };   // two v_fma_mix_f32, see fetch
        const float la = xl(lt0.a.x, lt0.b.x) * wy0 + xl(lt1.a.x, lt1.b.x) * fy;
        const float lb = xl(lt0.a.y, lt0.b.y) * wy0 + xl(lt1.a.y, lt1.b.y) * fy;
#else
        const float la = 0.5f, lb = 0.25f;
#endif
        out.x += envc.x * (F0.x * la + lb);
        out.y += envc.y * (F0.y * la + lb);
        out.z += envc.z * (F0.z * la + lb);
    }
#endif
    const uint32_t ho = __umul24(py, p.hdr_pitch) + px;
    if (F32OUT) *reinterpret_cast<float4*>(reinterpret_cast<char*>(p.hdr_f32) + ho * 16u) = make_float4(out.x, out.y, out.z, 1.0f);
    else store_h4(reinterpret_cast<pbr_half*>(reinterpret_cast<char*>(p.hdr) + ho * 8u), f4(out.x, out.y, out.z, 1.0f));
}

#ifdef PBR_SHADE_TIMING   // experiment build only (tools/shade_timeline.py): 100 MHz wall-clock stamps of every block
__device__ unsigned long long g_shade_item_stamp[4 * 160000];    // per block: start, tables staged, end (last wave), (rows << 48 | block)
extern "C" int pbr_debug_shade_stamps(unsigned long long* blocks, int n_blocks, unsigned long long* items, int n_items) {
    (void)blocks; (void)n_blocks;
    return (int)hipMemcpyFromSymbol(items, HIP_SYMBOL(g_shade_item_stamp), sizeof(unsigned long long) * 4 * n_items);
}
extern "C" int pbr_debug_shade_stamps_reset() {
    void* i = nullptr;
    int e = (int)hipGetSymbolAddress(&i, HIP_SYMBOL(g_shade_item_stamp));
    if (e == 0) e = (int)hipMemset(i, 0, sizeof(g_shade_item_stamp));
    return e;
}
#define SHADE_ISTAMP(it, i, v) do { if (threadIdx.x == 0 && (it) < 160000u) g_shade_item_stamp[(it) * 4 + (i)] = (v); } while (0)
// end stamp: the LAST wave of the block to get there (the array is zeroed before the launch)
#define SHADE_ISTAMP_MAX(it, i, v) do { if ((threadIdx.x & 63) == 0 && (it) < 160000u) atomicMax(&g_shade_item_stamp[(it) * 4 + (i)], (v)); } while (0)
#define SHADE_NOW() wall_clock64()
#else
#define SHADE_ISTAMP(it, i, v) do {} while (0)
#define SHADE_ISTAMP_MAX(it, i, v) do {} while (0)
#define SHADE_NOW() 0ull
#endif

// grid: rc.first[rc.n] blocks of 256 threads, dispatched in index order (the hardware's dispatcher is the work queue: a software queue
// of persistent blocks was built and measured in round 6 and lost to it at every size — EXPERIMENTS.md).
// dynamic LDS: 9 planes * LSTRIDE floats of light data, then (STAGED_LISTS) max_clusters * 136 B of light lists.
template <bool STAGED_LISTS, int LSTRIDE, bool F32OUT>
#ifndef SHADE_MIN_WAVES
#define SHADE_MIN_WAVES 5   // 96 VGPRs; 2 dwords of scratch per lane are spilled OUTSIDE the light loop.  Best of 4..8 measured (tools/probe_shade.py)
#endif
// (an exact register budget cannot be asked for: `amdgpu_num_vgpr(104)` is ignored beside the waves-per-EU bound on gfx950 — round 6, EXPERIMENTS.md)
__global__ __launch_bounds__(SHADE_BLOCK, SHADE_MIN_WAVES) void k_deferred_shade(ShadeParams p, int n_lights, int max_clusters, ShadeRects rc) {
    extern __shared__ float4 lds_raw[];
    __shared__ uint32_t s_mip_off[16];
    const unsigned long long t_start = SHADE_NOW();
    (void)t_start;
    if (threadIdx.x < 16) s_mip_off[threadIdx.x] = p.env_mip_off[threadIdx.x];
    float* llds = reinterpret_cast<float*>(lds_raw);
    uint32_t* lists = reinterpret_cast<uint32_t*>(llds + ((LIGHT_PLANES * LSTRIDE + 1) & ~1));   // 8-byte aligned
    int my_safe = 1, my_same = 1;
    const float att0 = n_lights > 0 ? p.lights[0].C0 : 1.0f, att1 = n_lights > 0 ? p.lights[0].C1 : 0.0f, att2 = n_lights > 0 ? p.lights[0].C2 : 0.0f;
    if (threadIdx.x == 0) {   // the null light: pads odd lists; black, so its pair lane contributes exactly 0
        llds[0 * LSTRIDE + n_lights] = 1.0e15f; llds[1 * LSTRIDE + n_lights] = 1.0e15f; llds[2 * LSTRIDE + n_lights] = 1.0e15f;
        llds[3 * LSTRIDE + n_lights] = 0.0f; llds[4 * LSTRIDE + n_lights] = 0.0f; llds[5 * LSTRIDE + n_lights] = 0.0f;
        llds[6 * LSTRIDE + n_lights] = 1.0f; llds[7 * LSTRIDE + n_lights] = 0.0f; llds[8 * LSTRIDE + n_lights] = 0.0f;
    }
    for (int i = threadIdx.x; i < n_lights; i += SHADE_BLOCK) {
        const pbr_light l = p.lights[i];
        my_safe &= (l.C0 >= EPSILON_F) & (l.C1 >= 0.0f) & (l.C2 >= 0.0f);
        my_same &= (l.C0 == att0) & (l.C1 == att1) & (l.C2 == att2);
        llds[0 * LSTRIDE + i] = l.Position[0];
        llds[1 * LSTRIDE + i] = l.Position[1];
        llds[2 * LSTRIDE + i] = l.Position[2];
        llds[3 * LSTRIDE + i] = l.Color[0] * l.Intensity;
        llds[4 * LSTRIDE + i] = l.Color[1] * l.Intensity;
        llds[5 * LSTRIDE + i] = l.Color[2] * l.Intensity;
        llds[6 * LSTRIDE + i] = l.C0;
        llds[7 * LSTRIDE + i] = l.C1;
        llds[8 * LSTRIDE + i] = l.C2;
    }
    // block -> rectangle -> (column block, row block); wave-uniform scalar arithmetic
    uint32_t r = 0;
    while (r + 1 < rc.n && blockIdx.x >= rc.first[r + 1]) r++;
    const uint32_t lb = blockIdx.x - rc.first[r];
    const uint32_t bx0 = rc.x0[r] + (lb % rc.cols[r]) * SHADE_BLOCK, x_end = rc.x0[r] + rc.w[r];
    // two-zone schedule: the first nb_big block rows walk rows_big rows each, the rest rows_small — the short
    // blocks are dispatched last and fill the tail of the launch (a 4K frame is only ~3.2 waves of resident blocks)
    const uint32_t by = lb / rc.cols[r], nb_big = rc.nb_big[r], rows_big = rc.rows_big, rows_small = rc.rows_small;
    const uint32_t y_begin = rc.y0[r] + (by < nb_big ? by * rows_big : nb_big * rows_big + (by - nb_big) * rows_small);
    const uint32_t y_end = min(y_begin + (by < nb_big ? rows_big : rows_small), rc.y0[r] + rc.h[r]);
    int tile_x0 = 0, tile_y0 = 0, tiles_x = 1;
    const uint32_t lds_base = (uint32_t)(uintptr_t)(lds_cf*)llds;   // < 64 KiB: the whole light table (<= 9 x 1025 floats) stays addressable in 16 bits
    if (STAGED_LISTS) {
        // cluster (x,y) tiles the block's pixel rectangle can fall into — same arithmetic as the per-pixel
        // ClusterIndex (floor(u*24), floor((1-v)*16)); monotone in the pixel coordinate, so the corners bound it
        const uint32_t bx1 = min(bx0 + SHADE_BLOCK, x_end) - 1;
        auto tx = [&](uint32_t x) { return clampi((int)floorf((((float)(p.x0 + x) + 0.5f) / (float)p.full_w) * (float)PBR_CLUSTER_X), 0, PBR_CLUSTER_X - 1); };
        auto ty = [&](uint32_t y) { return clampi((int)floorf((1.0f - ((float)(p.y0 + y) + 0.5f) / (float)p.full_h) * (float)PBR_CLUSTER_Y), 0, PBR_CLUSTER_Y - 1); };
        tile_x0 = tx(bx0);
        const int tile_x1 = tx(bx1);
        const int ty_a = ty(y_begin), ty_b = ty(y_end - 1);
        tile_y0 = min(ty_a, ty_b);
        const int tile_y1 = max(ty_a, ty_b);
        tiles_x = tile_x1 - tile_x0 + 1;
        const int n_cl = min(tiles_x * (tile_y1 - tile_y0 + 1) * PBR_CLUSTER_Z, max_clusters);   // host sized the LDS for the worst case
        {
            static_assert(PBR_CLUSTER_Z == 8 && PBR_MAX_LIGHTS_PER_CLUSTER == 32, "staging map");
#ifndef PBR_EXP_STAGE_R5   // (round 5's form below: a thread per entry, n_cl / 8 rounds of ~40 instructions; kept for A/B builds)
            // A thread owns FOUR consecutive entries of cluster (tid >> 3) + 32 k: one 16-byte load of indices per thread and round,
            // n_cl / 32 rounds.  The prologue is instruction issue, not latency: a new block's waves share their SIMDs with four blocks in
            // the middle of their pixel work, and ~700 prologue instructions per thread were 6 % of a 256-light block's work, 15 % of a
            // single-light one's (block timelines, profiles/r06_i_timeline_*: "staging" 5 - 12 us whether or not the loads are batched).
            // Entries from the list's (even) length on are never read by the walk: their threads skip the conversion.
            struct __attribute__((packed, aligned(4))) Idx4 { int32_t x, y, z, w; };
            const int part = threadIdx.x & 7;
            const float inv_tiles_x = rcp((float)tiles_x);
            for (int c = threadIdx.x >> 3; c < n_cl; c += SHADE_BLOCK / 8) {
                const int z = c & 7, t = c >> 3;
                // t / tiles_x, both <= MAX_STAGED_TILES: (t + 1/2) / tiles_x is never within 0.04 of an integer, the approximate reciprocal is exact enough
                const int ty_ = (int)(((float)t + 0.5f) * inv_tiles_x), cx = tile_x0 + (t - ty_ * tiles_x), cy = tile_y0 + ty_;
                const pbr_cluster* cl = p.clusters + (z + cx * PBR_CLUSTER_Z + cy * PBR_CLUSTER_X * PBR_CLUSTER_Z);
                const int cnt = n_lights > 0 ? min(max(cl->NumLights, 0), PBR_MAX_LIGHTS_PER_CLUSTER) : 0;
                const Idx4 v = *reinterpret_cast<const Idx4*>(cl->LightIndex + 4 * part);
                uint32_t* l = lists + c * LIST_STRIDE;
                if (4 * part < max(cnt, 2)) {
                    auto entry = [&](int k, int raw) { return lds_base + 4u * (uint32_t)(4 * part + k < cnt ? min(max(raw, 0), n_lights - 1) : n_lights); };   // never index past the staged table
                    uint2* e = reinterpret_cast<uint2*>(l + 2 + 4 * part);   // 136 c + 8 + 16 part bytes: 8-byte aligned
                    e[0] = make_uint2(entry(0, v.x), entry(1, v.y));
                    e[1] = make_uint2(entry(2, v.z), entry(3, v.w));
                }
                if (part == 0) *reinterpret_cast<uint2*>(l) = make_uint2((uint32_t)max((cnt + 1) & ~1, 2), 0u);
            }
#else
            // a thread owns entry (tid & 31) of cluster (tid >> 5) + 8 k: no division by the list stride, one index load per entry (round 5)
            const int j = threadIdx.x & 31;
            for (int c = threadIdx.x >> 5; c < n_cl; c += SHADE_BLOCK / 32) {
                const int z = c & 7, t = c >> 3;
                const int ty_ = t / tiles_x, cx = tile_x0 + (t - ty_ * tiles_x), cy = tile_y0 + ty_;
                const pbr_cluster* cl = p.clusters + (z + cx * PBR_CLUSTER_Z + cy * PBR_CLUSTER_X * PBR_CLUSTER_Z);
                const int cnt = n_lights > 0 ? min(max(cl->NumLights, 0), PBR_MAX_LIGHTS_PER_CLUSTER) : 0;
                uint32_t* l = lists + c * LIST_STRIDE;
                const int li = j < cnt ? min(max(cl->LightIndex[j], 0), n_lights - 1) : n_lights;   // never index past the staged table
                l[2 + j] = lds_base + 4u * (uint32_t)li;
                if (j < 2) l[j] = j == 0 ? (uint32_t)max((cnt + 1) & ~1, 2) : 0u;
            }
#endif
        }
    }
    // per-row terms of the block's <= SHADE_ROWS rows (vs_main :91-95, ClusterIndex clustered.hlsli:47): {v, cvv.y, cluster row}
    __shared__ float4 s_row[SHADE_ROWS];
    if (threadIdx.x < (uint32_t)SHADE_ROWS) {
        const float v = ((float)(p.y0 + y_begin + threadIdx.x) + 0.5f) / (float)p.full_h;
        const float ndc_y = 1.0f - 2.0f * v;
        s_row[threadIdx.x] = make_float4(v, ndc_y * 0.5f * p.near_height, (float)clampi((int)floorf((1.0f - v) * (float)PBR_CLUSTER_Y), 0, PBR_CLUSTER_Y - 1), 0.0f);
    }
    // bit 0: the attenuation floor cannot bind; bit 1: every staged light has the SAME attenuation polynomial (one radius for the
    // whole scene is common), so its three coefficients are per-kernel constants and a trip reads 13 LDS dwords instead of 19
    const int q_safe = (__syncthreads_and(my_safe) != 0 ? 1 : 0) | (__syncthreads_and(my_same) != 0 ? 2 : 0);
    SHADE_ISTAMP(blockIdx.x, 0, t_start);
    SHADE_ISTAMP(blockIdx.x, 1, SHADE_NOW());
    SHADE_ISTAMP(blockIdx.x, 3, ((unsigned long long)(y_end - y_begin) << 48) | blockIdx.x);
    const uint32_t px = bx0 + threadIdx.x;
#ifndef PBR_SHADE_TIMING
    if (px >= x_end) return;
    for (uint32_t py = y_begin; py < y_end; py++)
        shade_pixel<STAGED_LISTS, LSTRIDE, F32OUT>(p, llds, lists, s_mip_off, tile_x0, tile_y0, tiles_x, n_lights, q_safe, px, py, s_row[py - y_begin]);
#else
    if (px < x_end)
        for (uint32_t py = y_begin; py < y_end; py++)
            shade_pixel<STAGED_LISTS, LSTRIDE, F32OUT>(p, llds, lists, s_mip_off, tile_x0, tile_y0, tiles_x, n_lights, q_safe, px, py, s_row[py - y_begin]);
    SHADE_ISTAMP_MAX(blockIdx.x, 2, SHADE_NOW());
#endif
}

extern "C" {

pbr_status pbr_env_pad(pbr_ctx* ctx, const pbr_half* env, uint32_t size, uint32_t mips, pbr_half* out_padded) {
    if (!ctx) return PBR_ERR_INVALID;
    PBR_REQUIRE(ctx, env && out_padded, "pbr_env_pad: null pointer");
    PBR_REQUIRE(ctx, size >= 1 && size <= 8192 && mips >= 1 && mips <= 16 && (size >> (mips - 1)) >= 1, "pbr_env_pad: bad size/mips");
    for (uint32_t m = 0; m < mips; m++) {
        const int s = (int)(size >> m);
        const size_t n = (size_t)6 * (s + 1) * (s + 1) * 4;
        hipLaunchKernelGGL(k_env_pad, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                           env + 4 * cube_mip_offset(size, m), out_padded + 4 * env_padded_mip_offset(size, m), s);
        pbr_status r = launched(ctx, "k_env_pad");
        if (r) return r;
    }
    return PBR_OK;
}

}  // extern "C"

template <bool F32OUT>
static pbr_status shade_launch(pbr_ctx* ctx, const pbr_global* g, const pbr_tile* tile, const pbr_gbuffer* gb,
                               const pbr_half* lut, uint32_t lut_res,
                               const pbr_half* env, uint32_t env_size, uint32_t env_mips,
                               const pbr_cluster* clusters, const pbr_light* lights, int num_lights,
                               pbr_half* hdr, float* hdr_f32, uint32_t hdr_pitch,
                               const uint32_t (*rects)[4] = nullptr, uint32_t n_rects = 0) {
    if (!ctx) return PBR_ERR_INVALID;
    PBR_REQUIRE(ctx, g && tile && gb && lut && env && clusters && (F32OUT ? (const void*)hdr_f32 : (const void*)hdr), "pbr_deferred_shade: null pointer");
    PBR_REQUIRE(ctx, gb->A && gb->B && gb->C && gb->depth && gb->stencil, "pbr_deferred_shade: null G-buffer plane");
    PBR_REQUIRE(ctx, tile->w >= 1 && tile->h >= 1 && tile->w <= 65535 && tile->h <= 65535, "pbr_deferred_shade: bad tile size");
    PBR_REQUIRE(ctx, tile->x0 + tile->w <= tile->full_w && tile->y0 + tile->h <= tile->full_h, "pbr_deferred_shade: tile outside frame");
    PBR_REQUIRE(ctx, gb->pitch >= tile->w && hdr_pitch >= tile->w, "pbr_deferred_shade: pitch < width");
    // the kernel addresses every plane with 32-bit byte offsets (16 B per pixel for the fp32 probe's output)
    PBR_REQUIRE(ctx, (uint64_t)gb->pitch * tile->h * 4u < (1ull << 32) && (uint64_t)hdr_pitch * tile->h * 16u < (1ull << 32),
                "pbr_deferred_shade: tile too large for 32-bit plane offsets (pitch x rows x 16 bytes must stay below 4 GiB)");
    PBR_REQUIRE(ctx, lut_res >= 1 && env_size >= 1 && env_mips >= 1 && env_mips <= 16 && (env_size >> (env_mips - 1)) >= 1, "pbr_deferred_shade: bad LUT/env size");
    PBR_REQUIRE(ctx, lut_res >= 1 && lut_res <= 16384 && (uint64_t)pbr_env_padded_texels(env_size, env_mips) * 8u < (1ull << 32),
                "pbr_deferred_shade: LUT larger than 16384^2 or padded env chain of 4 GiB or more (32-bit texture offsets)");
    PBR_REQUIRE(ctx, ((uintptr_t)env & 7u) == 0 && ((uintptr_t)lut & 3u) == 0, "pbr_deferred_shade: env must be 8-byte and lut 4-byte aligned");
    PBR_REQUIRE(ctx, g->Near > 0.0f && g->Far > g->Near, "pbr_deferred_shade: need 0 < Near < Far");
    PBR_REQUIRE(ctx, num_lights >= 0 && num_lights <= PBR_MAX_SCENE_LIGHTS, "pbr_deferred_shade: light count out of [0, 1024]");
    PBR_REQUIRE(ctx, num_lights == 0 || lights != nullptr, "pbr_deferred_shade: null lights");
    ShadeParams p;
    p.sh = g->SkyBoxSH;
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) p.InvView[r * 3 + c] = g->InvView[r * 4 + c];
    for (int i = 0; i < 3; i++) p.CameraPos[i] = g->CameraPos[i];
    p.Near = g->Near; p.Far = g->Far;
    p.near_height = 2.0f * g->Near * tanf(g->Fov / 2.0f);
    p.near_width = p.near_height * g->Ratio;
    p.log_far_near = logf(g->Far / g->Near);
    p.inv_near = (float)(1.0 / (double)g->Near);
    p.slice_k = (float)((double)PBR_CLUSTER_Z / log2((double)g->Far / (double)g->Near));
    p.x0 = tile->x0; p.y0 = tile->y0; p.w = tile->w; p.h = tile->h; p.full_w = tile->full_w; p.full_h = tile->full_h;
    p.A = gb->A; p.B = gb->B; p.C = gb->C; p.depth = gb->depth; p.stencil = gb->stencil; p.pitch = gb->pitch;
    p.lut = lut; p.lut_res = lut_res; p.env = env; p.env_size = env_size; p.env_mips = env_mips;
    for (uint32_t m = 0; m < 16; m++) p.env_mip_off[m] = (uint32_t)env_padded_mip_offset(env_size, m < env_mips ? m : env_mips - 1);
    p.clusters = clusters; p.lights = lights; p.hdr = hdr; p.hdr_pitch = hdr_pitch; p.hdr_f32 = hdr_f32;
    // schedule (see the kernel): long blocks first, short ones for the tail
    static const float big_frac = pbr::knob_float("PBR_SHADE_BIGFRAC", 0.92f);   // re-swept after the per-pixel trims (the knobs build): 0.9-0.95 with 1-row tail blocks beats 0.85 / 2 by ~0.4 %
    static const uint32_t rows_small_cfg = (uint32_t)pbr::knob_int("PBR_SHADE_ROWS_SMALL", 1);
    static const uint32_t rows_big_cfg = (uint32_t)pbr::knob_int("PBR_SHADE_ROWS_BIG", 0);   // 0: by target size (below)
    const uint32_t rows_small = rows_small_cfg >= 1 && rows_small_cfg <= (uint32_t)SHADE_ROWS ? rows_small_cfg : 1u;
    const uint32_t whole[1][4] = {{0, 0, tile->w, tile->h}};
    if (!rects) { rects = whole; n_rects = 1; }
    PBR_REQUIRE(ctx, n_rects >= 1 && n_rects <= (uint32_t)SHADE_MAX_RECTS, "pbr_deferred_shade: 1 .. 5 rectangles");
    // Rows of a long block.  SHADE_ROWS (8) wherever that gives at least ~1.3 generations of the blocks the device holds at once
    // (compute units x 5): the tables a block stages are amortised best, and 4K (3.2 generations), a cfg5 rank's tile (1.7) and 8K
    // measure best there.  Smaller targets get the row count that makes ~1.6 generations — one generation and a bit (1.1) is the worst
    // place to be: the launch then lasts two block lifetimes for one block's worth of work per slot.  Measured (profiles/r06_j_rows_*,
    // r06_g_*): 1440x960 (the reference's own target, 0.56 generations at 8 rows) 3 rows -10 %; 1920x1080 (0.84) 4 rows -0.5 ... -2.4 %.
    uint32_t rows_big = (uint32_t)SHADE_ROWS;
    {
        uint64_t row_segments = 0;   // 256-pixel row pieces of the launch
        for (uint32_t r = 0; r < n_rects; r++) row_segments += (uint64_t)((rects[r][2] + SHADE_BLOCK - 1) / SHADE_BLOCK) * rects[r][3];
        const uint64_t slots = (uint64_t)ctx->cu_count * SHADE_MIN_WAVES;
        if (row_segments * 10 < slots * 13 * SHADE_ROWS) {
            const uint32_t rws = (uint32_t)((row_segments * 10 + slots * 8) / (slots * 16));   // round(row_segments / (1.6 slots))
            rows_big = rws < 2 ? 2u : (rws > (uint32_t)SHADE_ROWS ? (uint32_t)SHADE_ROWS : rws);
        }
        if (rows_big_cfg >= 1 && rows_big_cfg <= (uint32_t)SHADE_ROWS) rows_big = rows_big_cfg;
    }
    ShadeRects rc{};
    rc.n = n_rects; rc.rows_big = rows_big; rc.rows_small = rows_small < rows_big ? rows_small : rows_big;
    uint32_t blocks = 0;
    for (uint32_t r = 0; r < n_rects; r++) {
        const uint32_t* q = rects[r];
        PBR_REQUIRE(ctx, q[2] >= 1 && q[3] >= 1 && q[0] + q[2] <= tile->w && q[1] + q[3] <= tile->h, "pbr_deferred_shade: rectangle outside the tile");
        rc.x0[r] = q[0]; rc.y0[r] = q[1]; rc.w[r] = q[2]; rc.h[r] = q[3];
        rc.cols[r] = (q[2] + SHADE_BLOCK - 1) / SHADE_BLOCK;
        rc.nb_big[r] = (uint32_t)((float)(q[3] / rows_big) * fminf(fmaxf(big_frac, 0.0f), 1.0f));
        const uint32_t rest = q[3] - rc.nb_big[r] * rows_big;
        rc.first[r] = blocks;
        blocks += rc.cols[r] * (rc.nb_big[r] + (rest + rc.rows_small - 1) / rc.rows_small);
    }
    rc.first[n_rects] = blocks;
    dim3 grid(blocks);
    // A block covers 256 x 8 pixels.  It can stage its cluster lists when that rectangle spans at most
    // MAX_STAGED_TILES cluster tiles: a tile is full_w/24 x full_h/16 pixels, +1 per axis for straddling.
    const uint32_t span_x = (uint32_t)((uint64_t)(SHADE_BLOCK - 1) * PBR_CLUSTER_X / tile->full_w) + 2;
    const uint32_t span_y = (uint32_t)((uint64_t)(SHADE_ROWS - 1) * PBR_CLUSTER_Y / tile->full_h) + 2;
    const int lstride = num_lights <= 256 ? 257 : PBR_MAX_SCENE_LIGHTS + 1;   // odd strides: no ds_read2 merging of two planes of one light, conflict-free planes
    const size_t plane_bytes = (size_t)((LIGHT_PLANES * lstride + 1) & ~1) * sizeof(float);
    // staged lists must also fit the 64 KiB a block may ask for (1 024 lights: 36 KiB of planes leave room for 8 tiles)
    const bool staged = span_x * span_y <= (uint32_t)MAX_STAGED_TILES &&   // (no lights at all: every list is one null pair)
                        plane_bytes + (size_t)span_x * span_y * PBR_CLUSTER_Z * LIST_STRIDE * sizeof(uint32_t) <= 65536;
    const int max_clusters = staged ? (int)(span_x * span_y) * PBR_CLUSTER_Z : 0;
    // (PBR_SHADE_LDS_PAD, knobs build: KiB of LDS a block asks for beyond its tables — 18 caps a compute unit at four 96-register blocks and leaves
    //  128 registers per SIMD + 28 KiB of LDS free: the hole of the round-6 co-residency experiment, EXPERIMENTS.md)
    static const size_t lds_pad = (size_t)pbr::knob_int("PBR_SHADE_LDS_PAD", 0) * 1024;
    const size_t lds = plane_bytes + (size_t)max_clusters * LIST_STRIDE * sizeof(uint32_t) + (lds_pad <= 40960 ? lds_pad : 0);
    const dim3 blk(SHADE_BLOCK);
    if (staged && lstride == 257) hipLaunchKernelGGL((k_deferred_shade<true, 257, F32OUT>), grid, blk, lds, ctx->stream, p, num_lights, max_clusters, rc);
    else if (staged) hipLaunchKernelGGL((k_deferred_shade<true, PBR_MAX_SCENE_LIGHTS + 1, F32OUT>), grid, blk, lds, ctx->stream, p, num_lights, max_clusters, rc);
    else if (lstride == 257) hipLaunchKernelGGL((k_deferred_shade<false, 257, F32OUT>), grid, blk, lds, ctx->stream, p, num_lights, 0, rc);
    else hipLaunchKernelGGL((k_deferred_shade<false, PBR_MAX_SCENE_LIGHTS + 1, F32OUT>), grid, blk, lds, ctx->stream, p, num_lights, 0, rc);
    return launched(ctx, "k_deferred_shade");
}

extern "C" {

pbr_status pbr_deferred_shade(pbr_ctx* ctx, const pbr_global* g, const pbr_tile* tile, const pbr_gbuffer* gb,
                              const pbr_half* lut, uint32_t lut_res,
                              const pbr_half* env, uint32_t env_size, uint32_t env_mips,
                              const pbr_cluster* clusters, const pbr_light* lights, int num_lights,
                              pbr_half* hdr, uint32_t hdr_pitch) {
    return shade_launch<false>(ctx, g, tile, gb, lut, lut_res, env, env_size, env_mips, clusters, lights, num_lights, hdr, nullptr, hdr_pitch);
}

// The same pass on up to 5 rectangles of the tile (tile-local {x, y, w, h}) in ONE launch; pixels outside are left
// untouched.  Multi-GPU overlap: the tile's border ring first, its core while the ring's bloom strips travel.
pbr_status pbr_deferred_shade_rects(pbr_ctx* ctx, const pbr_global* g, const pbr_tile* tile, const pbr_gbuffer* gb,
                                    const pbr_half* lut, uint32_t lut_res,
                                    const pbr_half* env, uint32_t env_size, uint32_t env_mips,
                                    const pbr_cluster* clusters, const pbr_light* lights, int num_lights,
                                    pbr_half* hdr, uint32_t hdr_pitch, const uint32_t (*rects)[4], uint32_t n_rects) {
    if (ctx && !rects) return pbr::fail(ctx, PBR_ERR_INVALID, "pbr_deferred_shade_rects: null rectangle list");
    return shade_launch<false>(ctx, g, tile, gb, lut, lut_res, env, env_size, env_mips, clusters, lights, num_lights, hdr, nullptr, hdr_pitch, rects, n_rects);
}

// Parity probe: the same kernel body, storing float4 instead of rounding to the R16G16B16A16_FLOAT target — what the
// <= 1e-4 relative L-inf bound of the shaded buffer is stated on.  Not a product path.
pbr_status pbr_deferred_shade_f32(pbr_ctx* ctx, const pbr_global* g, const pbr_tile* tile, const pbr_gbuffer* gb,
                                  const pbr_half* lut, uint32_t lut_res,
                                  const pbr_half* env, uint32_t env_size, uint32_t env_mips,
                                  const pbr_cluster* clusters, const pbr_light* lights, int num_lights,
                                  float* hdr_f32, uint32_t hdr_pitch) {
    if (ctx && hdr_f32 && ((uintptr_t)hdr_f32 & 15u) != 0) return pbr::fail(ctx, PBR_ERR_INVALID, "pbr_deferred_shade_f32: output must be 16-byte aligned");
    return shade_launch<true>(ctx, g, tile, gb, lut, lut_res, env, env_size, env_mips, clusters, lights, num_lights, nullptr, hdr_f32, hdr_pitch);
}

}  // extern "C"
