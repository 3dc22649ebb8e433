/*
 * pbr_oracle_f64.cpp — the deferred shade evaluated in DOUBLE precision: the value the
 * reference's formulas have in exact arithmetic, to ~1e-15.  TEST INFRASTRUCTURE ONLY (see pbr_oracle.h).
 *
 * Why it exists.  The parity bound on the shaded HDR buffer is "<= 1e-4 relative L-inf" and both sides of that
 * comparison — the GPU kernel and the fp32 oracle (pbr_oracle.cpp) — are fp32 evaluations of a formula that is
 * ill-conditioned at GGX highlights (distribution_ggx: t = NdotH^2 (a^4 - 1) + 1 cancels down to a^4).  Comparing the two
 * with each other cannot tell which of them is off.  This file is the third party: every CONTINUOUS operation of
 * brdf.hlsli:6-67 and deferred_shading.hlsl:23-192 in double, from the same fp32 / fp16 / UNORM8 inputs.
 *
 * Discontinuous steps.  The path has four kinds of step functions: the x.8 fixed-point snap of filter coordinates (the
 * sampler model of pbr_oracle.h), the cube face choice, the cluster z-slice (an int() of a log), the octahedral fold.
 * An fp32 evaluation whose argument lies within a few ulp of a step edge lands on either side, legitimately.  So the
 * result here is an INTERVAL per channel [lo, hi]: where the exact argument of a snap / face choice lies within
 * `SNAP_TOL_ULPS` fp32 ulps of an edge, both sides are evaluated and the interval covers them (sums and products of
 * positive terms, so interval arithmetic is exact); elsewhere lo == hi.  A pixel whose cluster slice or octahedral fold is
 * that close to its edge is FLAGGED instead (its light list / normal could be another one altogether) and tests skip it;
 * they are a ~1e-5 share of the pixels.  The distance of an fp32 colour c to the truth is max(lo - c, c - hi, 0).
 *
 * Inputs are taken as the fp32 numbers the shader receives: matrices, camera, light records, UNORM8 / 255 decoded to the
 * NEAREST FLOAT (the fixed-function conversion happens before the shader), half texels.  tan(Fov / 2), pow, log, sqrt are
 * libm double.  Two decisions are copied from fp32 because every fp32 evaluation takes them identically (one correctly
 * rounded operation on identical inputs): the LUT's roughness coordinate (roughness * res, exact) and the env LOD
 * (roughness * 5, one multiply).
 */
#include "pbr_oracle.h"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace {

constexpr double PI_D = 3.14159265359;        // global.hlsli:4 (the shader's literal, not pi)
constexpr double INV_PI_D = 0.31830988618;    // global.hlsli:5
constexpr double EPS_D = 1e-6;
constexpr double SNAP_TOL_ULPS = 16.0;        // an fp32 coordinate within this many ulps of a step edge may land on either side

struct D3 { double x, y, z; };
static inline D3 d3(double x, double y, double z) { return D3{x, y, z}; }
static inline D3 operator+(D3 a, D3 b) { return d3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline D3 operator-(D3 a, D3 b) { return d3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline D3 operator*(D3 a, D3 b) { return d3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline D3 operator*(D3 a, double s) { return d3(a.x * s, a.y * s, a.z * s); }
static inline double dot(D3 a, D3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline D3 normalize(D3 v) { double l = std::sqrt(dot(v, v)); return d3(v.x / l, v.y / l, v.z / l); }
static inline D3 dmin(D3 a, D3 b) { return d3(std::min(a.x, b.x), std::min(a.y, b.y), std::min(a.z, b.z)); }
static inline D3 dmax(D3 a, D3 b) { return d3(std::max(a.x, b.x), std::max(a.y, b.y), std::max(a.z, b.z)); }

static inline double half_to_double(uint16_t h) { return (double)orc_f16_to_f32(h); }

// x.8 snap of the scaled coordinate c = u * size (pbr_oracle.cpp bilinear_coord): the snapped positions an fp32
// evaluation can produce — the exact one, and its neighbour when c * 256 + 0.5 is within tol of an integer.
struct SnapSet { int n; double x[2]; };   // positions in texel units, half-texel offset already removed
static inline SnapSet snap_set(double c, int size) {
    const double lim = (double)size + 1.5;
    c = c > lim ? lim : (c < -1.5 ? -1.5 : c);
    const double s = c * 256.0 + 0.5, fl = std::floor(s);
    const double tol = SNAP_TOL_ULPS * std::ldexp(std::max(std::fabs(c) * 256.0, 1.0), -24);
    SnapSet r;
    r.n = 1;
    r.x[0] = fl / 256.0 - 0.5;
    if (s - fl < tol) { r.x[1] = (fl - 1.0) / 256.0 - 0.5; r.n = 2; }
    else if (fl + 1.0 - s < tol) { r.x[1] = (fl + 1.0) / 256.0 - 0.5; r.n = 2; }
    return r;
}

struct Cube16 {   // a cube chain of half4 texels (the prefiltered env) or, with `f32` set, of float4 texels (a source sky)
    const uint16_t* data; uint32_t size; const float* f32 = nullptr;
    static size_t mip_offset(uint32_t size, uint32_t mip) {
        size_t off = 0;
        for (uint32_t m = 0; m < mip; m++) { size_t s = size >> m; off += 6 * s * s; }
        return off;
    }
    D3 texel(uint32_t mip, uint32_t face, int x, int y) const {
        const int s = (int)(size >> mip);
        const size_t at = 4 * (mip_offset(size, mip) + ((size_t)face * s + y) * s + x);
        if (f32) return d3((double)f32[at], (double)f32[at + 1], (double)f32[at + 2]);
        const uint16_t* p = data + at;
        return d3(half_to_double(p[0]), half_to_double(p[1]), half_to_double(p[2]));
    }
};

static inline D3 cube_dir_raw(uint32_t face, double u, double v) {   // env_map_gen.hlsl:27-41
    switch (face) {
        case 0: return d3(1.0, -v, -u);
        case 1: return d3(-1.0, -v, u);
        case 2: return d3(u, 1.0, v);
        case 3: return d3(u, -1.0, -v);
        case 4: return d3(u, -v, 1.0);
        default: return d3(-u, -v, -1.0);
    }
}
// projection onto the face of a given major axis (0 x, 1 y, 2 z); u, v in [0, 1]
static inline void face_uv_axis(D3 d, int axis, uint32_t& face, double& u, double& v) {
    double sc, tc, ma;
    if (axis == 0) { ma = std::fabs(d.x); if (d.x >= 0) { face = 0; sc = -d.z; tc = -d.y; } else { face = 1; sc = d.z; tc = -d.y; } }
    else if (axis == 1) { ma = std::fabs(d.y); if (d.y >= 0) { face = 2; sc = d.x; tc = d.z; } else { face = 3; sc = d.x; tc = -d.z; } }
    else { ma = std::fabs(d.z); if (d.z >= 0) { face = 4; sc = d.x; tc = -d.y; } else { face = 5; sc = -d.x; tc = -d.y; } }
    u = (sc / ma + 1.0) * 0.5;
    v = (tc / ma + 1.0) * 0.5;
}
static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// seamless texel fetch: the model's rule (pbr_oracle.cpp cube_fetch_seamless) — integer decisions, evaluated in double on
// exactly representable arguments (texel centres), so it agrees with the fp32 rule everywhere
static D3 fetch_seamless(const Cube16& c, uint32_t mip, uint32_t face, int x, int y) {
    const int s = (int)(c.size >> mip);
    if (x >= 0 && x < s && y >= 0 && y < s) return c.texel(mip, face, x, y);
    if ((x < 0 || x >= s) && (y < 0 || y >= s)) y = clampi(y, 0, s - 1);
    const double uu = 2.0 * ((double)x + 0.5) / (double)s - 1.0, vv = 2.0 * ((double)y + 0.5) / (double)s - 1.0;
    const D3 d = cube_dir_raw(face, uu, vv);
    const double ax = std::fabs(d.x), ay = std::fabs(d.y), az = std::fabs(d.z);
    const int axis = (ax >= ay && ax >= az) ? 0 : (ay >= az ? 1 : 2);
    uint32_t f2; double u2, v2;
    face_uv_axis(d, axis, f2, u2, v2);
    return c.texel(mip, f2, clampi((int)std::floor(u2 * s), 0, s - 1), clampi((int)std::floor(v2 * s), 0, s - 1));
}

static inline D3 lerp3(D3 a, D3 b, double f) { return f == 0.0 ? a : a * (1.0 - f) + b * f; }
static D3 bilinear_at(const Cube16& c, uint32_t mip, uint32_t face, double x, double y) {
    const double fx = std::floor(x), fy = std::floor(y);
    const int x0 = (int)fx, y0 = (int)fy;
    const double wx = x - fx, wy = y - fy;
    const D3 c00 = fetch_seamless(c, mip, face, x0, y0), c10 = fetch_seamless(c, mip, face, x0 + 1, y0);
    const D3 c01 = fetch_seamless(c, mip, face, x0, y0 + 1), c11 = fetch_seamless(c, mip, face, x0 + 1, y0 + 1);
    return lerp3(lerp3(c00, c10, wx), lerp3(c01, c11, wx), wy);
}
// [lo, hi] of the bilinear sample of one mip at (face, u, v) over the admissible snaps
static void level_interval(const Cube16& c, uint32_t mip, uint32_t face, double u, double v, D3& lo, D3& hi) {
    const int s = (int)(c.size >> mip);
    const SnapSet xs = snap_set(u * s, s), ys = snap_set(v * s, s);
    bool first = true;
    for (int i = 0; i < xs.n; i++)
        for (int j = 0; j < ys.n; j++) {
            const D3 t = bilinear_at(c, mip, face, xs.x[i], ys.x[j]);
            if (first) { lo = hi = t; first = false; } else { lo = dmin(lo, t); hi = dmax(hi, t); }
        }
}

// [lo, hi] of the trilinear cube sample along R at the (already snapped) LOD: over the admissible faces (the major axis and any
// axis within rounding of it) and the admissible x.8 snaps of the two levels' coordinates
static void trilinear_interval(const Cube16& cube, uint32_t mips, D3 R, double lod_s, D3& out_lo, D3& out_hi) {
    const uint32_t l0 = (uint32_t)std::floor(lod_s), l1 = l0 + 1 < mips ? l0 + 1 : mips - 1;
    const double lf = lod_s - (double)l0;
    const double ab[3] = {std::fabs(R.x), std::fabs(R.y), std::fabs(R.z)};
    const int major = (ab[0] >= ab[1] && ab[0] >= ab[2]) ? 0 : (ab[1] >= ab[2] ? 1 : 2);
    bool first = true;
    for (int ax = 0; ax < 3; ax++) {
        if (ax != major && ab[ax] < ab[major] * (1.0 - 1e-6)) continue;
        uint32_t face; double cu, cv;
        face_uv_axis(R, ax, face, cu, cv);
        D3 lo0, hi0, lo, hi;
        level_interval(cube, l0, face, cu, cv, lo0, hi0);
        lo = lo0; hi = hi0;
        if (lf != 0.0 && l1 != l0) {
            D3 lo1, hi1;
            level_interval(cube, l1, face, cu, cv, lo1, hi1);
            lo = lo0 * (1.0 - lf) + lo1 * lf;
            hi = hi0 * (1.0 - lf) + hi1 * lf;
        }
        if (first) { out_lo = lo; out_hi = hi; first = false; } else { out_lo = dmin(out_lo, lo); out_hi = dmax(out_hi, hi); }
    }
}

}  // namespace

extern "C" int orc_deferred_shade_f64(const pbr_global* g, const pbr_tile* tile, const pbr_gbuffer* gb,
                                      const uint16_t* lut, uint32_t lut_res,
                                      const uint16_t* env, uint32_t env_size, uint32_t env_mips,
                                      const pbr_cluster* clusters, const pbr_light* lights,
                                      double* lo_rgb, double* hi_rgb, uint8_t* flags, uint32_t out_pitch) {
    if (!g || !tile || !gb || !lut || !env || !clusters || !lo_rgb || !hi_rgb || !flags) return PBR_ERR_INVALID;
    const Cube16 cube{env, env_size};
    const double Near = g->Near, Far = g->Far;
    const double near_height = 2.0 * Near * std::tan((double)g->Fov / 2.0), near_width = near_height * (double)g->Ratio;
    const D3 cam = d3(g->CameraPos[0], g->CameraPos[1], g->CameraPos[2]);
    const float* M = g->InvView;
#pragma omp parallel for schedule(dynamic, 4)
    for (int64_t py = 0; py < (int64_t)tile->h; py++) {
        for (uint32_t px = 0; px < tile->w; px++) {
            const size_t gi = (size_t)py * gb->pitch + px, oi = (size_t)py * out_pitch + px;
            flags[oi] = 0;
            for (int k = 0; k < 3; k++) lo_rgb[3 * oi + k] = hi_rgb[3 * oi + k] = 0.0;
            if (gb->stencil[gi] == 0) { flags[oi] = 1; continue; }   // not shaded
            const double u = ((double)(tile->x0 + px) + 0.5) / (double)tile->full_w;
            const double v = ((double)(tile->y0 + (uint32_t)py) + 0.5) / (double)tile->full_h;
            const double ndc_x = 2.0 * u - 1.0, ndc_y = 1.0 - 2.0 * v;
            const D3 cvv = d3(ndc_x * 0.5 * near_width, ndc_y * 0.5 * near_height, Near);
            const D3 camera_vec = d3(M[0] * cvv.x + M[1] * cvv.y + M[2] * cvv.z, M[4] * cvv.x + M[5] * cvv.y + M[6] * cvv.z,
                                     M[8] * cvv.x + M[9] * cvv.y + M[10] * cvv.z);
            const uint32_t a = gb->A[gi], b = gb->B[gi], c = gb->C[gi];
            auto unorm = [](uint32_t k) { return (double)((float)k / 255.0f); };   // fixed-function decode: nearest float
            const D3 albedo = d3(unorm(a & 255u), unorm((a >> 8) & 255u), unorm((a >> 16) & 255u));
            const double emission = unorm(a >> 24), roughness = unorm(c & 255u), metallic = unorm((c >> 8) & 255u);
            // octahedral decode (global.hlsli:130-138), custom sign(0) = +1
            D3 n = d3(unorm(b & 255u) * 2.0 - 1.0, unorm((b >> 8) & 255u) * 2.0 - 1.0, 0.0);
            n.z = 1.0 - std::fabs(n.x) - std::fabs(n.y);
            if (std::fabs(n.z) < 1e-6) flags[oi] |= 2;                             // fold decided by rounding
            if (n.z < 0.0) {
                const double nx = (n.x < 0.0 ? -1.0 : 1.0) * (1.0 - std::fabs(n.y)), ny = (n.y < 0.0 ? -1.0 : 1.0) * (1.0 - std::fabs(n.x));
                n.x = nx; n.y = ny;
            }
            n = normalize(n);
            const double depth = gb->depth[gi];
            const double z_vs = Near * Far / (Far - depth * (Far - Near));
            const D3 pos = cam + camera_vec * (z_vs / Near);
            const D3 view = normalize(cam - pos);

            // EnvironmentDiffuse, deferred_shading.hlsl:23-54
            const pbr_sh_pack& sh = g->SkyBoxSH;
            const double A4[4] = {n.x, n.y, n.z, 1.0}, B4[4] = {n.x * n.y, n.y * n.z, n.z * n.z, n.z * n.x}, C1 = n.x * n.x - n.y * n.y;
            auto dot4 = [](const float* p, const double* q) { return p[0] * q[0] + p[1] * q[1] + p[2] * q[2] + p[3] * q[3]; };
            const D3 irr = d3(dot4(sh.sha_r, A4) + dot4(sh.shb_r, B4) + sh.shc[0] * C1, dot4(sh.sha_g, A4) + dot4(sh.shb_g, B4) + sh.shc[1] * C1,
                              dot4(sh.sha_b, A4) + dot4(sh.shb_b, B4) + sh.shc[2] * C1);
            const D3 env_diffuse = (albedo * (1.0 - metallic)) * INV_PI_D * irr;

            // EnvironmentSpecular, :56-70
            const D3 F0 = d3(0.04 + metallic * (albedo.x - 0.04), 0.04 + metallic * (albedo.y - 0.04), 0.04 + metallic * (albedo.z - 0.04));
            const double NdV = dot(n, view), NdotV = std::max(NdV, 0.0);
            const D3 R = normalize(n * (2.0 * NdV) - view);
            // LOD: roughness * 5 is ONE fp32 multiply on identical inputs in every fp32 evaluation: taken from there
            float lodf = (float)roughness * (float)PBR_ENV_MIPS;
            const float maxl = (float)(env_mips - 1);
            lodf = lodf < 0.0f ? 0.0f : (lodf > maxl ? maxl : lodf);
            lodf = std::floor(lodf * 256.0f + 0.5f) * (1.0f / 256.0f);
            const uint32_t l0 = (uint32_t)std::floor(lodf), l1 = l0 + 1 < env_mips ? l0 + 1 : env_mips - 1;
            const double lf = (double)lodf - (double)l0;
            // face: the major axis, and any other axis within rounding of it (both faces are admissible)
            const double ab[3] = {std::fabs(R.x), std::fabs(R.y), std::fabs(R.z)};
            const int major = (ab[0] >= ab[1] && ab[0] >= ab[2]) ? 0 : (ab[1] >= ab[2] ? 1 : 2);
            D3 env_lo{}, env_hi{};
            bool first = true;
            for (int ax = 0; ax < 3; ax++) {
                if (ax != major && ab[ax] < ab[major] * (1.0 - 1e-6)) continue;
                uint32_t face; double cu, cv;
                face_uv_axis(R, ax, face, cu, cv);
                D3 lo0, hi0, lo, hi;
                level_interval(cube, l0, face, cu, cv, lo0, hi0);
                lo = lo0; hi = hi0;
                if (lf != 0.0 && l1 != l0) {
                    D3 lo1, hi1;
                    level_interval(cube, l1, face, cu, cv, lo1, hi1);
                    lo = lo0 * (1.0 - lf) + lo1 * lf;
                    hi = hi0 * (1.0 - lf) + hi1 * lf;
                }
                if (first) { env_lo = lo; env_hi = hi; first = false; } else { env_lo = dmin(env_lo, lo); env_hi = dmax(env_hi, hi); }
            }
            // LUT (half2, bilinear clamp): x from roughness * res (exact in fp32 for a power-of-two res; snapped identically by
            // everyone), y from NdotV * res over its admissible snaps
            const int res = (int)lut_res;
            const double cxs = std::floor(roughness * res * 256.0 + 0.5) / 256.0 - 0.5;
            const double fxl = std::floor(cxs), wx = cxs - fxl;
            const int x0 = clampi((int)fxl, 0, res - 1), x1 = clampi((int)fxl + 1, 0, res - 1);
            auto lutat = [&](int xx, int yy, int ch) { return half_to_double(lut[2 * ((size_t)yy * lut_res + xx) + ch]); };
            const SnapSet ys = snap_set(NdotV * res, res);
            double lut_lo[2] = {0, 0}, lut_hi[2] = {0, 0};
            for (int j = 0; j < ys.n; j++) {
                const double fy = std::floor(ys.x[j]), wy = ys.x[j] - fy;
                const int y0 = clampi((int)fy, 0, res - 1), y1 = clampi((int)fy + 1, 0, res - 1);
                for (int ch = 0; ch < 2; ch++) {
                    const double r0 = wx == 0.0 ? lutat(x0, y0, ch) : lutat(x0, y0, ch) * (1.0 - wx) + lutat(x1, y0, ch) * wx;
                    const double r1 = wx == 0.0 ? lutat(x0, y1, ch) : lutat(x0, y1, ch) * (1.0 - wx) + lutat(x1, y1, ch) * wx;
                    const double val = wy == 0.0 ? r0 : r0 * (1.0 - wy) + r1 * wy;
                    if (j == 0) lut_lo[ch] = lut_hi[ch] = val;
                    else { lut_lo[ch] = std::min(lut_lo[ch], val); lut_hi[ch] = std::max(lut_hi[ch], val); }
                }
            }
            // env texels and LUT values are >= 0 (radiance, split-sum integrals): the product's interval is [lo*lo, hi*hi]
            const D3 w_lo = F0 * lut_lo[0] + d3(lut_lo[1], lut_lo[1], lut_lo[1]), w_hi = F0 * lut_hi[0] + d3(lut_hi[1], lut_hi[1], lut_hi[1]);
            const D3 spec_lo = dmin(env_lo * w_lo, env_hi * w_hi), spec_hi = dmax(env_lo * w_lo, env_hi * w_hi);

            // cluster: the fp32 decision (clustered.hlsli:45-60 as the oracle evaluates it); flagged when the slice
            // coordinate is within 1e-4 of an integer — another evaluation may walk another light list there
            const float uf = ((float)(tile->x0 + px) + 0.5f) / (float)tile->full_w, vf = ((float)(tile->y0 + (uint32_t)py) + 0.5f) / (float)tile->full_h;
            const float zf = g->Near * g->Far / (g->Far - gb->depth[gi] * (g->Far - g->Near));
            const int ci = orc_cluster_index(g, uf, vf, zf);
            const double zc = std::min(std::max(z_vs, Near), Far);
            const double slice = (double)PBR_CLUSTER_Z * std::log(zc / Near) / std::log(Far / Near);
            if (std::fabs(slice - std::round(slice)) < 1e-4 && slice > 0.5 && slice < (double)PBR_CLUSTER_Z - 0.5) flags[oi] |= 4;
            const double sxc = u * PBR_CLUSTER_X, syc = (1.0 - v) * PBR_CLUSTER_Y;
            if (std::fabs(sxc - std::round(sxc)) < 1e-6 || std::fabs(syc - std::round(syc)) < 1e-6) flags[oi] |= 4;
            const pbr_cluster& cl = clusters[ci];
            D3 pl = d3(0, 0, 0);
            for (int i = 0; i < cl.NumLights; i++) {   // deferred_shading.hlsl:159-186 + brdf.hlsli:47-67
                const pbr_light& lt = lights[cl.LightIndex[i]];
                D3 dir = d3(lt.Position[0], lt.Position[1], lt.Position[2]) - pos;
                const double dist = std::sqrt(dot(dir, dir));
                dir = dir * (1.0 / dist);
                const double NdotL = std::max(dot(n, dir), 0.0);
                const double att = 1.0 / std::max((double)lt.C0 + (double)lt.C1 * dist + (double)lt.C2 * dist * dist, EPS_D);
                const D3 H = normalize(dir + view);
                const double NdotH = std::max(dot(n, H), 0.0);
                const double p5 = std::pow(std::max(1.0 - NdotL, EPS_D), 5.0);
                const D3 F = F0 + (d3(1, 1, 1) - F0) * p5;
                const double aa = roughness * roughness, a4 = aa * aa;
                const double t = NdotH * NdotH * (a4 - 1.0) + 1.0;
                const double D = a4 / std::max(PI_D * t * t, EPS_D);
                const double k = (roughness + 1.0) * (roughness + 1.0) / 8.0;
                const double G = (NdotV / std::max(NdotV * (1.0 - k) + k, EPS_D)) * (NdotL / std::max(NdotL * (1.0 - k) + k, EPS_D));
                const D3 Kd = (d3(1, 1, 1) - F) * (1.0 - metallic);
                const double denom = std::max(4.0 * NdotL * NdotV, 0.0001);
                const D3 f = Kd * albedo * INV_PI_D + F * (D * G / denom);
                pl = pl + f * d3(lt.Color[0], lt.Color[1], lt.Color[2]) * ((double)lt.Intensity * att * NdotL);
            }
            const D3 rest = env_diffuse + pl + albedo * emission;
            const D3 lo = rest + spec_lo, hi = rest + spec_hi;
            lo_rgb[3 * oi] = lo.x; lo_rgb[3 * oi + 1] = lo.y; lo_rgb[3 * oi + 2] = lo.z;
            hi_rgb[3 * oi] = hi.x; hi_rgb[3 * oi + 1] = hi.y; hi_rgb[3 * oi + 2] = hi.z;
        }
    }
    return PBR_OK;
}

// ==================================================================== a3 in double: precompute_brdf.hlsl:20-62
// The split-sum LUT evaluated in double precision, same estimator (the 1 024 Hammersley / GGX-importance samples of
// brdf.hlsli:71-114, IntegrateBRDF's sum, k = roughness^2 / 2): what the reference's formulas give in exact arithmetic for the
// texel's (roughness, NdotV) = (x / (res - 1), (y + 1) / res).  No step functions on this path (the max(., eps) floors are
// continuous), so the result is a number, not an interval.  out_ab: rows * res pairs (A, B) of doubles.
// The third party between the GPU kernel — whose sample step is algebraically rearranged (no normalize(L), 1 / max(ab, eps) as
// a min of reciprocals, fused multiply-adds) — and the fp32 restatement of the shader's own order of operations.
extern "C" int orc_brdf_lut_f64(uint32_t res, uint32_t y0, uint32_t rows, double* out_ab) {
    if (!out_ab || res < 2 || y0 + rows > res) return PBR_ERR_INVALID;
    const double TWO_PI_D = 2.0 * PI_D;   // the shader's literal PI (global.hlsli:4) doubled, as brdf.hlsli:83 does
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t yy = 0; yy < (int64_t)rows; yy++) {
        const uint32_t y = y0 + (uint32_t)yy;
        for (uint32_t x = 0; x < res; x++) {
            const double roughness = (double)x / (double)(res - 1), NdotV = (double)(y + 1) / (double)res;
            const D3 V = d3(std::sqrt(1.0 - NdotV * NdotV), 0.0, NdotV);
            const double a = roughness * roughness, k = roughness * roughness / 2.0;
            double A = 0.0, B = 0.0;
            for (uint32_t i = 0; i < PBR_SAMPLE_COUNT; i++) {
                uint32_t bits = i;
                bits = (bits << 16u) | (bits >> 16u);
                bits = ((bits & 0x55555555u) << 1u) | ((bits & 0xAAAAAAAAu) >> 1u);
                bits = ((bits & 0x33333333u) << 2u) | ((bits & 0xCCCCCCCCu) >> 2u);
                bits = ((bits & 0x0F0F0F0Fu) << 4u) | ((bits & 0xF0F0F0F0u) >> 4u);
                bits = ((bits & 0x00FF00FFu) << 8u) | ((bits & 0xFF00FF00u) >> 8u);
                const double xi_x = (double)i / (double)PBR_SAMPLE_COUNT, xi_y = (double)bits * 2.3283064365386963e-10;
                const double phi = TWO_PI_D * xi_x;
                const double cos_theta = std::sqrt((1.0 - xi_y) / (1.0 + (a * a - 1.0) * xi_y));
                const double sin_theta = std::sqrt(std::max(1.0 - cos_theta * cos_theta, 0.0));
                // N = (0,0,1): tangent frame of ggx_important_sample = (up x ... ) -> H = normalize(T hx + B hy + N hz) with
                // up = (1,0,0) for |N.z| >= 0.999: T = normalize(N x up) = (0,1,0), B = N x T = (-1,0,0)
                const D3 H = normalize(d3(-(sin_theta * std::sin(phi)), sin_theta * std::cos(phi), cos_theta));
                const double VdH = dot(V, H);
                const D3 L = normalize(H * (2.0 * VdH) - V);
                const double NdotL = std::max(L.z, 0.0), NdotH = std::max(H.z, 0.0), VdotH = std::max(VdH, 0.0);
                if (NdotL > 0.0) {
                    const double Fc = std::pow(1.0 - VdotH, 5.0);
                    const double gv = NdotV / std::max(NdotV * (1.0 - k) + k, EPS_D);
                    const double gl = NdotL / std::max(NdotL * (1.0 - k) + k, EPS_D);
                    const double G_Vis = (gv * gl * VdotH) / std::max(NdotH * NdotV, 0.0001);
                    A += (1.0 - Fc) * G_Vis;
                    B += Fc * G_Vis;
                }
            }
            out_ab[2 * ((size_t)yy * res + x) + 0] = A / (double)PBR_SAMPLE_COUNT;
            out_ab[2 * ((size_t)yy * res + x) + 1] = B / (double)PBR_SAMPLE_COUNT;
        }
    }
    return PBR_OK;
}

// ==================================================================== a4 in double: env_map_gen.hlsl:50-105
// The GGX prefilter of chosen texels (index (face * s + y) * s + x of mip `mip`) in double precision: the estimator of
// env_map_gen.hlsl::cs_main on the fp32 source chain — N = V = R = the texel-CORNER direction (Q8), the 1 024 GGX-importance
// samples, N.L-weighted trilinear fetches at LOD = 0.5 log2(sample solid angle / texel solid angle) — with the path's step
// functions treated as in the shade: where the LOD lies within 8e-6 of an x.8 step (the fp32 chain pdf -> log2 carries ~1e-6),
// a filter coordinate within 16 ulps of a snap edge, or the direction within 1e-6 of a face edge (texel corners on cube edges
// are EXACT ties), every admissible side is evaluated and the result is an interval [lo, hi] per channel (radiance and the
// weights are >= 0: interval arithmetic is exact).  out_lo / out_hi: count x 3 doubles.
extern "C" int orc_prefilter_env_texels_f64(const float* sky, uint32_t sky_size, uint32_t sky_mips, uint32_t size, uint32_t mips,
                                            uint32_t mip, const uint32_t* texels, uint32_t count, double* out_lo, double* out_hi) {
    if (!sky || !out_lo || !out_hi || !texels || mips < 1 || mip >= mips || (size >> mip) == 0 || sky_mips < 1) return PBR_ERR_INVALID;
    Cube16 cube{nullptr, sky_size, sky};
    const uint32_t s = size >> mip;
    const double roughness = mips > 1 ? (double)mip / (double)(mips - 1) : 0.0;   // DeferredPipeline.cpp:99
    for (uint32_t k = 0; k < count; k++)
        if (texels[k] >= 6u * s * s) return PBR_ERR_INVALID;
    const double TWO_PI_D = 2.0 * PI_D, maxl = (double)(sky_mips - 1);
    const double texel_sa = 4.0 * PI_D / (double)(6u * size * size);   // base size for every mip (Q8)
#pragma omp parallel for schedule(dynamic, 8)
    for (int64_t k = 0; k < (int64_t)count; k++) {
        const uint32_t t = texels[k], face = t / (s * s), y = (t / s) % s, x = t % s;
        const double u = (double)x / (double)s, v = (double)y / (double)s;
        const D3 N = normalize(cube_dir_raw(face, 2.0 * u - 1.0, 2.0 * v - 1.0));
        const D3 up = std::fabs(N.z) < 0.999 ? d3(0, 0, 1) : d3(1, 0, 0);
        const D3 cr = d3(N.y * up.z - N.z * up.y, N.z * up.x - N.x * up.z, N.x * up.y - N.y * up.x);
        const D3 T = normalize(cr);
        const D3 Bt = d3(N.y * T.z - N.z * T.y, N.z * T.x - N.x * T.z, N.x * T.y - N.y * T.x);
        const double a = roughness * roughness, a4 = a * a;
        D3 sum_lo = d3(0, 0, 0), sum_hi = d3(0, 0, 0);
        double total_w = 0.0;
        for (uint32_t i = 0; i < PBR_SAMPLE_COUNT; i++) {
            uint32_t bits = i;
            bits = (bits << 16u) | (bits >> 16u);
            bits = ((bits & 0x55555555u) << 1u) | ((bits & 0xAAAAAAAAu) >> 1u);
            bits = ((bits & 0x33333333u) << 2u) | ((bits & 0xCCCCCCCCu) >> 2u);
            bits = ((bits & 0x0F0F0F0Fu) << 4u) | ((bits & 0xF0F0F0F0u) >> 4u);
            bits = ((bits & 0x00FF00FFu) << 8u) | ((bits & 0xFF00FF00u) >> 8u);
            const double xi_x = (double)i / (double)PBR_SAMPLE_COUNT, xi_y = (double)bits * 2.3283064365386963e-10;
            const double phi = TWO_PI_D * xi_x;
            const double cos_theta = std::sqrt((1.0 - xi_y) / (1.0 + (a4 - 1.0) * xi_y));
            const double sin_theta = std::sqrt(std::max(1.0 - cos_theta * cos_theta, 0.0));
            const double hx = sin_theta * std::cos(phi), hy = sin_theta * std::sin(phi), hz = cos_theta;
            const D3 H = normalize(T * hx + Bt * hy + N * hz);
            const double VdH = dot(N, H);   // V = N
            const D3 L = normalize(H * (2.0 * VdH) - N);
            const double NdotL = std::max(dot(N, L), 0.0);
            if (!(NdotL > 0.0)) continue;
            const double NdotH = std::max(VdH, 0.0), HdotV = NdotH;
            const double tt = NdotH * NdotH * (a4 - 1.0) + 1.0;
            const double D = a4 / std::max(PI_D * tt * tt, EPS_D);
            const double pdf = D * NdotH / (4.0 * HdotV + 0.0001);
            const double sample_sa = 1.0 / ((double)PBR_SAMPLE_COUNT * pdf + 0.0001);
            double lod = roughness == 0.0 ? 0.0 : 0.5 * std::log2(sample_sa / texel_sa);
            if (!(lod == lod)) lod = 0.0;
            lod = lod < 0.0 ? 0.0 : (lod > maxl ? maxl : lod);
            // x.8 snap of the LOD: the exact one and, within 8e-6 of a step, its neighbour
            const double sl = lod * 256.0 + 0.5, fl = std::floor(sl);
            double cand[2] = {fl / 256.0, 0.0};
            int nc = 1;
            if (sl - fl < 2e-3 && fl >= 1.0) cand[nc++] = (fl - 1.0) / 256.0;
            else if (fl + 1.0 - sl < 2e-3 && (fl + 1.0) / 256.0 <= maxl) cand[nc++] = (fl + 1.0) / 256.0;
            D3 lo{}, hi{};
            for (int c = 0; c < nc; c++) {
                D3 l{}, h{};
                trilinear_interval(cube, sky_mips, L, cand[c], l, h);
                if (c == 0) { lo = l; hi = h; } else { lo = dmin(lo, l); hi = dmax(hi, h); }
            }
            sum_lo = sum_lo + lo * NdotL; sum_hi = sum_hi + hi * NdotL;
            total_w += NdotL;
        }
        const D3 rl = sum_lo * (1.0 / total_w), rh = sum_hi * (1.0 / total_w);
        out_lo[3 * k] = rl.x; out_lo[3 * k + 1] = rl.y; out_lo[3 * k + 2] = rl.z;
        out_hi[3 * k] = rh.x; out_hi[3 * k + 1] = rh.y; out_hi[3 * k + 2] = rh.z;
    }
    return PBR_OK;
}

