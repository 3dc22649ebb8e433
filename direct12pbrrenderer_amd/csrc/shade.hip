// shade.hip — deferred Cook-Torrance shade of a G-buffer tile (deferred_shading.hlsl:91-192).
//
// Replaces the stencil-masked full-screen draw of DeferredShadingPass::Execute
// (DeferredPipeline.cpp:187-206).  One lane per pixel, a wave = 64 consecutive pixels of a row,
// so every G-buffer plane is read as one 256-byte (RGBA8 / depth) or 64-byte (stencil)
// coalesced segment per wave and the half4 output is one 512-byte store.  The light table is
// staged into LDS once per block (48-byte records, float4-aligned) so the per-light fetch in
// the divergent cluster loop is three ds_read_b128 instead of eleven scattered global loads.
#include "pbr_internal.hpp"
#include "pbr_device.hpp"

using namespace pbr;

struct ShadeParams {
    pbr_sh_pack sh;
    float InvView[9];      // 3x3 part, row-major
    float CameraPos[3];
    float Near, Far, Fov, Ratio;
    uint32_t x0, y0, w, h, full_w, full_h;
    const uint32_t* A;
    const uint32_t* B;
    const uint32_t* C;
    const float* depth;
    const uint8_t* stencil;
    uint32_t pitch;
    const pbr_half* lut;
    uint32_t lut_res;
    const pbr_half* env;
    uint32_t env_size, env_mips;
    const pbr_cluster* clusters;
    const pbr_light* lights;
    pbr_half* hdr;
    uint32_t hdr_pitch;
};

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

// LDS light record: 12 floats
struct LightLds { float4 pos_int; float4 col_c0; float4 c1c2; };

constexpr int SHADE_BLOCK = 256;

// rows of 256 pixels one block walks after staging the light table once
constexpr int SHADE_ROWS = 8;

__device__ __forceinline__ void shade_pixel(const ShadeParams& p, const LightLds* llds, int n_lights,
                                            uint32_t px, uint32_t py) {
    const size_t gi = (size_t)py * p.pitch + px;
    if (p.stencil[gi] == 0) return;   // stencil ref 0 < value (DeferredPipeline.h:176-181)

    const uint32_t a = p.A[gi], b = p.B[gi], c = p.C[gi];
    const float depth_ndc = p.depth[gi];

    // uv / camera ray from the GLOBAL pixel (vs_main :91-121, screen triangle D3D12Device.cpp:167-176)
    const float u = ((float)(p.x0 + px) + 0.5f) / (float)p.full_w;
    const float v = ((float)(p.y0 + py) + 0.5f) / (float)p.full_h;
    const float ndc_x = 2.0f * u - 1.0f, ndc_y = 1.0f - 2.0f * v;
    const float near_height = 2.0f * p.Near * tanf(p.Fov / 2.0f);
    const float near_width = near_height * p.Ratio;
    const V3 cvv = v3(ndc_x * 0.5f * near_width, ndc_y * 0.5f * near_height, p.Near);
    const V3 camera_vec = v3(p.InvView[0] * cvv.x + p.InvView[1] * cvv.y + p.InvView[2] * cvv.z,
                             p.InvView[3] * cvv.x + p.InvView[4] * cvv.y + p.InvView[5] * cvv.z,
                             p.InvView[6] * cvv.x + p.InvView[7] * cvv.y + p.InvView[8] * cvv.z);

    const float inv255 = 1.0f / 255.0f;
    const V3 albedo = v3((float)(a & 255u) / 255.0f, (float)((a >> 8) & 255u) / 255.0f, (float)((a >> 16) & 255u) / 255.0f);
    const float emission = (float)(a >> 24) / 255.0f;
    const float roughness = (float)(c & 255u) / 255.0f;
    const float metallic = (float)((c >> 8) & 255u) / 255.0f;
    (void)inv255;
    const V3 n = normalize3(decode_octahedron((float)(b & 255u) / 255.0f, (float)((b >> 8) & 255u) / 255.0f));

    // ViewSpaceDepth :74-77, ReconstructWorldPosition :79-83
    const float z_vs = p.Near * p.Far / (p.Far - depth_ndc * (p.Far - p.Near));
    const V3 cam = v3(p.CameraPos[0], p.CameraPos[1], p.CameraPos[2]);
    const float zs = z_vs / p.Near;
    const V3 pos = v3(cam.x + camera_vec.x * zs, cam.y + camera_vec.y * zs, cam.z + camera_vec.z * zs);
    const V3 view = normalize3(cam - pos);

    // ---- EnvironmentDiffuse :23-54
    V3 out;
    {
        const float bx = n.x * n.y, by = n.y * n.z, bz = n.z * n.z, bw = n.z * n.x;
        const float cc = n.x * n.x - n.y * n.y;
        const pbr_sh_pack& s = p.sh;
        float ir = (s.sha_r[0] * n.x + s.sha_r[1] * n.y + s.sha_r[2] * n.z + s.sha_r[3]) +
                   ((s.shb_r[0] * bx + s.shb_r[1] * by + s.shb_r[2] * bz + s.shb_r[3] * bw) + s.shc[0] * cc);
        float ig = (s.sha_g[0] * n.x + s.sha_g[1] * n.y + s.sha_g[2] * n.z + s.sha_g[3]) +
                   ((s.shb_g[0] * bx + s.shb_g[1] * by + s.shb_g[2] * bz + s.shb_g[3] * bw) + s.shc[1] * cc);
        float ib = (s.sha_b[0] * n.x + s.sha_b[1] * n.y + s.sha_b[2] * n.z + s.sha_b[3]) +
                   ((s.shb_b[0] * bx + s.shb_b[1] * by + s.shb_b[2] * bz + s.shb_b[3] * bw) + s.shc[2] * cc);
        const float kd = (1.0f - metallic) * INV_PI_F;
        out = v3(albedo.x * kd * ir, albedo.y * kd * ig, albedo.z * kd * ib);
    }

    // ---- EnvironmentSpecular :56-70
    const V3 F0 = v3(0.04f + metallic * (albedo.x - 0.04f), 0.04f + metallic * (albedo.y - 0.04f), 0.04f + metallic * (albedo.z - 0.04f));
    const float NdV = dot3(n, view);
    const float NdotV = fmaxf(NdV, 0.0f);
    {
        const V3 R = normalize3(n * (2.0f * NdV) - view);
        const F4 envc = cube_trilinear<CubeTexelF16>(p.env, p.env_size, p.env_mips, R, roughness * (float)PBR_ENV_MIPS);   // Q4
        const int lr = (int)p.lut_res;
        const BilinearCoord cx = bilinear_coord(roughness, lr), cy = bilinear_coord(NdotV, lr);   // Q5
        const int x0 = clampi(cx.i0, 0, lr - 1), x1 = clampi(cx.i1, 0, lr - 1);
        const int y0 = clampi(cy.i0, 0, lr - 1), y1 = clampi(cy.i1, 0, lr - 1);
        const H2* lut = reinterpret_cast<const H2*>(p.lut);
        const H2 l00 = lut[(size_t)y0 * lr + x0], l10 = lut[(size_t)y0 * lr + x1];
        const H2 l01 = lut[(size_t)y1 * lr + x0], l11 = lut[(size_t)y1 * lr + x1];
        const float wx0 = 1.0f - cx.f, wy0 = 1.0f - cy.f;
        const float la = ((float)l00.x * wx0 + (float)l10.x * cx.f) * wy0 + ((float)l01.x * wx0 + (float)l11.x * cx.f) * cy.f;
        const float lb = ((float)l00.y * wx0 + (float)l10.y * cx.f) * wy0 + ((float)l01.y * wx0 + (float)l11.y * cx.f) * cy.f;
        out.x += envc.x * (F0.x * la + lb);
        out.y += envc.y * (F0.y * la + lb);
        out.z += envc.z * (F0.z * la + lb);
    }

    // ---- clustered point lights :159-186
    int ci;
    {
        // ClusterIndex(uv, z), clustered.hlsli:45-60.  logf (not the fast intrinsic): the result is truncated.
        int sx = (int)floorf(u * (float)PBR_CLUSTER_X);
        int sy = (int)floorf((1.0f - v) * (float)PBR_CLUSTER_Y);
        float zc = fminf(fmaxf(z_vs, p.Near), p.Far);
        int sz = (int)((float)PBR_CLUSTER_Z * logf(zc / p.Near) / logf(p.Far / p.Near));
        sx = clampi(sx, 0, PBR_CLUSTER_X - 1);
        sy = clampi(sy, 0, PBR_CLUSTER_Y - 1);
        sz = clampi(sz, 0, PBR_CLUSTER_Z - 1);
        ci = sz + sx * PBR_CLUSTER_Z + sy * PBR_CLUSTER_X * PBR_CLUSTER_Z;
    }
    const pbr_cluster* cl = p.clusters + ci;
    const int nl = n_lights > 0 ? min(max(cl->NumLights, 0), PBR_MAX_LIGHTS_PER_CLUSTER) : 0;

    // loop invariants of brdf() (brdf.hlsli:47-67)
    const float ra = roughness * roughness;
    const float a4 = ra * ra;
    const float a4m1 = a4 - 1.0f;
    const float k = (roughness + 1.0f) * (roughness + 1.0f) / 8.0f;
    const float one_k = 1.0f - k;
    const float gv = NdotV / fmaxf(NdotV * one_k + k, EPSILON_F);
    const float one_m = 1.0f - metallic;
    const V3 omF0 = v3(1.0f - F0.x, 1.0f - F0.y, 1.0f - F0.z);
    V3 pl = v3(0.0f, 0.0f, 0.0f);
    for (int i = 0; i < nl; i++) {
        const int li = min(max(cl->LightIndex[i], 0), n_lights - 1);   // never index past the staged table
        const LightLds& r = llds[li];
        const float4 q0 = r.pos_int, q1 = r.col_c0, q2 = r.c1c2;
        const V3 lp = v3(q0.x, q0.y, q0.z), lc = v3(q1.x, q1.y, q1.z);
        const float intensity = q0.w, c0 = q1.w, c1 = q2.x, c2 = q2.y;
        V3 dir = lp - pos;
        const float d2 = dot3(dir, dir);
        const float invd = rsq(d2);
        const float dist = d2 * invd;
        dir = dir * invd;
        const float NdotL = fmaxf(dot3(n, dir), 0.0f);
        const V3 H = normalize3(dir + view);
        const float NdotH = fmaxf(dot3(n, H), 0.0f);
        // fresnel on NdotL (Q3)
        const float fm = fmaxf(1.0f - NdotL, EPSILON_F);
        const float fm2 = fm * fm;
        const float f5 = fm2 * fm2 * fm;
        const V3 F = v3(F0.x + omF0.x * f5, F0.y + omF0.y * f5, F0.z + omF0.z * f5);
        const float t = (NdotH * NdotH) * a4m1 + 1.0f;
        const float D = a4 * rcp(fmaxf(PI_F * t * t, EPSILON_F));
        const float gl = NdotL * rcp(fmaxf(NdotL * one_k + k, EPSILON_F));
        const float G = gv * gl;
        const float spec = D * G * rcp(fmaxf(4.0f * NdotL * NdotV, 0.0001f));
        const float att = rcp(fmaxf(c0 + c1 * dist + c2 * dist * dist, EPSILON_F));
        const float scale = intensity * att * NdotL;
        const float kdx = (1.0f - F.x) * one_m, kdy = (1.0f - F.y) * one_m, kdz = (1.0f - F.z) * one_m;
        pl.x += (kdx * albedo.x * INV_PI_F + F.x * spec) * lc.x * scale;
        pl.y += (kdy * albedo.y * INV_PI_F + F.y * spec) * lc.y * scale;
        pl.z += (kdz * albedo.z * INV_PI_F + F.z * spec) * lc.z * scale;
    }
    out = out + pl;
    // emission (Q1: the directional light of :144-156 is computed by the reference but never added)
    out = out + albedo * emission;
    store_h4(p.hdr + 4 * ((size_t)py * p.hdr_pitch + px), f4(out.x, out.y, out.z, 1.0f));
}

// grid (ceil(w/256), ceil(h/SHADE_ROWS)), block 256, dynamic LDS = n_lights * 48 B
__global__ __launch_bounds__(SHADE_BLOCK) void k_deferred_shade(ShadeParams p, int n_lights) {
    extern __shared__ float4 lds_raw[];
    LightLds* llds = reinterpret_cast<LightLds*>(lds_raw);
    for (int i = threadIdx.x; i < n_lights; i += SHADE_BLOCK) {
        const pbr_light l = p.lights[i];
        LightLds r;
        r.pos_int = make_float4(l.Position[0], l.Position[1], l.Position[2], l.Intensity);
        r.col_c0 = make_float4(l.Color[0], l.Color[1], l.Color[2], l.C0);
        r.c1c2 = make_float4(l.C1, l.C2, 0.0f, 0.0f);
        llds[i] = r;
    }
    __syncthreads();
    const uint32_t px = blockIdx.x * SHADE_BLOCK + threadIdx.x;
    if (px >= p.w) return;
    const uint32_t y_begin = blockIdx.y * SHADE_ROWS;
    const uint32_t y_end = min(y_begin + SHADE_ROWS, p.h);
    for (uint32_t py = y_begin; py < y_end; py++) shade_pixel(p, llds, n_lights, px, py);
}

extern "C" {

pbr_status pbr_deferred_shade(pbr_ctx* ctx, const pbr_global* g, const pbr_tile* tile, const pbr_gbuffer* gb,
                              const pbr_half* lut, uint32_t lut_res,
                              const pbr_half* env, uint32_t env_size, uint32_t env_mips,
                              const pbr_cluster* clusters, const pbr_light* lights, int num_lights,
                              pbr_half* hdr, uint32_t hdr_pitch) {
    if (!ctx) return PBR_ERR_INVALID;
    PBR_REQUIRE(ctx, g && tile && gb && lut && env && clusters && hdr, "pbr_deferred_shade: null pointer");
    PBR_REQUIRE(ctx, gb->A && gb->B && gb->C && gb->depth && gb->stencil, "pbr_deferred_shade: null G-buffer plane");
    PBR_REQUIRE(ctx, tile->w >= 1 && tile->h >= 1 && tile->w <= 65535 && tile->h <= 65535, "pbr_deferred_shade: bad tile size");
    PBR_REQUIRE(ctx, tile->x0 + tile->w <= tile->full_w && tile->y0 + tile->h <= tile->full_h, "pbr_deferred_shade: tile outside frame");
    PBR_REQUIRE(ctx, gb->pitch >= tile->w && hdr_pitch >= tile->w, "pbr_deferred_shade: pitch < width");
    PBR_REQUIRE(ctx, lut_res >= 1 && env_size >= 1 && env_mips >= 1 && (env_size >> (env_mips - 1)) >= 1, "pbr_deferred_shade: bad LUT/env size");
    PBR_REQUIRE(ctx, g->Near > 0.0f && g->Far > g->Near, "pbr_deferred_shade: need 0 < Near < Far");
    ShadeParams p;
    p.sh = g->SkyBoxSH;
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) p.InvView[r * 3 + c] = g->InvView[r * 4 + c];
    for (int i = 0; i < 3; i++) p.CameraPos[i] = g->CameraPos[i];
    p.Near = g->Near; p.Far = g->Far; p.Fov = g->Fov; p.Ratio = g->Ratio;
    p.x0 = tile->x0; p.y0 = tile->y0; p.w = tile->w; p.h = tile->h; p.full_w = tile->full_w; p.full_h = tile->full_h;
    p.A = gb->A; p.B = gb->B; p.C = gb->C; p.depth = gb->depth; p.stencil = gb->stencil; p.pitch = gb->pitch;
    p.lut = lut; p.lut_res = lut_res; p.env = env; p.env_size = env_size; p.env_mips = env_mips;
    p.clusters = clusters; p.lights = lights; p.hdr = hdr; p.hdr_pitch = hdr_pitch;
    PBR_REQUIRE(ctx, num_lights >= 0 && num_lights <= PBR_MAX_SCENE_LIGHTS, "pbr_deferred_shade: light count out of [0, 1024]");
    PBR_REQUIRE(ctx, num_lights == 0 || lights != nullptr, "pbr_deferred_shade: null lights");
    dim3 grid((tile->w + SHADE_BLOCK - 1) / SHADE_BLOCK, (tile->h + SHADE_ROWS - 1) / SHADE_ROWS);
    hipLaunchKernelGGL(k_deferred_shade, grid, dim3(SHADE_BLOCK), (size_t)num_lights * sizeof(LightLds), ctx->stream, p, num_lights);
    return launched(ctx, "k_deferred_shade");
}

}  // extern "C"
