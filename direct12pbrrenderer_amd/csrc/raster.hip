// raster.hip — the passes and data formats either side of the shade, minus the rasterizer:
//   k_skybox          skybox.hlsl:12-28         (SkyboxPass::Execute, DeferredPipeline.cpp:59-75)
//   k_gbuffer_encode  gbuffer.hlsl::ps_main :88-149  (GBufferPass::Execute, DeferredPipeline.cpp:138-185)
//   k_rgbe_decode     Radiance .hdr texels -> fp32 (ResourceLoader::LoadHDRImageFile, ResourceLoader.cpp:381-406)
// Both are streaming, HBM-bound kernels: one lane per pixel, rows contiguous across the wave.
// Built with -ffp-contract=off: same operation order as the oracle (the ray feeds floor() in the
// cube addressing, the gamma/octahedral results feed UNORM8 rounding).
#include "pbr_internal.hpp"
#include "pbr_device.hpp"

using namespace pbr;

namespace {

struct SkyParams {
    float InvView[9];
    float near_width, near_height, Near;
    float full_w_f, full_h_f;
    uint32_t x0, y0, w, h;
    uint32_t sky_size, sky_mips;
    uint32_t pitch, hdr_pitch;
};

__device__ __forceinline__ V3 sky_ray(const SkyParams& p, float gx, float gy) {
    const float u = (gx + 0.5f) / p.full_w_f, v = (gy + 0.5f) / p.full_h_f;
    const float ndc_x = 2.0f * u - 1.0f, ndc_y = 1.0f - 2.0f * v;
    const V3 c = v3(ndc_x * 0.5f * p.near_width, ndc_y * 0.5f * p.near_height, p.Near);
    return v3(p.InvView[0] * c.x + p.InvView[1] * c.y + p.InvView[2] * c.z,
              p.InvView[3] * c.x + p.InvView[4] * c.y + p.InvView[5] * c.z,
              p.InvView[6] * c.x + p.InvView[7] * c.y + p.InvView[8] * c.z);
}

// sc/ma, tc/ma of `d` on a given face (no major-axis test): the footprint of the neighbouring
// rays is measured on the centre pixel's face
__device__ __forceinline__ void project_on_face(V3 d, uint32_t face, float& u, float& v) {
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

__global__ __launch_bounds__(256) void k_skybox(SkyParams p, const float* __restrict__ sky,
                                                const uint8_t* __restrict__ stencil, pbr_half* __restrict__ hdr) {
    const uint32_t px = blockIdx.x * 64u + (threadIdx.x & 63u);
    const uint32_t py = blockIdx.y * 4u + (threadIdx.x >> 6);
    if (px >= p.w || py >= p.h) return;
    if (stencil[(size_t)py * p.pitch + px] != 0) return;   // geometry: the shade owns this pixel
    const float gx = (float)(p.x0 + px), gy = (float)(p.y0 + py);
    const V3 d = sky_ray(p, gx, gy);
    uint32_t face; float fu, fv;
    cube_face_uv(d, face, fu, fv);
    float u0, v0, ux, vx, uy, vy;
    project_on_face(d, face, u0, v0);
    project_on_face(sky_ray(p, gx + 1.0f, gy), face, ux, vx);
    project_on_face(sky_ray(p, gx, gy + 1.0f), face, uy, vy);
    const float half_size = 0.5f * (float)p.sky_size;
    const float rx = half_size * sqrtf((ux - u0) * (ux - u0) + (vx - v0) * (vx - v0));
    const float ry = half_size * sqrtf((uy - u0) * (uy - u0) + (vy - v0) * (vy - v0));
    const float lod = log2f(fmaxf(rx, ry));
    const F4 c = cube_trilinear<CubeTexelF32>(sky, p.sky_size, p.sky_mips, d, lod);
    store_h4(hdr + 4 * ((size_t)py * p.hdr_pitch + px), f4(c.x, c.y, c.z, 1.0f));
}

__device__ __forceinline__ uint32_t unorm8(float x) { return (uint32_t)floorf(saturatef(x) * 255.0f + 0.5f); }
__device__ __forceinline__ float sign_custom(float x) { return x < 0.0f ? -1.0f : 1.0f; }
// decode_gamma (global.hlsli:73-77): pow(c, 2.2) the way the shader compiler lowers it, exp2(2.2 * log2(c)) on
// the transcendental unit (v_log_f32 / v_exp_f32, 1 ULP each).  The result only feeds an 8-bit UNORM target:
// relative error < 1e-6 moves a value across a rounding boundary on ~1e-4 of the texels (by one step).
// pow(0) = 0, pow(negative) = NaN -> saturate -> 0, like the libm formulation.
__device__ __forceinline__ float decode_gamma(float c) {
    return __builtin_amdgcn_exp2f(2.2f * __builtin_amdgcn_logf(c));
}

__global__ __launch_bounds__(256) void k_gbuffer_encode(const float4* __restrict__ m0, const float4* __restrict__ m1,
                                                        const float4* __restrict__ m2, uint32_t w, uint32_t h,
                                                        uint32_t pitch, uint32_t* __restrict__ A,
                                                        uint32_t* __restrict__ B, uint32_t* __restrict__ C) {
    const uint32_t x = blockIdx.x * 64u + (threadIdx.x & 63u);
    const uint32_t y = blockIdx.y * 4u + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const size_t i = (size_t)y * pitch + x;
    const float4 a = m0[i], b = m1[i], c = m2[i];
    // decode_gamma, global.hlsli:73-77
    const uint32_t pa = unorm8(decode_gamma(a.x)) | (unorm8(decode_gamma(a.y)) << 8) | (unorm8(decode_gamma(a.z)) << 16) |
                        (unorm8(a.w) << 24);
    // pack_normal(normalize(n)), global.hlsli:117-128
    V3 n = normalize3_exact(v3(b.x, b.y, b.z));
    const float sum = fabsf(n.x) + fabsf(n.y) + fabsf(n.z);
    float dx = n.x / sum, dy = n.y / sum;
    const float dz = n.z / sum;
    if (dz < 0.0f) {
        const float nx = sign_custom(dx) * (1.0f - fabsf(dy));
        const float ny = sign_custom(dy) * (1.0f - fabsf(dx));
        dx = nx; dy = ny;
    }
    const uint32_t pb = unorm8(dx * 0.5f + 0.5f) | (unorm8(dy * 0.5f + 0.5f) << 8) | (255u << 16);
    const uint32_t pc = unorm8(b.w) | (unorm8(c.x) << 8) | (unorm8(c.y) << 16);
    A[i] = pa; B[i] = pb; C[i] = pc;
}

// Radiance RGBE -> fp32 RGBA (4 B in, 16 B out per texel; one texel per lane, grid-stride)
__global__ __launch_bounds__(256) void k_rgbe_decode(const uint32_t* __restrict__ rgbe, size_t texels, float4* __restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < texels; i += (size_t)gridDim.x * 256u) {
        const uint32_t v = rgbe[i];
        const int e = (int)(v >> 24);
        // 2^(e-136) built in the exponent field (e-136+127 in [-8, 246]: subnormal scale below e = 10)
        const float scale = e ? ldexpf(1.0f, e - 136) : 0.0f;
        out[i] = make_float4((float)(v & 255u) * scale, (float)((v >> 8) & 255u) * scale, (float)((v >> 16) & 255u) * scale, 1.0f);
    }
}

}  // namespace

extern "C" {

pbr_status pbr_rgbe_decode(pbr_ctx* ctx, const uint8_t* rgbe, size_t texels, float* out_rgba) {
    if (!ctx) return PBR_ERR_INVALID;
    PBR_REQUIRE(ctx, rgbe && out_rgba && texels, "pbr_rgbe_decode: null pointer / empty image");
    PBR_REQUIRE(ctx, (((uintptr_t)rgbe & 3u) | ((uintptr_t)out_rgba & 15u)) == 0u, "pbr_rgbe_decode: unaligned buffer");
    size_t blocks = (texels + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_rgbe_decode, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, reinterpret_cast<const uint32_t*>(rgbe), texels,
                       reinterpret_cast<float4*>(out_rgba));
    return pbr::launched(ctx, "k_rgbe_decode");
}

pbr_status pbr_skybox(pbr_ctx* ctx, const pbr_global* g, const pbr_tile* tile, const pbr_cube_f32* sky,
                      const uint8_t* stencil, uint32_t pitch, pbr_half* hdr, uint32_t hdr_pitch) {
    if (!ctx) return PBR_ERR_INVALID;
    PBR_REQUIRE(ctx, g && tile && sky && sky->data && stencil && hdr, "pbr_skybox: null pointer");
    PBR_REQUIRE(ctx, tile->w && tile->h && tile->full_w && tile->full_h && pitch >= tile->w && hdr_pitch >= tile->w,
                "pbr_skybox: bad tile / pitch");
    PBR_REQUIRE(ctx, sky->size && sky->mips && (sky->size >> (sky->mips - 1)) >= 1, "pbr_skybox: bad cube");
    SkyParams p;
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) p.InvView[r * 3 + c] = g->InvView[r * 4 + c];
    p.near_height = 2.0f * g->Near * tanf(g->Fov / 2.0f);
    p.near_width = p.near_height * g->Ratio;
    p.Near = g->Near;
    p.full_w_f = (float)tile->full_w; p.full_h_f = (float)tile->full_h;
    p.x0 = tile->x0; p.y0 = tile->y0; p.w = tile->w; p.h = tile->h;
    p.sky_size = sky->size; p.sky_mips = sky->mips;
    p.pitch = pitch; p.hdr_pitch = hdr_pitch;
    dim3 grid((tile->w + 63) / 64, (tile->h + 3) / 4);
    hipLaunchKernelGGL(k_skybox, grid, dim3(256), 0, ctx->stream, p, sky->data, stencil, hdr);
    return pbr::launched(ctx, "k_skybox");
}

pbr_status pbr_gbuffer_encode(pbr_ctx* ctx, const float* m0, const float* m1, const float* m2,
                              uint32_t w, uint32_t h, uint32_t pitch, uint32_t* A, uint32_t* B, uint32_t* C) {
    if (!ctx) return PBR_ERR_INVALID;
    PBR_REQUIRE(ctx, m0 && m1 && m2 && A && B && C, "pbr_gbuffer_encode: null pointer");
    PBR_REQUIRE(ctx, w && h && pitch >= w, "pbr_gbuffer_encode: bad size");
    PBR_REQUIRE(ctx, ((((uintptr_t)m0) | ((uintptr_t)m1) | ((uintptr_t)m2)) & 15u) == 0u,
                "pbr_gbuffer_encode: material planes must be 16-byte aligned");
    dim3 grid((w + 63) / 64, (h + 3) / 4);
    hipLaunchKernelGGL(k_gbuffer_encode, grid, dim3(256), 0, ctx->stream, reinterpret_cast<const float4*>(m0),
                       reinterpret_cast<const float4*>(m1), reinterpret_cast<const float4*>(m2), w, h, pitch, A, B, C);
    return pbr::launched(ctx, "k_gbuffer_encode");
}

}  // extern "C"
