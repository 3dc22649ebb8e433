/*
 * pbr_hip.h — C ABI of the MI355X-native deferred-PBR shading path.
 *
 * This is the drop-in boundary for the DeferredRendering HLSL compute / full-screen
 * passes of zrlhahaha/Direct12PBRRenderer.  The reference has no FFI: its seam is
 * D3D12CommandList::Dispatch(ShadingState*, gx, gy, gz) / DrawScreen(ShadingState*)
 * (Engine/Include/Renderer/Device/Direct12/D3D12CommandList.h:83,103).  Every entry
 * point below replaces ONE reference dispatch (cited per function); the C++ pass
 * classes under direct12pbrrenderer_amd/host/ call them from their Execute() bodies.
 *
 * Conventions
 *  - all image/buffer pointers are DEVICE pointers to linear, row-major planes;
 *  - every call enqueues on the context's stream and returns immediately
 *    (pbr_sync() blocks); a context is NOT thread-safe (one recording thread, as the
 *    reference: Engine/Source/App.cpp:370-377);
 *  - return value: 0 = ok, <0 = error (text via pbr_last_error);
 *  - "half" storage is IEEE binary16 in a uint16_t (R16G16_FLOAT / R16G16B16A16_FLOAT
 *    targets of the reference), RGBA8 planes are one uint32_t per pixel, R in the low
 *    byte (DXGI_FORMAT_R8G8B8A8_UNORM memory order).
 */
#ifndef PBR_HIP_H
#define PBR_HIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int pbr_status;
enum {
    PBR_OK              =  0,
    PBR_ERR_INVALID     = -1,  /* bad argument (null pointer, zero size, limits exceeded) */
    PBR_ERR_HIP         = -2,  /* HIP runtime error, see pbr_last_error                    */
    PBR_ERR_NOMEM       = -3,
    PBR_ERR_UNSUPPORTED = -4,
    PBR_ERR_COMM        = -5   /* RCCL error                                               */
};

typedef struct pbr_ctx pbr_ctx;
typedef uint16_t pbr_half;

/* ---- constants the reference hard-codes (kept in sync by hand there) ------------------ */
#define PBR_CLUSTER_X              24   /* DeferredRendering/Shader/clustered.hlsli:10-12   */
#define PBR_CLUSTER_Y              16
#define PBR_CLUSTER_Z              8
#define PBR_MAX_LIGHTS_PER_CLUSTER 32   /* clustered.hlsli:9                                */
#define PBR_MAX_SCENE_LIGHTS       1024 /* Engine/Include/Renderer/Pipeline/DeferredPipeline.h:329 */
#define PBR_NUM_CLUSTERS           (PBR_CLUSTER_X * PBR_CLUSTER_Y * PBR_CLUSTER_Z)
#define PBR_HISTOGRAM_BINS         256  /* DeferredPipeline.h:409                           */
#define PBR_SAMPLE_COUNT           1024 /* precompute_brdf.hlsl:3, env_map_gen.hlsl:3       */
#define PBR_ENV_MIPS               5    /* global.hlsli:9 PREFILTER_ENVMAP_MIPMAP_SIZE      */
#define PBR_BLOOM_STEP             3    /* DeferredPipeline.h:211                           */
#define PBR_BLOOM_MIPS             5    /* DeferredPipeline.h:212                           */

/* ---- POD mirrors of reference structs -------------------------------------------------- */

/* SH2CoefficientsPack, Engine/Include/Utils/SH.h:20-29 (7 x float4 = 112 B). */
typedef struct pbr_sh_pack {
    float sha_r[4], shb_r[4], sha_g[4], shb_g[4], sha_b[4], shb_b[4], shc[4];
} pbr_sh_pack;

/* ConstantBufferGlobal, Engine/Include/Renderer/Pipeline/IPipeline.h:38-62 == HLSL cbuffer
 * GlobalConstant, global.hlsli:38-57.  Matrices row-major, mul(M, v) = M*v (column vector). */
typedef struct pbr_global {
    pbr_sh_pack SkyBoxSH;
    float InvView[16];
    float View[16];
    float Projection[16];
    float InvProjection[16];
    float CameraPos[3];
    float Ratio;
    float Resolution[2];
    float Near;
    float Far;
    float Fov;
    float DeltaTime;
    float Time;
} pbr_global;                       /* 412 bytes */

/* PointLight, DeferredPipeline.h:341-347 / clustered.hlsli:31-37 (44 B). */
typedef struct pbr_light {
    float Position[3];
    float Color[3];
    float Intensity;
    float Radius, C0, C1, C2;       /* PointLightAttenuation, Scene.h:115-124 */
} pbr_light;

/* Cluster, DeferredPipeline.h:333-339 / clustered.hlsli:15-21 (156 B). */
typedef struct pbr_cluster {
    float MinBound[3];
    float MaxBound[3];
    int32_t NumLights;
    int32_t LightIndex[PBR_MAX_LIGHTS_PER_CLUSTER];
} pbr_cluster;

/* A tile of a larger frame (new: multi-GPU split).  Planes passed with a tile
 * cover w x h pixels (tile-local addressing); uv, the camera ray and ClusterIndex use the
 * GLOBAL pixel (x0 + x, y0 + y) of a full_w x full_h frame.  Single GPU: {0,0,W,H,W,H}. */
typedef struct pbr_tile {
    uint32_t x0, y0, w, h, full_w, full_h;
} pbr_tile;

/* G-buffer as written by gbuffer.hlsl::ps_main (gbuffer.hlsl:10-26,144-146); formats
 * DeferredPipeline.h:107-110.  A: rgb albedo (linear), a emission.  B: rg octahedral normal.
 * C: r roughness, g metallic, b AO.  depth: D32 in [0,1].  stencil > 0 <=> geometry.
 * pitch = row pitch in pixels of every plane. */
typedef struct pbr_gbuffer {
    const uint32_t* A;
    const uint32_t* B;
    const uint32_t* C;
    const float*    depth;
    const uint8_t*  stencil;
    uint32_t        pitch;
} pbr_gbuffer;

/* fp32 RGBA cube with a mip chain: mips concatenated (mip 0 first); inside a mip the six
 * faces +X,-X,+Y,-Y,+Z,-Z; each face row-major [y][x][4].  size = mip-0 edge. */
typedef struct pbr_cube_f32 {
    const float* data;
    uint32_t size;
    uint32_t mips;
} pbr_cube_f32;

/* ---- layout helpers (pure host functions) ----------------------------------------------- */
/* texels (not bytes) in a cube with `mips` levels */
size_t pbr_cube_texels(uint32_t size, uint32_t mips);
/* texel offset of (mip, face 0) */
size_t pbr_cube_mip_offset(uint32_t size, uint32_t mip);
/* texels of the "padded" prefiltered-env layout pbr_deferred_shade samples — a FOOTPRINT layout: for every bilinear
 * footprint origin (x, y) in [-1, s-1]^2 of every face of every mip, the four texels (x,y), (x+1,y), (x,y+1), (x+1,y+1),
 * each already resolved by the seamless-cube rule, stored together (32 contiguous bytes): the per-pixel trilinear fetch
 * is branch-free and touches ONE cache line per level.  4 x the plain chain (67 MB at 512^2 x 5).  Entry (face, y+1, x+1)
 * of mip m starts at texel pbr_env_padded_mip_offset(size, m) + ((face * (s+1) + y+1) * (s+1) + x+1) * 4. */
size_t pbr_env_padded_texels(uint32_t size, uint32_t mips);
size_t pbr_env_padded_mip_offset(uint32_t size, uint32_t mip);
/* texels of a PBR_BLOOM_MIPS-level 2D chain of a w x h image (level l is (w>>l) x (h>>l)) */
size_t pbr_bloom_chain_texels(uint32_t w, uint32_t h);
size_t pbr_bloom_level_offset(uint32_t w, uint32_t h, uint32_t level);

/* ---- context ------------------------------------------------------------------------------ */
pbr_status  pbr_ctx_create(int hip_device, pbr_ctx** out);
void        pbr_ctx_destroy(pbr_ctx* ctx);
/* enqueue on an existing hipStream_t (e.g. torch's current stream).  NULL selects HIP's default
 * (null) stream.  A new context starts on a private non-blocking stream. */
pbr_status  pbr_ctx_set_stream(pbr_ctx* ctx, void* hip_stream);
/* go back to the context's private stream */
pbr_status  pbr_ctx_use_own_stream(pbr_ctx* ctx);
/* the hipStream_t the next call will enqueue on (for a host that orders its own events / copies with the context's
 * work: the C++ pass graph records its per-frame fence there); NULL = HIP's default stream */
void*       pbr_ctx_get_stream(const pbr_ctx* ctx);
/* A second, high-priority stream of the context.  _begin: it waits for everything enqueued so far, and the calls made
 * until _end enqueue on it; _end: back to the context's stream — what follows runs concurrently with the side stream's
 * work; _join: the context's stream waits for the side stream.  (Multi-GPU: the tile's border ring and its halo exchange
 * on the side stream, the tile's core on the main one.) */
pbr_status  pbr_ctx_side_begin(pbr_ctx* ctx);
pbr_status  pbr_ctx_side_end(pbr_ctx* ctx);
pbr_status  pbr_ctx_side_join(pbr_ctx* ctx);
/* Bloom, the two large 2x-up levels of frames above ~1.6 Mpixel (levels of >= 400 tiles of 128 x 32): by default they run in polyphase
 * form — not the shader's operation order: <= 1 fp16 ULP per stage, <= 2 for the chain — so a whole frame and a smaller tile of it (which
 * takes the shader-order kernels) agree to 2 fp16 ULP, not bit for bit.  on != 0: every level in the shader's operation order
 * (k_blur_hv), bit-identical to the staged dispatches at any size — for hosts that compare tiles with frames or frames across sizes;
 * costs the polyphase form's gain (~2 % of a 4K frame). */
pbr_status  pbr_ctx_set_bloom_shader_order(pbr_ctx* ctx, int on);
const char* pbr_last_error(const pbr_ctx* ctx);
/* blocks until everything enqueued through the context is done: its stream AND side-stream work not joined yet */
pbr_status  pbr_sync(pbr_ctx* ctx);
const char* pbr_version(void);
/* NULL, or why this process cannot use the library: a second ROCm installation is mapped next to the HIP runtime in
 * use (PyTorch ships its own libamdhip64; it must be loaded first).  pbr_ctx_create refuses with
 * PBR_ERR_UNSUPPORTED in that case and prints this text. */
const char* pbr_runtime_error(void);
/* the rule behind pbr_runtime_error as a pure function: 1 iff the HIP runtime in use (mapped from hip_dir) is not the
 * one PyTorch (libtorch_hip.so in torch_dir) ships — i.e. torch_dir holds its own libamdhip64.so and hip_dir is another
 * directory.  A PyTorch built against the system ROCm (no bundled runtime) is NOT a mismatch. */
int pbr_runtime_mismatch_dirs(const char* hip_dir, const char* torch_dir);

/* ---- one-shot IBL precompute --------------------------------------------------------------- */
/* precompute_brdf.hlsl:20-62 dispatched by PrecomputeBRDFPass::Execute (DeferredPipeline.cpp:117-136).
 * out: res*res half2, row-major [y][x]; x -> roughness, y -> NdotV. */
pbr_status pbr_brdf_lut(pbr_ctx* ctx, uint32_t res, pbr_half* out_rg);

/* Radiance RGBE texels (R, G, B mantissas + shared exponent, 4 bytes) -> fp32 RGBA, alpha 1: the per-texel half of
 * DirectX::LoadFromHDRFile as called by ResourceLoader::LoadHDRImageFile (ResourceLoader.cpp:381-406; DirectXTex
 * is an un-vendored, unpinned vcpkg dependency).  Published rule (G. Ward, "Real Pixels", Graphics Gems II):
 * e == 0 -> 0, else channel = mantissa * 2^(e - 136).  The file-level parse (header, scanline RLE) is host work
 * (host/HdrImage.h). */
pbr_status pbr_rgbe_decode(pbr_ctx* ctx, const uint8_t* rgbe, size_t texels, float* out_rgba);

/* 2x2 box mips of an fp32 RGBA cube in place (stands in for DirectXTex GenerateMipMaps,
 * ResourceLoader.cpp:465-507).  cube->data mip 0 must be filled; mips 1.. are written. */
pbr_status pbr_cube_gen_mips(pbr_ctx* ctx, float* cube_data, uint32_t size, uint32_t mips);

/* env_map_gen.hlsl:50-105, all PBR_ENV_MIPS dispatches of PreFilterEnvMapPass::Execute
 * (DeferredPipeline.cpp:77-115): mip i is filtered with roughness i/(mips-1).
 * out: half4 cube chain, layout as pbr_cube_f32 with edge `size`.
 * Mips >= 1 are sampled from a half-precision copy of the source chain WHEN THAT COPY IS EXACT — every rgb texel of every
 * source mip survives fp32 -> half -> fp32 bit for bit, which is the case for everything the reference can feed this pass (its
 * sky assets are BC6H_UF16: BasicStorage.h:10-11) — and from the fp32 chain otherwise; decided on the device, the call stays
 * asynchronous.  Either way <= 1 fp16 ULP (or 1e-3 relative) from the shader's sequential sum.
 * The per-mip GGX sample tables depend on (size, mips, sky->mips) only: the context keeps the last set on the device, so the FIRST
 * call with a new shape builds and uploads them (a blocking copy, ~0.2 ms of host work) and later calls do not; like every entry
 * point, one call at a time per context. */
pbr_status pbr_prefilter_env(pbr_ctx* ctx, const pbr_cube_f32* sky, uint32_t size, uint32_t mips,
                             pbr_half* out_rgba);
/* ONE dispatch of env_map_gen.hlsl: cbuffer {Roughness, MipLevel, PrefilterEnvMapTextureSize}
 * (DeferredPipeline.h:46-51).  out_mip_rgba: the 6 x (size>>mip_level)^2 half4 texels of that mip. */
pbr_status pbr_prefilter_env_mip(pbr_ctx* ctx, const pbr_cube_f32* sky, uint32_t size, uint32_t mip_level,
                                 float roughness, pbr_half* out_mip_rgba);

/* Build the padded copy of a prefiltered env chain (one-shot, after pbr_prefilter_env).
 * env_rgba: plain half4 cube chain (pbr_cube_f32 layout); out_padded: pbr_env_padded_texels half4. */
pbr_status pbr_env_pad(pbr_ctx* ctx, const pbr_half* env_rgba, uint32_t size, uint32_t mips, pbr_half* out_padded);

/* SHBaker::ProjectEnvironmentMap + PackCubeMapSHCoefficient (Engine/Source/Utils/SH.cpp:87-153,
 * 201-222) as a deterministic solid-angle quadrature over every mip-0 texel.
 * out_pack: DEVICE pointer to 28 floats (pbr_sh_pack). */
pbr_status pbr_sh9_project(pbr_ctx* ctx, const pbr_cube_f32* sky, float* out_pack);

/* ---- per-frame passes ----------------------------------------------------------------------- */
/* clustered_compute.hlsl:18-42 (ClusteredPass::Execute, DeferredPipeline.cpp:253).
 * g: HOST pointer (copied into kernel arguments).  clusters: device, PBR_NUM_CLUSTERS. */
pbr_status pbr_cluster_build(pbr_ctx* ctx, const pbr_global* g, pbr_cluster* clusters);

/* clustered_culling.hlsl:18-41 (DeferredPipeline.cpp:256).  lights: device, n <= 1024. */
pbr_status pbr_cluster_cull(pbr_ctx* ctx, const pbr_global* g, const pbr_light* lights, int n,
                            pbr_cluster* clusters);
/* both dispatches of ClusteredPass::Execute (DeferredPipeline.cpp:253-256) in one launch: bounds + light lists;
 * same results as pbr_cluster_build followed by pbr_cluster_cull */
pbr_status pbr_clustered(pbr_ctx* ctx, const pbr_global* g, const pbr_light* lights, int num_lights,
                         pbr_cluster* clusters);

/* deferred_shading.hlsl:91-192 full-screen pass, stencil-masked (DeferredPipeline.cpp:187-206).
 * gb: HOST struct of device planes.  lut: res x res half2.  env_padded: the PADDED half4 cube chain
 * produced by pbr_env_pad (the fixed-function seamless-cube addressing, done once instead of per tap).
 * lights / num_lights: the PointLights buffer the cluster lists index (num_lights <= 1024; the
 * kernel stages exactly num_lights records into LDS, indices are clamped to that range).
 * hdr: tile-local w x h half4, pitch hdr_pitch pixels; untouched where stencil == 0.
 * Limits (32-bit offsets inside the kernel, PBR_ERR_INVALID beyond them): pitch x tile rows x 16 bytes < 4 GiB per plane,
 * lut_res <= 16384, padded env chain < 4 GiB (env_size <= 4096 with a full mip chain). */
pbr_status pbr_deferred_shade(pbr_ctx* ctx, const pbr_global* g, const pbr_tile* tile,
                              const pbr_gbuffer* gb,
                              const pbr_half* lut, uint32_t lut_res,
                              const pbr_half* env_padded, uint32_t env_size, uint32_t env_mips,
                              const pbr_cluster* clusters, const pbr_light* lights, int num_lights,
                              pbr_half* hdr, uint32_t hdr_pitch);

/* pbr_deferred_shade on up to 5 rectangles of the tile (tile-local {x, y, w, h}) in ONE launch; pixels outside are left
 * untouched.  The overlapped multi-GPU frame shades the tile's border ring first (<= 4 rectangles), starts the halo
 * exchange of its bloom strips, and shades the core while they travel. */
pbr_status pbr_deferred_shade_rects(pbr_ctx* ctx, const pbr_global* g, const pbr_tile* tile,
                                    const pbr_gbuffer* gb,
                                    const pbr_half* lut, uint32_t lut_res,
                                    const pbr_half* env_padded, uint32_t env_size, uint32_t env_mips,
                                    const pbr_cluster* clusters, const pbr_light* lights, int num_lights,
                                    pbr_half* hdr, uint32_t hdr_pitch, const uint32_t (*rects)[4], uint32_t n_rects);

/* Parity probe (not a product path): the same shade, storing the fp32 colour (float4 per pixel, alpha 1, pitch
 * hdr_pitch pixels, 16-byte aligned) instead of rounding it to the half4 target — the buffer the <= 1e-4 relative
 * L-inf parity bound is stated on. */
pbr_status pbr_deferred_shade_f32(pbr_ctx* ctx, const pbr_global* g, const pbr_tile* tile,
                                  const pbr_gbuffer* gb,
                                  const pbr_half* lut, uint32_t lut_res,
                                  const pbr_half* env_padded, uint32_t env_size, uint32_t env_mips,
                                  const pbr_cluster* clusters, const pbr_light* lights, int num_lights,
                                  float* hdr_f32, uint32_t hdr_pitch);

/* ---- SURVEY 8f "next" rows: the two raster passes either side of the shade, minus rasterization ---- */
/* skybox.hlsl:12-28 (SkyboxPass::Execute, DeferredPipeline.cpp:59-75): the sky sphere is drawn at the far
 * plane with depth test and no depth write, i.e. it lands exactly on the pixels geometry did not cover
 * (stencil == 0).  hdr(px) = SkyBox.Sample(LinearWrap, camera ray through px).rgb, alpha 1; pixels with
 * stencil > 0 are left untouched (the shade overwrites them).  The sampler's implicit LOD is defined as
 * log2 of the larger forward-difference footprint (in mip-0 texels) of the ray on the centre pixel's face. */
pbr_status pbr_skybox(pbr_ctx* ctx, const pbr_global* g, const pbr_tile* tile, const pbr_cube_f32* sky,
                      const uint8_t* stencil, uint32_t pitch, pbr_half* hdr, uint32_t hdr_pitch);

/* gbuffer.hlsl::ps_main :88-149 without the rasterizer / texture fetches: per-pixel material attributes ->
 * G-buffer planes.  m0 = (albedo.rgb as authored (gamma space), emission), m1 = (normal_ws.xyz, roughness),
 * m2 = (metallic, ambient occlusion, -, -): three float4 planes of pitch `pitch` pixels.
 * A = UNORM8(decode_gamma(albedo), emission), B = UNORM8(octahedral(normalize(n)), 1, 0),
 * C = UNORM8(roughness, metallic, ao, 0)  (global.hlsli:73-77,101-133; formats DeferredPipeline.h:107-109). */
pbr_status pbr_gbuffer_encode(pbr_ctx* ctx, const float* m0, const float* m1, const float* m2,
                              uint32_t w, uint32_t h, uint32_t pitch, uint32_t* A, uint32_t* B, uint32_t* C);

/* bloom_prefilter.hlsl:17-60 (DeferredPipeline.cpp:411-427): hdr (w x h) -> out (w>>1 x h>>1). */
pbr_status pbr_bloom_prefilter(pbr_ctx* ctx, const pbr_half* hdr, uint32_t w, uint32_t h,
                               uint32_t pitch, pbr_half* out, float threshold, float knee);
/* blur_horizontal.hlsl / blur.hlsli:24-55: in (iw x ih) sampled bilinearly at the texel
 * centres of out (ow x oh), 9 taps one OUTPUT texel apart in x. */
pbr_status pbr_blur_h(pbr_ctx* ctx, const pbr_half* in, uint32_t iw, uint32_t ih,
                      pbr_half* out, uint32_t ow, uint32_t oh);
/* blur_vertical.hlsl / blur.hlsli:58-89 */
pbr_status pbr_blur_v(pbr_ctx* ctx, const pbr_half* in, uint32_t iw, uint32_t ih,
                      pbr_half* out, uint32_t ow, uint32_t oh);
/* bloom_upsample_add.hlsl:13-25: out = H(lower) + H(upper), out has upper's size. */
pbr_status pbr_bloom_upsample_add(pbr_ctx* ctx, const pbr_half* upper, uint32_t uw, uint32_t uh,
                                  const pbr_half* lower, uint32_t lw, uint32_t lh, pbr_half* out);
/* bloom_merge.hlsl:7-11: hdr += in (both w x h; hdr pitch in pixels). */
pbr_status pbr_bloom_merge(pbr_ctx* ctx, pbr_half* hdr, uint32_t pitch, const pbr_half* in,
                           uint32_t w, uint32_t h);
/* One upsample level of BloomPass::Execute as ONE call (DeferredPipeline.cpp:472-540: the `Upsample Horizontal Add` +
 * `Blur Vertical` pair; with upper == NULL the `Upsample Merge` pair blur_horizontal + blur_vertical, :541-559, without the merge):
 *   out = V( H(upper) + H(lower sampled at out's size) )        out, upper: ow x oh;  lower: lw x lh, ow == 2 lw, oh == 2 lh
 * — the fused kernel pbr_bloom runs for such a level pair, exposed for stage-level tests and hosts that fuse the pair.  The H
 * result is rounded to fp16 where the first dispatch stores it.  Levels of >= 400 tiles of 128 x 32 texels take the polyphase
 * form of the 2x-up blur (csrc/bloom.hip: k_blur_up_poly): <= 1 fp16 ULP from the two staged calls (SURVEY 8c's bloom-stage
 * tolerance), not bit-identical; smaller levels are bit-identical.  Sizes must be even and <= 8192; out must not alias an input. */
pbr_status pbr_bloom_up_level(pbr_ctx* ctx, const pbr_half* upper, const pbr_half* lower, uint32_t lw, uint32_t lh,
                              pbr_half* out, uint32_t ow, uint32_t oh);
/* BloomPass::Execute (DeferredPipeline.cpp:400-570), all 16 dispatches.  hdr is bit-identical to the sequence of stage calls
 * above for frames below ~1.6 Mpixel; from there on the large 2x-up levels run in polyphase form (pbr_bloom_up_level) and hdr
 * is within 2 fp16 ULP of that sequence (>= 99.9 % of the texels identical).  chain_a / chain_b: pbr_bloom_chain_texels(w,h) half4 texels each (BloomMipchain /
 * BloomTempTexture) — SCRATCH: their contents after the call are unspecified.  (Wherever a level is exactly half
 * the one above and at most 8192 wide/high, the H and V pass of that level pair run as one kernel and the H result
 * is never written; elsewhere the staged kernels run.  Level 0 of chain_a — the V blur the merge consumes — is
 * never written.) */
pbr_status pbr_bloom(pbr_ctx* ctx, pbr_half* hdr, uint32_t w, uint32_t h, uint32_t pitch,
                     pbr_half* chain_a, pbr_half* chain_b, float threshold, float knee);

/* bloom_prefilter.hlsl on part of the image (multi-GPU halo path): the half-res outputs rect = {x, y, w, h} of the
 * w x h image `hdr` are computed and stored at out[(out_y + y) * out_pitch + (out_x + x)] — a tile writes the level-1
 * texels of its interior into the level-1 plane of its extended rectangle. */
pbr_status pbr_bloom_prefilter_rect(pbr_ctx* ctx, const pbr_half* hdr, uint32_t w, uint32_t h, uint32_t pitch,
                                    pbr_half* out, uint32_t out_pitch, uint32_t out_x, uint32_t out_y,
                                    const uint32_t rect[4], float threshold, float knee);
/* the same for up to 5 rectangles in one launch (the four bands of a tile's border ring) */
pbr_status pbr_bloom_prefilter_rects(pbr_ctx* ctx, const pbr_half* hdr, uint32_t w, uint32_t h, uint32_t pitch,
                                     pbr_half* out, uint32_t out_pitch, uint32_t out_x, uint32_t out_y,
                                     const uint32_t (*rects)[4], uint32_t n_rects, float threshold, float knee);
/* BloomPass::Execute minus its first dispatch, on the extended rectangle E (ew x eh, a multiple of 16 and <= 8192 on
 * a side) of a tile: level 1 of chain_a (offset pbr_bloom_level_offset(ew, eh, 1)) must hold the prefiltered image of
 * ALL of E — the interior from pbr_bloom_prefilter_rect, the rest from the neighbouring tiles (pbr_halo_exchange).
 * Runs the 3 + 3 level pairs on E and merges (DeferredPipeline.cpp:521-570) only merge_rect = {x, y, w, h} of E into
 * hdr, which covers hdr_rect of E (hdr[0] = texel (hdr_rect.x, hdr_rect.y), pitch hdr_pitch pixels).  hist256 != NULL:
 * the luminance histogram of the merged pixels is added (pbr_lum_histogram).  Chains are scratch. */
pbr_status pbr_bloom_tiled(pbr_ctx* ctx, pbr_half* hdr, uint32_t hdr_pitch, const uint32_t hdr_rect[4],
                           uint32_t ew, uint32_t eh, pbr_half* chain_a, pbr_half* chain_b, const uint32_t merge_rect[4],
                           float min_log, float inv_range, uint32_t* hist256);

/* pbr_bloom + the luminance histogram (pbr_lum_histogram) of rect = {x, y, w, h} of the bloomed
 * image, accumulated inside the final bloom kernel (saves one full read of the HDR buffer).
 * ADDS into hist256 like the separate pass. */
pbr_status pbr_bloom_histogram(pbr_ctx* ctx, pbr_half* hdr, uint32_t w, uint32_t h, uint32_t pitch,
                               pbr_half* chain_a, pbr_half* chain_b, float threshold, float knee,
                               const uint32_t rect[4], float min_log, float inv_range, uint32_t* hist256);

/* hdr_luminance_histogram.hlsl:23-59 (DeferredPipeline.cpp:276-298): ADDS into hist256. */
pbr_status pbr_lum_histogram(pbr_ctx* ctx, const pbr_half* hdr, uint32_t w, uint32_t h,
                             uint32_t pitch, float min_log, float inv_range, uint32_t* hist256);
/* hdr_average_histogram.hlsl:26-73 (DeferredPipeline.cpp:300-317): updates *avg_inout, zeroes hist. */
pbr_status pbr_lum_average(pbr_ctx* ctx, uint32_t* hist256, uint32_t pixel_count, float min_log,
                           float range, float delta_time, float* avg_inout);
/* hdr_tone_mapping.hlsl:9-52 (DeferredPipeline.cpp:320-336): hdr -> RGBA8 UNORM. */
pbr_status pbr_tonemap(pbr_ctx* ctx, const pbr_half* hdr, uint32_t w, uint32_t h, uint32_t pitch,
                       const float* avg, uint32_t* rgba8, uint32_t out_pitch);
/* hdr_average_histogram.hlsl + hdr_tone_mapping.hlsl as ONE launch (the last two dispatches of a frame): *avg_out = the adapted
 * luminance of pbr_lum_average(hist256, *avg_in), rgba8 = pbr_tonemap(hdr, *avg_out) — bit-identical to the two calls.  Every block of
 * the launch re-derives the average from the 256 bins, so NOTHING it reads may be written by it: avg_out != avg_in, and the histogram
 * zeroed "for the next frame" is hist_clear256 != hist256 (NULL: none) — a caller alternates two histograms (the next frame accumulates
 * into the one cleared here; the counts read here are cleared by the next frame's call) and two luminance cells. */
pbr_status pbr_average_tonemap(pbr_ctx* ctx, const uint32_t* hist256, uint32_t pixel_count, float min_log, float range, float delta_time,
                               const float* avg_in, float* avg_out, uint32_t* hist_clear256,
                               const pbr_half* hdr, uint32_t w, uint32_t h, uint32_t pitch, uint32_t* rgba8, uint32_t out_pitch);
/* ---- multi-GPU (new) -------------------------------------------------------------- */
/* RCCL communicator over the ranks of one node.  unique_id: 128 bytes from
 * pbr_comm_unique_id() on rank 0, broadcast by the caller (e.g. torch.distributed store). */
pbr_status pbr_comm_unique_id(void* out_128_bytes);
/* Collective over all ranks.  Creates TWO communicators over the same ranks: the frame communicator (halo exchange)
 * from the unique id, and an ncclCommSplit of it for the histogram all-reduce — the two collectives may be in flight
 * on different streams at the same time (overlapped frame tail), and RCCL orders operations per communicator only. */
pbr_status pbr_comm_init(pbr_ctx* ctx, int world, int rank, const void* unique_id_128_bytes);
/* ncclAllReduce(sum, uint32, 256) on the ctx stream, on the histogram communicator; no-op without a communicator / world 1 */
pbr_status pbr_allreduce_hist(pbr_ctx* ctx, uint32_t* hist256);

/* Halo exchange of half4 rectangles of one plane (level 1 of the bloom pyramid) with neighbouring tiles: for every
 * peer, `send` = {x, y, w, h} of the plane that goes to rank `rank`, `recv` = the rectangle that arrives from it
 * (w == 0: nothing).  Both sides derive the rectangles from the tile layout, so sizes never travel.  One pack launch,
 * one ncclGroup of ncclSend / ncclRecv over xGMI, one unpack launch, all on the ctx stream.
 * staging: device scratch of pbr_halo_staging_bytes(); at most 16 peers. */
typedef struct pbr_halo_peer {
    int32_t  rank;
    uint32_t send[4];
    uint32_t recv[4];
} pbr_halo_peer;
size_t     pbr_halo_staging_bytes(const pbr_halo_peer* peers, uint32_t n_peers);
pbr_status pbr_halo_exchange(pbr_ctx* ctx, pbr_half* plane, uint32_t pitch, uint32_t rows,
                             const pbr_halo_peer* peers, uint32_t n_peers, void* staging, size_t staging_bytes);
/* the pack (unpack = 0: send rectangles -> staging) and unpack (unpack = 1: staging -> recv rectangles) halves on
 * their own, for a transport other than the context's communicator.  Staging layout: all send rectangles in peer
 * order, then all recv rectangles. */
pbr_status pbr_halo_pack(pbr_ctx* ctx, pbr_half* plane, uint32_t pitch, uint32_t rows,
                         const pbr_halo_peer* peers, uint32_t n_peers, void* staging, size_t staging_bytes, int unpack);

/* ---- measurement aid ---------------------------------------------------------------------------- */
/* Streaming read of `bytes` bytes (16-byte loads, grid-stride over `blocks` blocks of 256 lanes, one xor word per
 * block into sink[blocks]) on the ctx stream: the kernel bench.py times to report the MEASURED HBM-read bandwidth of
 * the device next to the 8 TB/s nominal peak.  buf 16-byte aligned. */
pbr_status pbr_membench_read(pbr_ctx* ctx, const void* buf, size_t bytes, uint32_t* sink, uint32_t blocks);
/* VALU issue-rate probe: `blocks` blocks of 256 lanes (one wave per SIMD each; blocks = CUs x waves-per-SIMD fills the
 * chip at that occupancy), every wave issuing iters x 8 independent instructions of one class — op 0: v_mul_f32 (the plain
 * class), 1: v_fma_f32, 2: v_pk_fma_f32 (packed fp32), 3: v_rcp_f32 (transcendental) — between two reads of the shader-core
 * cycle counter (s_memtime) and of the constant 100 MHz counter (s_memrealtime).  stamps: DEVICE, 4 x uint64 per wave
 * {cycles at start, at end, 100 MHz ticks at start, at end}, blocks * 4 waves.  With the launch duration (HIP events on the
 * ctx stream) this gives the chip's sustained issue rate in wave-instructions/s for that class AND the shader clock it holds
 * under that load — the denominators bench.py's `roofline.valu` needs from the box it runs on, not from a committed file. */
pbr_status pbr_valubench(pbr_ctx* ctx, uint32_t op, uint32_t blocks, uint32_t iters, uint64_t* stamps);

/* ---- Knobs build only (libpbr_hip_knobs.so, -DPBR_DEBUG_KNOBS): measurement entry points that are NOT part of the product library.
 * Round 6: the CU partition left the product API — measured in rounds 4-5 on four boxes it never paid (throughput mode with the
 * partition: +0 ... +8 % frame time; EXPERIMENTS.md), and a drop-in does not carry a switch nobody should flip. ---- */
#ifdef PBR_DEBUG_KNOBS
/* Partition the device's compute units between the context's private stream and its side stream (throughput mode: a frame's bloom +
 * exposure tail on the side stream's CUs while the next frame's shade has the others to itself — two kernels that both want the whole
 * chip only take turns otherwise).  masks: bit i = CU i (hipExtStreamCreateWithCUMask), `words` 32-bit words each; NULL = every CU.
 * On MI355X the bits run XCD by XCD in groups of four (bits 0-3 = four CUs of XCD 0, 4-7 = of XCD 1, ... 32-35 = the next four of XCD 0),
 * and a kernel's workgroups are dealt to the XCDs in equal shares whatever their CU counts: a partition must hold the same number of CUs
 * of every XCD, and a group of four bits is honoured as a whole only — i.e. a multiple of 32 low bits — or its XCD with the fewest sets the
 * pace (measured: profiles/r04_h_cu_partition_*.txt, r04_j_cu_partition_per_xcd.txt).
 * Masked streams are created with hipExtStreamCreateWithCUMask, which takes no flags: unlike the non-blocking streams they replace, they
 * SYNCHRONISE WITH THE LEGACY NULL STREAM — any null-stream work of the process (a hipMemset, a default-stream torch op or event record)
 * serialises both partitions, so no default-stream work belongs inside a partitioned frame.
 * Recreates both streams (the context must be idle on its private stream: pbr_ctx_use_own_stream, no side work pending) and
 * waits for the device.  A context bound to a foreign stream (pbr_ctx_set_stream) keeps that stream: only the side stream is masked. */
pbr_status  pbr_ctx_set_cu_masks(pbr_ctx* ctx, const uint32_t* main_mask, const uint32_t* side_mask, uint32_t words);
#endif

#ifdef __cplusplus
}
#endif
#endif /* PBR_HIP_H */
