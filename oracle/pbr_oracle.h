/*
 * pbr_oracle.h — CPU restatement of the reference's deferred-PBR shading arithmetic.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only
 * as the checker / the timed CPU baseline.  The HIP path never calls it.
 *
 * PARITY UNPINNED: the reference (zrlhahaha/Direct12PBRRenderer) has no test, golden image
 * or known-answer vector for any shader on this path (UnitTest/ covers allocators and the
 * thread pool only), its HLSL cannot be compiled here (no dxc / D3D12) and its CPU math
 * (Engine/Source/Utils/{MathLib,SH}.cpp) does not build outside MSVC/Windows.  This oracle is
 * therefore pinned only by analytic known-answer tests and by reading the shader source;
 * every function cites the reference lines it restates.
 *
 * Arithmetic model: fp32 throughout, no implicit FMA contraction (-ffp-contract=off); the
 * only fused operations are written as explicit fmaf: the samplers' lerps and the blur's
 * multiply-accumulate (HLSL `mad`).  libm transcendentals, dot products evaluated left to right, normalize(v) = v * (1/sqrt(v.v)),
 * lerp(a,b,t) = a + t*(b-a); D3D fixed-function behaviour is DEFINED here as:
 * filter addressing in fixed point as the D3D functional spec prescribes — the scaled coordinate u*size is
 * snapped to x.8 (round to nearest; D3D12_SUBTEXEL_FRACTIONAL_BIT_COUNT = 8) before the half-texel offset,
 * the trilinear LOD fraction likewise to 1/256 (D3D12_MIP_LOD_FRACTIONAL_BIT_COUNT = 8); the lerps
 * themselves run in fp32 and a tap with weight exactly 0 does not contribute; clamp addressing,
 * seamless cube edges (out-of-face taps are re-projected onto the neighbouring face, a tap
 * that leaves the face in both axes is first clamped in y), fp32->fp16 round-to-nearest-even
 * with overflow to inf, UNORM8 = floor(saturate(x)*255+0.5), UNORM8->float = c/255.
 */
#ifndef PBR_ORACLE_H
#define PBR_ORACLE_H

#include "../include/pbr_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* number of OpenMP threads the oracle will use (1 if built without OpenMP) */
int  orc_num_threads(void);
void orc_set_num_threads(int n);

/* fp16 helpers */
uint16_t orc_f32_to_f16(float f);
float    orc_f16_to_f32(uint16_t h);

/* ---- scalar pieces exported for known-answer tests ---- */
float orc_radical_inverse(uint32_t bits);                          /* brdf.hlsli:101-109 */
void  orc_ggx_sample(float roughness, const float n[3], float xi_x, float xi_y, float out_h[3]); /* brdf.hlsli:71-97 */
void  orc_brdf(float metallic, float roughness, const float albedo[3], const float n[3],
               const float v[3], const float l[3], float out[3]);  /* brdf.hlsli:47-67 */
void  orc_octa_decode(float u, float v, float out_n[3]);           /* global.hlsli:101-115,135-138 (normalized) */
void  orc_octa_encode(const float n[3], float out_uv[2]);          /* global.hlsli:117-128 */
float orc_view_space_depth(const pbr_global* g, float ndc_depth);  /* deferred_shading.hlsl:74-77 */
int   orc_cluster_index(const pbr_global* g, float u, float v, float z_vs); /* clustered.hlsli:45-60 */
float orc_attenuation(float d, float c0, float c1, float c2);      /* deferred_shading.hlsl:86-89 */
void  orc_aces(const float x[3], float out[3]);                    /* hdr_tone_mapping.hlsl:27-36 */
uint32_t orc_luminance_bin(float lum, float min_log, float inv_range); /* hdr_luminance_histogram.hlsl:23-35 */
void  orc_env_diffuse(const pbr_sh_pack* sh, const float albedo[3], float metallic,
                      const float n[3], float out[3]);             /* deferred_shading.hlsl:23-54 */
void  orc_cube_dir(uint32_t face, float u, float v, float out[3]); /* env_map_gen.hlsl:20-44 */
/* trilinear seamless-cube fetch of an fp32 cube (sampler s3, D3D12Device.cpp:665-684) */
void  orc_sample_cube_f32(const float* data, uint32_t size, uint32_t mips, const float dir[3],
                          float lod, float out[4]);
void  orc_sample_cube_f16(const uint16_t* data, uint32_t size, uint32_t mips, const float dir[3],
                          float lod, float out[4]);
void  orc_sample_2d_f16x4(const uint16_t* img, uint32_t w, uint32_t h, uint32_t pitch,
                          float u, float v, float out[4]);

/* ---- passes (host pointers everywhere; same layouts as include/pbr_hip.h) ---- */
int orc_brdf_lut(uint32_t res, uint16_t* out_rg);                                  /* a3  */
int orc_brdf_lut_rows(uint32_t res, uint32_t y0, uint32_t rows, uint16_t* out_rg); /* a3, rows [y0,y0+rows) written at out_rg[0..] */
int orc_cube_gen_mips(float* cube_data, uint32_t size, uint32_t mips);
int orc_prefilter_env(const float* sky, uint32_t sky_size, uint32_t sky_mips,
                      uint32_t size, uint32_t mips, uint16_t* out_rgba);           /* a4  */
/* one output mip only (for bounded tests / baseline timing) */
int orc_prefilter_env_mip(const float* sky, uint32_t sky_size, uint32_t sky_mips,
                          uint32_t size, uint32_t mips, uint32_t mip, uint16_t* out_mip_rgba);
int orc_prefilter_env_texels(const float* sky, uint32_t sky_size, uint32_t sky_mips, uint32_t size, uint32_t mips,
                             uint32_t mip, const uint32_t* texels, uint32_t count, uint16_t* out);   /* a4 on chosen texels */
int orc_sh9_project(const float* sky_mip0, uint32_t size, float out_pack[28]);     /* a5 quadrature */
int orc_sh9_project_mc(const float* sky_mip0, uint32_t size, uint32_t seed, uint32_t samples,
                       float out_pack[28]);                                        /* a5 seeded MC restatement */
int orc_cluster_build(const pbr_global* g, pbr_cluster* clusters);                 /* a13 */
int orc_cluster_cull(const pbr_global* g, const pbr_light* lights, int n, pbr_cluster* clusters);
int orc_deferred_shade(const pbr_global* g, const pbr_tile* tile, const pbr_gbuffer* gb,
                       const uint16_t* lut, uint32_t lut_res,
                       const uint16_t* env, uint32_t env_size, uint32_t env_mips,
                       const pbr_cluster* clusters, const pbr_light* lights,
                       uint16_t* hdr, uint32_t hdr_pitch, float* hdr_f32_or_null);
/* + per-pixel fp32 conditioning of the colour (see pbr_oracle.cpp): sens = first-order change per unit error of N.H;
   flip = what one 1/256-texel step of the fixed-point sampler can change in the IBL specular term */
int orc_deferred_shade_sens(const pbr_global* g, const pbr_tile* tile, const pbr_gbuffer* gb,
                       const uint16_t* lut, uint32_t lut_res,
                       const uint16_t* env, uint32_t env_size, uint32_t env_mips,
                       const pbr_cluster* clusters, const pbr_light* lights,
                       uint16_t* hdr, uint32_t hdr_pitch, float* hdr_f32_or_null, float* sens_rgb_or_null, float* flip_rgb_or_null); /* a8-a12 */
/* The same pass in DOUBLE precision (pbr_oracle_f64.cpp): the exact value of the reference's formulas on the same inputs, as
   an interval [lo, hi] per channel (3 doubles per pixel each, pitch out_pitch pixels) — lo == hi except where the argument of
   a sampler snap / cube-face choice lies within a few fp32 ulps of its step edge, where both sides are admissible.
   flags (1 byte per pixel): 1 = stencil 0 (not shaded), 2 = octahedral fold decided by rounding, 4 = cluster cell decided by
   rounding (another evaluation may walk another light list): tests skip flagged pixels.  The third party between the GPU
   kernel and the fp32 restatement above: distance of an fp32 colour c to the truth = max(lo - c, c - hi, 0). */
int orc_deferred_shade_f64(const pbr_global* g, const pbr_tile* tile, const pbr_gbuffer* gb,
                       const uint16_t* lut, uint32_t lut_res,
                       const uint16_t* env, uint32_t env_size, uint32_t env_mips,
                       const pbr_cluster* clusters, const pbr_light* lights,
                       double* lo_rgb, double* hi_rgb, uint8_t* flags, uint32_t out_pitch);
/* The split-sum LUT (a3) in double precision: rows [y0, y0 + rows) of the res x res plane as (A, B) pairs of doubles — the
   value the estimator of precompute_brdf.hlsl:20-62 has in exact arithmetic (pbr_oracle_f64.cpp). */
int orc_brdf_lut_f64(uint32_t res, uint32_t y0, uint32_t rows, double* out_ab);
/* The GGX prefilter (a4) of chosen texels in double precision: per channel the interval [lo, hi] of the estimator's exact value
   over the admissible sides of the path's step functions (x.8 snaps of LOD and filter coordinates, cube-face ties) — count x 3
   doubles each (pbr_oracle_f64.cpp). */
int orc_prefilter_env_texels_f64(const float* sky, uint32_t sky_size, uint32_t sky_mips, uint32_t size, uint32_t mips,
                                 uint32_t mip, const uint32_t* texels, uint32_t count, double* out_lo, double* out_hi);
int orc_skybox(const pbr_global* g, const pbr_tile* tile, const float* sky, uint32_t sky_size, uint32_t sky_mips,
               const uint8_t* stencil, uint32_t pitch, uint16_t* hdr, uint32_t hdr_pitch);                  /* 8f-1 */
int orc_gbuffer_encode(const float* m0, const float* m1, const float* m2, uint32_t w, uint32_t h, uint32_t pitch,
                       uint32_t* A, uint32_t* B, uint32_t* C);                                              /* 8f-2 */
int orc_rgbe_decode(const uint8_t* rgbe, size_t texels, float* out_rgba);                                  /* 8f-3 */
int orc_bloom_prefilter(const uint16_t* hdr, uint32_t w, uint32_t h, uint32_t pitch,
                        uint16_t* out, float threshold, float knee);               /* a14 */
int orc_blur_h(const uint16_t* in, uint32_t iw, uint32_t ih, uint16_t* out, uint32_t ow, uint32_t oh);
int orc_blur_v(const uint16_t* in, uint32_t iw, uint32_t ih, uint16_t* out, uint32_t ow, uint32_t oh);
int orc_bloom_upsample_add(const uint16_t* upper, uint32_t uw, uint32_t uh,
                           const uint16_t* lower, uint32_t lw, uint32_t lh, uint16_t* out);
int orc_bloom_merge(uint16_t* hdr, uint32_t pitch, const uint16_t* in, uint32_t w, uint32_t h);
int orc_bloom(uint16_t* hdr, uint32_t w, uint32_t h, uint32_t pitch,
              uint16_t* chain_a, uint16_t* chain_b, float threshold, float knee);  /* a15 */
int orc_lum_histogram(const uint16_t* hdr, uint32_t w, uint32_t h, uint32_t pitch,
                      float min_log, float inv_range, uint32_t* hist256);          /* a16 */
int orc_lum_average(uint32_t* hist256, uint32_t pixel_count, float min_log, float range,
                    float delta_time, float* avg_inout);                           /* a17 */
/* the un-truncated average bin of a17 (diagnostic for tests: distance to an integer) */
float orc_lum_average_bin(const uint32_t* hist256, uint32_t pixel_count);
int orc_tonemap(const uint16_t* hdr, uint32_t w, uint32_t h, uint32_t pitch, const float* avg,
                uint32_t* rgba8, uint32_t out_pitch);                              /* a18 */

#ifdef __cplusplus
}
#endif
#endif
