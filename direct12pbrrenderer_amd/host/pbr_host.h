/* pbr_host.h — C entry points of libpbr_host.so: a small embedding of the C++ pass graph
 * (DeferredRenderPipeline + FrameGraph + RenderScheduler over the HIP kernels) so that tests
 * and tools can drive whole frames through the reference-shaped pass API. */
#ifndef PBR_HOST_H
#define PBR_HOST_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef struct pbrh_renderer pbrh_renderer;

pbrh_renderer* pbrh_create(int hip_device, uint32_t width, uint32_t height, uint32_t env_size, uint32_t lut_res,
                           char* err, size_t err_len);
/* ---- multi-GPU, one renderer per device: this device's tile of a full_w x full_h frame cut cols x rows
 * (rank = row * cols + col; BASELINE cfg5: 7680 x 4320, cols 4, rows 2).  Every target covers the tile's SHADED rectangle;
 * uv / camera ray / ClusterIndex use global pixels, the histogram counts and the tone-map writes the interior only, the
 * average divides by the full frame's pixel count.  halo = 0 (apron mode): the device shades interior + 256 px and blooms
 * that.  halo = 1: it shades interior + 4 px; BloomPass::Execute computes the interior's half-res level, receives the rest
 * of the extended rectangle's half-res level from the neighbouring devices (pbr_halo_exchange over the communicator of
 * pbrh_comm_init) and runs the pyramid on the extended rectangle (pbr_bloom_tiled). */
pbrh_renderer* pbrh_create_tile(int hip_device, uint32_t full_w, uint32_t full_h, uint32_t cols, uint32_t rows, uint32_t rank, int halo,
                                uint32_t env_size, uint32_t lut_res, char* err, size_t err_len);
/* CPU only: the layout arithmetic behind pbrh_create_tile.  rects = interior, shaded, bloom rectangle (x, y, w, h each, in
 * global pixels); peers = the halo plan, 9 ints per peer: rank, send x y w h, recv x y w h (half-res texels local to the bloom
 * rectangle's half-res plane).  Returns the number of peers, -1 on a bad grid. */
int pbrh_tile_layout(uint32_t full_w, uint32_t full_h, uint32_t cols, uint32_t rows, uint32_t rank, int halo, uint32_t rects[12], int32_t* peers, int max_peers);
/* halo mode without a communicator (several tiles rendered one after the other on ONE device): on = the exchange step only
 * packs the outgoing strips into the renderer's staging area and unpacks whatever the incoming part of it holds;
 * pbrh_halo_copy_from(dst, src) copies the strip src sends to dst into dst's staging area (device to device). */
int pbrh_set_halo_loopback(pbrh_renderer* r, int on);
int pbrh_halo_copy_from(pbrh_renderer* dst, pbrh_renderer* src);
/* 1 (default) = the reference's frame loop: every frame ends with the fence wait (D3D12Device.cpp:993-1003).  k > 1 =
 * throughput mode: a frame's end waits for frame i - k + 1 only, so the host records ahead of the GPU. */
int pbrh_set_frames_in_flight(pbrh_renderer* r, int k);
/* throughput mode only (after pbrh_set_frames_in_flight(k > 1)): a frame's tail — histogram all-reduce, average, tone-map — on
 * the context's high-priority side stream, beside the next frame's cluster pass and shade; the HDR target and the histogram are
 * double-buffered.  Same frames as the plain order (pbrh_read waits for everything in flight).
 * on = 1: as described.  on = 2 (frames without a halo exchange: one GPU, apron mode): the side stream takes over at the bloom pass —
 * bloom chain, histogram, average, tone-map run beside the next frame's cluster pass and shade.  0: off. */
int pbrh_set_tail_overlap(pbrh_renderer* r, int on);
void pbrh_destroy(pbrh_renderer* r);
const char* pbrh_last_error(const pbrh_renderer* r);
/* fp32 RGBA cube mip 0 (host, 6*size*size*4 floats): uploaded, box mips + SH9 computed on the GPU */
int pbrh_set_skybox(pbrh_renderer* r, const float* cube_mip0, uint32_t size);
/* LoadCubeMap: <dir>/{px,nx,py,ny,pz,nz}.hdr (Radiance RGBE) -> sky cube + mips + SH9 on the GPU */
int pbrh_load_skybox(pbrh_renderer* r, const char* dir);
/* CPU only: parse one .hdr file held in memory (header + flat / run-length scanlines) into RGBE texels */
int pbrh_parse_hdr(const uint8_t* file, size_t bytes, uint32_t* w, uint32_t* h, uint8_t* rgbe, size_t rgbe_bytes, char* err, size_t err_len);
/* n lights: position[3], color[3], radius, intensity (8 floats each) */
int pbrh_set_lights(pbrh_renderer* r, const float* lights, int n);
/* the "mSceneLight" records of a reference scene file (Asset/Scene/main.json: Scene.h:192, ReflectionDef.h:119-149) replace
 * the renderer's lights, in file order (Scene::PostDeserialized, Scene.cpp:83-99) */
int pbrh_load_scene_lights(pbrh_renderer* r, const char* scene_json_path);
/* CPU only: the same records of a scene file held in memory as 8-float records (as pbrh_set_lights takes them); returns the
 * count (may exceed max_lights), -1 + reason in err on malformed input */
int pbrh_parse_scene_lights(const char* json, size_t bytes, float* lights, int max_lights, char* err, size_t err_len);
int pbrh_set_gbuffer(pbrh_renderer* r, const uint32_t* A, const uint32_t* B, const uint32_t* C, const float* depth, const uint8_t* stencil);
/* alternative to pbrh_set_gbuffer: the rasterizer's per-pixel material attributes (three float4 planes, see
 * pbr_gbuffer_encode in pbr_hip.h); GBufferPass encodes them on the GPU */
int pbrh_set_materials(pbrh_renderer* r, const float* m0, const float* m1, const float* m2, const float* depth, const uint8_t* stencil);
int pbrh_set_initial_luminance(pbrh_renderer* r, float v);
/* ---- multi-GPU: this renderer's target is the apron-extended tile at (x0, y0) of a full_w x full_h frame;
 * it OWNS the interior rectangle (ix, iy, iw, ih) of its target.  uv / camera ray / ClusterIndex use global pixels, the
 * camera's aspect ratio is the full frame's, the histogram counts and the tone-map writes the interior only, and the
 * average divides by the full frame's pixel count. */
int pbrh_set_tile(pbrh_renderer* r, uint32_t x0, uint32_t y0, uint32_t full_w, uint32_t full_h,
                  uint32_t ix, uint32_t iy, uint32_t iw, uint32_t ih);
/* one process per GPU: RCCL communicator of the context (unique id from pbr_comm_unique_id on rank 0); the average
 * pass then all-reduces the 256-bin histogram (pbr_allreduce_hist) */
int pbrh_comm_init(pbrh_renderer* r, int world, int rank, const void* unique_id_128_bytes);
/* no communicator (tiles rendered one after the other on one device, or a host that moves the 1 KiB itself): the
 * other tiles' counts, added before the average (NULL: none); and a host copy of this tile's own counts */
int pbrh_set_external_histogram(pbrh_renderer* r, const uint32_t* counts256);
int pbrh_capture_histogram(pbrh_renderer* r, int on);
int pbrh_captured_histogram(pbrh_renderer* r, uint32_t* dst256);
/* on: ClusteredPass, BloomPass and the one-shot PreFilterEnvMapPass hand their fixed dispatch sequences over as one call each
 * (pbr_clustered, pbr_bloom[_histogram], pbr_prefilter_env); off (default): every reference dispatch is issued one by one.
 * The per-frame passes give the same frame bit for bit either way; the fused env chain is within 1 fp16 ULP of the five
 * dispatches (tests/test_host_graph.py).  May be switched between frames. */
int pbrh_set_fused(pbrh_renderer* r, int on);
/* n frames; *ms_per_frame = average wall time per frame (every frame ends with the per-frame fence wait) */
int pbrh_render_n(pbrh_renderer* r, int n, float delta_time, double* ms_per_frame);
/* one frame through RenderScheduler::ExecutePipeline; blocks until the GPU is done */
int pbrh_render(pbrh_renderer* r, float delta_time);
/* "PreFilterEnvMap>PrecomputeBRDF>..." */
int pbrh_execution_order(pbrh_renderer* r, char* buf, size_t len);
int pbrh_dispatch_count(const pbrh_renderer* r);
/* the named ranges (the reference's PIXScope strings; roctx ranges here) the last frame opened, '>'-separated, in order */
int pbrh_event_log(const pbrh_renderer* r, char* buf, size_t len);
/* copy a frame-graph resource (by its FGResourceIDs name) to host memory; returns bytes copied or <0 */
long pbrh_read(pbrh_renderer* r, const char* resource_name, void* dst, size_t dst_bytes);
/* the global constants the last frame used (412 bytes) */
int pbrh_get_global(const pbrh_renderer* r, void* dst_412_bytes);
/* builds DeferredRenderPipeline + FrameGraph without touching a GPU and returns the sorted pass order */
/* CPU only (no device): which of the n lights (8 floats each, as pbrh_set_lights) Scene::CullLight hands to the
 * light buffer for the reference camera at cam_pos_yaw = (x, y, z, yaw), and in what order; returns the count
 * (may exceed max_indices), -1 if a light's culling bound leaves the world box */
int pbrh_cull_lights(uint32_t width, uint32_t height, const float cam_pos_yaw[4], const float* lights, int n, int* indices, int max_indices);
/* CPU only: the PointLight[] (pbr_light, 44 bytes each) ClusteredPass::Execute commits for these lights and that camera
 * (DeferredPipeline.cpp:224-241): cull membership + order, attenuation presets (Scene.cpp:132-165); count, or -1 */
int pbrh_light_buffer(uint32_t width, uint32_t height, const float cam_pos_yaw[4], const float* lights, int n, void* out_pbr_lights, int capacity);
/* CPU only: the scene file's lights through Scene::PostDeserialized (AddSceneLights) and Scene::CullLight for the reference default
 * camera moved / rotated as given: bounds6[i] = world AABB {min xyz, max xyz} of light i as the cull sees it (SceneObject::GetWorldBound:
 * the object's matrix — FromEulerAngle(mRotation in degrees), SetScale, translation — applied to the two corners of the local cube,
 * MathLib.cpp:5-10), visible[k] = indices in visiting order.  Returns the number of lights (<= max_lights filled), *n_visible the number
 * visited; -1 + reason on malformed input. */
int pbrh_scene_light_bounds(uint32_t width, uint32_t height, const float cam_pos_yaw[4], const char* json, size_t bytes,
                            float* bounds6, int max_lights, int* visible, int* n_visible, char* err, size_t err_len);
int pbrh_dry_run_execution_order(uint32_t width, uint32_t height, char* buf, size_t len);
/* ShadingState contract probes (no GPU work): 1 = the call returned true */
int pbrh_probe_binding(const char* shader_file, int is_compute, const char* semantic_name, int kind);
#ifdef __cplusplus
}
#endif
#endif
