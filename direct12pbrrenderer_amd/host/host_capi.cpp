// host_capi.cpp — C embedding of the pass graph (see pbr_host.h).
#include "pbr_host.h"
#include "HdrImage.h"
#include "SceneFile.h"

#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>

#include "DeferredPipeline.h"

using namespace MRendererHip;

struct pbrh_renderer {
    std::unique_ptr<DeferredRenderPipeline> pipeline;
    std::unique_ptr<RenderScheduler> scheduler;
    std::unique_ptr<Scene> scene;
    std::unique_ptr<Camera> camera;
    uint32 width = 0, height = 0;
    float time = 0.0f;
    std::string err;
};

template <class Fn>
static int guarded(pbrh_renderer* r, Fn&& fn) {
    try {
        fn();
        return 0;
    } catch (const std::exception& e) {
        if (r) r->err = e.what();
        return -1;
    }
}

extern "C" {

pbrh_renderer* pbrh_create(int device, uint32_t width, uint32_t height, uint32_t env_size, uint32_t lut_res, char* err, size_t err_len) {
    auto* r = new pbrh_renderer();
    try {
        FGResourceDescriptionTable::Instance()->Reset();   // one pipeline per process in the reference; allow re-creation here
        r->width = width;
        r->height = height;
        r->pipeline = std::make_unique<DeferredRenderPipeline>(RenderSize{width, height}, env_size, lut_res);
        r->scheduler = std::make_unique<RenderScheduler>(r->pipeline.get(), device, width, height);
        r->scene = std::make_unique<Scene>();
        // App.cpp:99-101
        r->camera = std::make_unique<Camera>(0.333f * 3.14159265359f, width, height, 0.1f, 1000.0f);
        r->camera->Move(Vector3{0, 3, 10});
        r->camera->Rotate(0, 3.14159265359f, 0);
        return r;
    } catch (const std::exception& e) {
        if (err && err_len) std::snprintf(err, err_len, "%s", e.what());
        delete r;
        return nullptr;
    }
}

// One device's tile of a full_w x full_h frame cut cols x rows (rank = row * cols + col): the targets cover the shaded
// rectangle, the bloom chains the extended one (TileLayout.h).  halo = 0: apron mode.
pbrh_renderer* pbrh_create_tile(int device, uint32_t full_w, uint32_t full_h, uint32_t cols, uint32_t rows, uint32_t rank, int halo,
                                uint32_t env_size, uint32_t lut_res, char* err, size_t err_len) {
    auto* r = new pbrh_renderer();
    try {
        const TileLayout layout = TileLayout::OfFrame(full_w, full_h, cols, rows, rank, halo != 0);
        FGResourceDescriptionTable::Instance()->Reset();
        r->width = layout.Shaded.w;
        r->height = layout.Shaded.h;
        r->pipeline = std::make_unique<DeferredRenderPipeline>(layout, env_size, lut_res);
        r->scheduler = std::make_unique<RenderScheduler>(r->pipeline.get(), device, r->width, r->height);
        r->scene = std::make_unique<Scene>();
        r->scheduler->CommandList()->SetLayout(layout);
        if (layout.Tiled()) r->pipeline->mAutoExposurePass->SetFullFramePixelCount(full_w * full_h);
        // the camera sees the FULL frame (App.cpp:99-101 with the frame's aspect ratio)
        r->camera = std::make_unique<Camera>(0.333f * 3.14159265359f, full_w, full_h, 0.1f, 1000.0f);
        r->camera->Move(Vector3{0, 3, 10});
        r->camera->Rotate(0, 3.14159265359f, 0);
        return r;
    } catch (const std::exception& e) {
        if (err && err_len) std::snprintf(err, err_len, "%s", e.what());
        delete r;
        return nullptr;
    }
}

// CPU only: the layout arithmetic of pbrh_create_tile.  rects = interior, shaded, bloom rectangle (x, y, w, h each, global
// pixels); peers = the halo plan (rank, send x y w h, recv x y w h: 9 ints per peer, level-1 texels local to the bloom
// rectangle's level-1 plane).  Returns the number of peers (may exceed max_peers), -1 on a bad grid.
int pbrh_tile_layout(uint32_t full_w, uint32_t full_h, uint32_t cols, uint32_t rows, uint32_t rank, int halo, uint32_t rects[12], int32_t* peers, int max_peers) {
    try {
        const TileLayout l = TileLayout::OfFrame(full_w, full_h, cols, rows, rank, halo != 0);
        const PixelRect* src[3] = {&l.Interior, &l.Shaded, &l.Bloom};
        for (int k = 0; k < 3 && rects; k++) {
            rects[4 * k] = src[k]->x; rects[4 * k + 1] = src[k]->y; rects[4 * k + 2] = src[k]->w; rects[4 * k + 3] = src[k]->h;
        }
        const std::vector<pbr_halo_peer> plan = l.HaloPlan();
        for (size_t i = 0; i < plan.size() && (int)i < max_peers && peers; i++) {
            int32_t* q = peers + 9 * i;
            q[0] = plan[i].rank;
            for (int k = 0; k < 4; k++) { q[1 + k] = (int32_t)plan[i].send[k]; q[5 + k] = (int32_t)plan[i].recv[k]; }
        }
        return (int)plan.size();
    } catch (const std::exception&) {
        return -1;
    }
}

void pbrh_destroy(pbrh_renderer* r) { delete r; }
const char* pbrh_last_error(const pbrh_renderer* r) { return r ? r->err.c_str() : "null renderer"; }

int pbrh_set_skybox(pbrh_renderer* r, const float* cube, uint32_t size) {
    return guarded(r, [&] {
        uint32 mips = 1;
        while ((size >> mips) >= 1) mips++;
        auto sky = std::make_shared<SkyBox>();
        sky->Cube = std::make_shared<DeviceTexture2DArray>(size, mips, ETextureFormat_R32G32B32A32_FLOAT);
        ThrowIfFailed(hipMemcpy(sky->Cube->DevicePtr(), cube, (size_t)6 * size * size * 16, hipMemcpyHostToDevice), "upload sky");
        pbr_ctx* ctx = r->scheduler->CommandList()->Context();
        if (pbr_cube_gen_mips(ctx, (float*)sky->Cube->DevicePtr(), size, mips) != PBR_OK) throw HipException(pbr_last_error(ctx));
        DeviceStructuredBuffer pack(112, 4);
        pbr_cube_f32 c{(const float*)sky->Cube->DevicePtr(), size, mips};
        if (pbr_sh9_project(ctx, &c, (float*)pack.DevicePtr()) != PBR_OK) throw HipException(pbr_last_error(ctx));
        if (pbr_sync(ctx) != PBR_OK) throw HipException(pbr_last_error(ctx));
        ThrowIfFailed(hipMemcpy(&sky->SH, pack.DevicePtr(), 112, hipMemcpyDeviceToHost), "read SH");
        r->scene->SetSkyBox(sky);
        r->pipeline->mPrefilterEnvMapPass->Invalidate();
    });
}

int pbrh_load_skybox(pbrh_renderer* r, const char* dir) {
    return guarded(r, [&] {
        r->scene->SetSkyBox(LoadCubeMap(r->scheduler->CommandList()->Context(), dir));
        r->pipeline->mPrefilterEnvMapPass->Invalidate();
    });
}

// CPU only: parse one .hdr file held in memory; *w, *h and (when rgbe != NULL and rgbe_bytes suffices) the expanded
// RGBE texels.  Returns 0, or -1 with the reason in err.
int pbrh_parse_hdr(const uint8_t* file, size_t bytes, uint32_t* w, uint32_t* h, uint8_t* rgbe, size_t rgbe_bytes, char* err, size_t err_len) {
    try {
        HdrImage img = ParseRadianceHDR(file, bytes);
        if (w) *w = img.Width;
        if (h) *h = img.Height;
        if (rgbe) {
            if (rgbe_bytes < img.Rgbe.size()) throw HipException("hdr: output buffer too small");
            std::memcpy(rgbe, img.Rgbe.data(), img.Rgbe.size());
        }
        return 0;
    } catch (const std::exception& e) {
        if (err && err_len) std::snprintf(err, err_len, "%s", e.what());
        return -1;
    }
}

int pbrh_set_lights(pbrh_renderer* r, const float* l, int n) {
    return guarded(r, [&] {
        r->scene->ClearLights();
        for (int i = 0; i < n; i++, l += 8) r->scene->AddLight(SceneLight(Vector3{l[0], l[1], l[2]}, Vector3{l[3], l[4], l[5]}, l[6], l[7]));
    });
}

// Scene::PostDeserialized for the light list of a reference scene file (Asset/Scene/main.json): SceneFile.h
int pbrh_load_scene_lights(pbrh_renderer* r, const char* scene_json_path) {
    return guarded(r, [&] { AddSceneLights(r->scene.get(), LoadSceneLights(scene_json_path)); });
}

// CPU only: the "mSceneLight" records of a scene file held in memory, as the 8-float records pbrh_set_lights takes
// (translation, colour, radius, intensity), in file order.  Returns the count (may exceed max_lights), -1 + reason on
// malformed input or on a light whose object carries a rotation / scale (those go through pbrh_load_scene_lights).
int pbrh_parse_scene_lights(const char* json, size_t bytes, float* lights, int max_lights, char* err, size_t err_len) {
    try {
        const std::vector<SceneLightRecord> recs = ParseSceneLights(json, bytes);
        for (size_t i = 0; i < recs.size(); i++) {
            const SceneLightRecord& q = recs[i];
            if (q.Rotation.x != 0 || q.Rotation.y != 0 || q.Rotation.z != 0 || q.Scale.x != 1 || q.Scale.y != 1 || q.Scale.z != 1)
                throw HipException("scene json: a light object with a rotation or scale has no 8-float record");
            if ((int)i < max_lights && lights) {
                float* o = lights + 8 * i;
                o[0] = q.Translation.x; o[1] = q.Translation.y; o[2] = q.Translation.z;
                o[3] = q.Color.x; o[4] = q.Color.y; o[5] = q.Color.z;
                o[6] = q.Radius; o[7] = q.Intensity;
            }
        }
        return (int)recs.size();
    } catch (const std::exception& e) {
        if (err && err_len) std::snprintf(err, err_len, "%s", e.what());
        return -1;
    }
}

int pbrh_scene_light_bounds(uint32_t width, uint32_t height, const float cam_pos_yaw[4], const char* json, size_t bytes,
                            float* bounds6, int max_lights, int* visible, int* n_visible, char* err, size_t err_len) {
    try {
        Scene scene;
        AddSceneLights(&scene, ParseSceneLights(json, bytes));
        const int n = (int)scene.GetLightCount();
        for (int i = 0; i < n && i < max_lights && bounds6; i++) {
            const AABB b = scene.LightAt((size_t)i)->GetWorldBound();
            float* o = bounds6 + 6 * i;
            o[0] = b.Min.x; o[1] = b.Min.y; o[2] = b.Min.z; o[3] = b.Max.x; o[4] = b.Max.y; o[5] = b.Max.z;
        }
        if (n_visible) {
            Camera camera(0.333f * 3.14159265359f, width, height, 0.1f, 1000.0f);
            camera.Move(Vector3{cam_pos_yaw[0], cam_pos_yaw[1], cam_pos_yaw[2]});
            camera.Rotate(0, cam_pos_yaw[3], 0);
            const Matrix4x4 vp = camera.GetProjectionMatrix() * camera.GetLocalSpaceMatrix();
            int count = 0;
            scene.CullLight(FrustumVolume::FromMatrix(vp.m), [&](SceneLight* light) {
                if (visible && count < max_lights) visible[count] = (int)(light - scene.LightAt(0));
                count++;
            });
            *n_visible = count;
        }
        return n;
    } catch (const std::exception& e) {
        if (err && err_len) std::snprintf(err, err_len, "%s", e.what());
        return -1;
    }
}

int pbrh_set_gbuffer(pbrh_renderer* r, const uint32_t* A, const uint32_t* B, const uint32_t* C, const float* depth, const uint8_t* stencil) {
    return guarded(r, [&] {
        GBufferSource& g = r->scene->GBuffer();
        const size_t n = (size_t)r->width * r->height;
        g.Width = r->width;
        g.Height = r->height;
        g.M0.clear(); g.M1.clear(); g.M2.clear();
        g.Dirty = true;
        g.A.assign(A, A + n);
        g.B.assign(B, B + n);
        g.C.assign(C, C + n);
        g.Depth.assign(depth, depth + n);
        g.Stencil.assign(stencil, stencil + n);
    });
}

int pbrh_set_materials(pbrh_renderer* r, const float* m0, const float* m1, const float* m2, const float* depth, const uint8_t* stencil) {
    return guarded(r, [&] {
        GBufferSource& g = r->scene->GBuffer();
        const size_t n = (size_t)r->width * r->height;
        g.Width = r->width;
        g.Height = r->height;
        g.A.clear(); g.B.clear(); g.C.clear();
        g.Dirty = true;
        g.M0.assign(m0, m0 + 4 * n);
        g.M1.assign(m1, m1 + 4 * n);
        g.M2.assign(m2, m2 + 4 * n);
        g.Depth.assign(depth, depth + n);
        g.Stencil.assign(stencil, stencil + n);
    });
}

int pbrh_set_initial_luminance(pbrh_renderer* r, float v) {
    return guarded(r, [&] { r->pipeline->mAutoExposurePass->SetInitialLuminance(v); });
}

int pbrh_set_tile(pbrh_renderer* r, uint32_t x0, uint32_t y0, uint32_t full_w, uint32_t full_h,
                  uint32_t ix, uint32_t iy, uint32_t iw, uint32_t ih) {
    return guarded(r, [&] {
        if (x0 + r->width > full_w || y0 + r->height > full_h) throw HipException("pbrh_set_tile: the target does not fit the frame");
        if (iw == 0 || ih == 0 || ix + iw > r->width || iy + ih > r->height) throw HipException("pbrh_set_tile: interior outside the target");
        HipCommandList* cmd = r->scheduler->CommandList();
        cmd->SetTile(pbr_tile{x0, y0, r->width, r->height, full_w, full_h});
        cmd->SetInterior(HipCommandList::Rect{ix, iy, iw, ih});
        r->pipeline->mAutoExposurePass->SetFullFramePixelCount(full_w * full_h);
        // the camera sees the FULL frame (App.cpp:99-101 with the frame's aspect ratio)
        r->camera = std::make_unique<Camera>(0.333f * 3.14159265359f, full_w, full_h, 0.1f, 1000.0f);
        r->camera->Move(Vector3{0, 3, 10});
        r->camera->Rotate(0, 3.14159265359f, 0);
    });
}

int pbrh_comm_init(pbrh_renderer* r, int world, int rank, const void* uid) {
    return guarded(r, [&] {
        pbr_ctx* ctx = r->scheduler->CommandList()->Context();
        if (pbr_comm_init(ctx, world, rank, uid) != PBR_OK) throw HipException(pbr_last_error(ctx));
    });
}

int pbrh_set_halo_loopback(pbrh_renderer* r, int on) {
    return guarded(r, [&] {
        r->scheduler->CommandList()->SetHaloTransport(on ? HipCommandList::HaloTransport::Loopback : HipCommandList::HaloTransport::Rccl);
    });
}

// dst's receive strip from src's rank <- src's send strip to dst's rank (device-to-device; both renderers idle)
int pbrh_halo_copy_from(pbrh_renderer* dst, pbrh_renderer* src) {
    return guarded(dst, [&] {
        if (!src) throw HipException("pbrh_halo_copy_from: null source");
        HipCommandList *d = dst->scheduler->CommandList(), *s = src->scheduler->CommandList();
        d->WaitIdle();
        s->WaitIdle();
        size_t doff = 0, dbytes = 0, soff = 0, sbytes = 0;
        const bool want = d->HaloStripOffset((int)s->Layout().Rank, true, &doff, &dbytes);
        const bool have = s->HaloStripOffset((int)d->Layout().Rank, false, &soff, &sbytes);
        if (want != have || dbytes != sbytes) throw HipException("pbrh_halo_copy_from: the two layouts do not agree on the strip");
        if (!want) return;   // not neighbours
        ThrowIfFailed(hipMemcpy((char*)d->HaloStaging()->DevicePtr() + doff, (const char*)s->HaloStaging()->DevicePtr() + soff, dbytes, hipMemcpyDeviceToDevice),
                      "copy halo strip");
    });
}

int pbrh_set_frames_in_flight(pbrh_renderer* r, int k) {
    return guarded(r, [&] { r->scheduler->CommandList()->SetFramesInFlight(k < 1 ? 1u : (uint32)k); });
}

int pbrh_set_tail_overlap(pbrh_renderer* r, int on) {
    return guarded(r, [&] {
        HipCommandList* cmd = r->scheduler->CommandList();
        cmd->SetTailOverlap(on);
        if (on) r->scheduler->GetFrameGraph()->DoubleBufferResources({DeferredPipelineResource::DeferredShadingRT, DeferredPipelineResource::LuminanceHistogram});
    });
}

int pbrh_set_external_histogram(pbrh_renderer* r, const uint32_t* counts256) {
    return guarded(r, [&] { r->scheduler->CommandList()->SetExternalHistogram(counts256); });
}
int pbrh_capture_histogram(pbrh_renderer* r, int on) {
    return guarded(r, [&] { r->scheduler->CommandList()->CaptureHistogram(on != 0); });
}
int pbrh_captured_histogram(pbrh_renderer* r, uint32_t* dst256) {
    return guarded(r, [&] {
        const auto& h = r->scheduler->CommandList()->CapturedHistogram();
        if (h.size() != 256) throw HipException("pbrh_captured_histogram: no frame was rendered with capture on");
        std::memcpy(dst256, h.data(), 1024);
    });
}

int pbrh_render(pbrh_renderer* r, float dt) {
    return guarded(r, [&] {
        r->time += dt;
        r->scheduler->ExecutePipeline(r->scene.get(), r->camera.get(), dt, r->time);
    });
}

int pbrh_set_fused(pbrh_renderer* r, int on) {
    return guarded(r, [&] { r->scheduler->CommandList()->SetFusedPasses(on != 0); });
}

// n frames through RenderScheduler::ExecutePipeline (each ends with the reference's per-frame fence wait); average
// wall time per frame in milliseconds
int pbrh_render_n(pbrh_renderer* r, int n, float dt, double* ms_per_frame) {
    return guarded(r, [&] {
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < n; i++) {
            r->time += dt;
            r->scheduler->ExecutePipeline(r->scene.get(), r->camera.get(), dt, r->time);
        }
        r->scheduler->CommandList()->WaitIdle();   // throughput mode: the last frames are still in flight
        const auto t1 = std::chrono::steady_clock::now();
        if (ms_per_frame) *ms_per_frame = std::chrono::duration<double, std::milli>(t1 - t0).count() / (n > 0 ? n : 1);
    });
}

int pbrh_execution_order(pbrh_renderer* r, char* buf, size_t len) {
    return guarded(r, [&] {
        std::string s;
        for (IRenderPass* p : r->scheduler->GetFrameGraph()->ExecutionOrder()) {
            if (!s.empty()) s += ">";
            s += p->Name();
        }
        std::snprintf(buf, len, "%s", s.c_str());
    });
}

int pbrh_event_log(const pbrh_renderer* r, char* buf, size_t len) {
    if (!r || !buf || !len) return -1;
    std::string s;
    for (const std::string& e : r->scheduler->CommandList()->EventLog()) {
        if (!s.empty()) s += ">";
        s += e;
    }
    std::snprintf(buf, len, "%s", s.c_str());
    return 0;
}

int pbrh_dispatch_count(const pbrh_renderer* r) { return r ? (int)r->scheduler->CommandList()->DispatchCount() : -1; }

long pbrh_read(pbrh_renderer* r, const char* name, void* dst, size_t dst_bytes) {
    long n = -1;
    int st = guarded(r, [&] {
        r->scheduler->CommandList()->WaitIdle();
        IDeviceResource* res = r->scheduler->GetFrameGraph()->FindResource(FGResourceIDs::Instance()->NameToID(name));
        size_t bytes = std::min(dst_bytes, res->Bytes());
        ThrowIfFailed(hipMemcpy(dst, res->DevicePtr(), bytes, hipMemcpyDeviceToHost), "read back");
        n = (long)bytes;
    });
    return st ? -1 : n;
}

int pbrh_get_global(const pbrh_renderer* r, void* dst) {
    if (!r || !dst) return -1;
    std::memcpy(dst, &r->scheduler->CommandList()->GlobalConstant(), sizeof(pbr_global));
    return 0;
}

// CPU-only: the reference default camera (App.cpp:99-101) moved/rotated as given, a Scene with the n lights
// added in order, and the light indices Scene::CullLight visits, in visiting order.
int pbrh_cull_lights(uint32_t width, uint32_t height, const float cam_pos_yaw[4], const float* l, int n, int* indices, int max_indices) {
    try {
        Camera camera(0.333f * 3.14159265359f, width, height, 0.1f, 1000.0f);
        camera.Move(Vector3{cam_pos_yaw[0], cam_pos_yaw[1], cam_pos_yaw[2]});
        camera.Rotate(0, cam_pos_yaw[3], 0);
        Scene scene;
        for (int i = 0; i < n; i++, l += 8) scene.AddLight(SceneLight(Vector3{l[0], l[1], l[2]}, Vector3{l[3], l[4], l[5]}, l[6], l[7]));
        const Matrix4x4 vp = camera.GetProjectionMatrix() * camera.GetLocalSpaceMatrix();
        int count = 0;
        scene.CullLight(FrustumVolume::FromMatrix(vp.m), [&](SceneLight* light) {
            if (count < max_indices) indices[count] = (int)(light - scene.LightAt(0));
            count++;
        });
        return count;
    } catch (const std::exception&) {
        return -1;
    }
}

// CPU only: the PointLight[] records (44 bytes each, pbr_light) ClusteredPass::Execute would commit for these lights and this
// camera — Scene::CullLight's membership and order, SceneLight::CaclAttenuationCoefficients' preset.  Returns the count, -1 on error.
int pbrh_light_buffer(uint32_t width, uint32_t height, const float cam_pos_yaw[4], const float* l, int n, void* out_pbr_lights, int capacity) {
    try {
        Camera camera(0.333f * 3.14159265359f, width, height, 0.1f, 1000.0f);
        camera.Move(Vector3{cam_pos_yaw[0], cam_pos_yaw[1], cam_pos_yaw[2]});
        camera.Rotate(0, cam_pos_yaw[3], 0);
        Scene scene;
        for (int i = 0; i < n; i++, l += 8) scene.AddLight(SceneLight(Vector3{l[0], l[1], l[2]}, Vector3{l[3], l[4], l[5]}, l[6], l[7]));
        return FillLightBuffer(&scene, &camera, (pbr_light*)out_pbr_lights, capacity);
    } catch (const std::exception&) {
        return -1;
    }
}

int pbrh_dry_run_execution_order(uint32_t width, uint32_t height, char* buf, size_t len) {
    try {
        DeviceMemory::DryRun() = true;
        FGResourceDescriptionTable::Instance()->Reset();
        DeferredRenderPipeline pipeline(RenderSize{width, height});
        FrameGraph graph(&pipeline);
        graph.Setup();
        graph.Compile();
        std::string s;
        for (IRenderPass* p : graph.ExecutionOrder()) {
            if (!s.empty()) s += ">";
            s += p->Name();
        }
        std::snprintf(buf, len, "%s", s.c_str());
        DeviceMemory::DryRun() = false;
        return 0;
    } catch (const std::exception& e) {
        DeviceMemory::DryRun() = false;
        std::snprintf(buf, len, "error: %s", e.what());
        return -1;
    }
}

int pbrh_probe_binding(const char* shader_file, int is_compute, const char* name, int kind) {
    try {
        ShadingState s;
        s.SetShader(shader_file, is_compute != 0);
        switch (kind) {
            case 0: return s.SetTexture(name, (DeviceTexture*)nullptr) ? 1 : 0;
            case 1: return s.SetRWTexture(name, (DeviceTexture2D*)nullptr) ? 1 : 0;
            case 2: return s.SetStructuredBuffer(name, nullptr) ? 1 : 0;
            default: return s.SetRWStructuredBuffer(name, nullptr) ? 1 : 0;
        }
    } catch (const std::exception&) {
        return -1;
    }
}

}  // extern "C"
