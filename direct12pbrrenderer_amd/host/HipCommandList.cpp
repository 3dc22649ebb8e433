// HipCommandList.cpp — ShadingState binding checks + the shader-file -> C-ABI dispatch table.
#include "HipCommandList.h"

#include "ShaderConstants.h"

#include <dlfcn.h>

namespace MRendererHip {

// ---------------------------------------------------------------------------------------------
// "Reflection": the resource names each shader file declares (what DXC reflection reports to
// ShadingState::FindShaderAttribute, Engine/Source/Renderer/Pipeline/IPipeline.cpp:188-199).
// Files are the reference's DeferredRendering/Shader/*.hlsl; the kernels are in csrc/*.hip.
static const std::vector<ShaderReflection>& Reflections() {
    static const std::vector<ShaderReflection> table = {
        {"precompute_brdf.hlsl", true, {}, {"PrecomputeBRDF"}, {}, {}, sizeof(PrecomputeBRDFConstant)},
        {"env_map_gen.hlsl", true, {"SkyBox"}, {"PrefilterEnvMap"}, {}, {}, sizeof(PreFilterEnvMapConstant)},
        {"clustered_compute.hlsl", true, {}, {}, {}, {"Clusters"}, sizeof(ClusteredShaderConstant)},
        {"clustered_culling.hlsl", true, {}, {}, {}, {"Clusters", "PointLights"}, sizeof(ClusteredShaderConstant)},
        {"deferred_shading.hlsl", false,
         {"GBufferA", "GBufferB", "GBufferC", "DepthStencil", "PrecomputeBRDF", "PrefilterEnvMap"}, {}, {"Clusters", "PointLights"}, {}, 0},
        {"bloom_prefilter.hlsl", true, {"InputTexture"}, {"OutputTexture"}, {}, {}, sizeof(BloomPrefilterConstant)},
        {"blur_horizontal.hlsl", true, {"InputTexture"}, {"OutputTexture"}, {}, {}, sizeof(BlurConstant)},
        {"blur_vertical.hlsl", true, {"InputTexture"}, {"OutputTexture"}, {}, {}, sizeof(BlurConstant)},
        {"bloom_upsample_add.hlsl", true, {"UpperLevel", "LowerLevel"}, {"OutputTexture"}, {}, {}, sizeof(BlurConstant)},
        {"bloom_merge.hlsl", true, {"InputTexture"}, {"OutputTexture"}, {}, {}, 0},
        {"hdr_luminance_histogram.hlsl", true, {"LuminanceTexture"}, {}, {}, {"LuminanceHistogram"}, sizeof(LuminanceHistogramConstant)},
        {"hdr_average_histogram.hlsl", true, {}, {}, {}, {"LuminanceHistogram", "AverageLuminance"}, sizeof(AverageLuminanceConstant)},
        {"hdr_tone_mapping.hlsl", false, {"LuminanceTexture"}, {}, {}, {"AverageLuminance"}, 0},
        // raster shaders of the reference: out of scope; known so that SetShader succeeds
        {"gbuffer.hlsl", false, {}, {}, {}, {}, 0},
        {"skybox.hlsl", false, {"SkyBox"}, {}, {}, {}, 0},
    };
    return table;
}

const ShaderReflection* FindShader(std::string_view file) {
    for (const auto& r : Reflections())
        if (r.File == file) return &r;
    return nullptr;
}

// --------------------------------------------------------------------------------------------- ShadingState
void ShadingState::SetShader(std::string_view shader_file_path, bool is_compute) {
    mShader = FindShader(shader_file_path);
    if (!mShader) throw HipException("ShadingState::SetShader: no HIP kernel for shader file " + std::string(shader_file_path));
    if (mShader->IsCompute != is_compute) throw HipException("ShadingState::SetShader: compute/graphics mismatch for " + std::string(shader_file_path));
    mIsCompute = is_compute;
    ClearResourceBinding();
}

bool ShadingState::Known(const std::vector<std::string_view>& names, std::string_view semantic_name, const char* kind) const {
    if (mShader && std::find(names.begin(), names.end(), semantic_name) != names.end()) return true;
    // the reference logs and returns false when the name is not in the shader's reflection data
    std::fprintf(stderr, "[ShadingState] %s: no %s named '%.*s'\n", std::string(File()).c_str(), kind, (int)semantic_name.size(), semantic_name.data());
    return false;
}

bool ShadingState::SetTexture(std::string_view name, DeviceTexture* texture) {
    if (!Known(mShader ? mShader->Textures : std::vector<std::string_view>{}, name, "texture")) return false;
    if (mTextures.size() >= MaxShaderResourceViews && !mTextures.count(std::string(name))) return false;
    mTextures[std::string(name)] = TextureBinding{texture, -1};
    return true;
}
bool ShadingState::SetTexture(std::string_view name, DeviceTexture2D* texture, uint32 mip_slice) {
    if (!Known(mShader ? mShader->Textures : std::vector<std::string_view>{}, name, "texture")) return false;
    if (texture && mip_slice >= texture->MipLevels()) throw HipException("ShadingState::SetTexture: mip slice out of range");
    mTextures[std::string(name)] = TextureBinding{texture, (int32)mip_slice};
    return true;
}
bool ShadingState::SetRWTexture(std::string_view name, DeviceTexture2D* texture) {
    if (!Known(mShader ? mShader->RWTextures : std::vector<std::string_view>{}, name, "RW texture")) return false;
    mRWTextures[std::string(name)] = TextureBinding{texture, 0};
    return true;
}
bool ShadingState::SetRWTexture(std::string_view name, DeviceTexture2D* texture, uint32 mip_slice) {
    if (!Known(mShader ? mShader->RWTextures : std::vector<std::string_view>{}, name, "RW texture")) return false;
    if (texture && mip_slice >= texture->MipLevels()) throw HipException("ShadingState::SetRWTexture: mip slice out of range");
    mRWTextures[std::string(name)] = TextureBinding{texture, (int32)mip_slice};
    return true;
}
bool ShadingState::SetRWTextureArray(std::string_view name, DeviceTexture2DArray* texture) {
    if (!Known(mShader ? mShader->RWTextures : std::vector<std::string_view>{}, name, "RW texture array")) return false;
    mRWTextures[std::string(name)] = TextureBinding{texture, -1};
    return true;
}
bool ShadingState::SetStructuredBuffer(std::string_view name, DeviceStructuredBuffer* buffer) {
    if (!Known(mShader ? mShader->StructuredBuffers : std::vector<std::string_view>{}, name, "structured buffer")) return false;
    mBuffers[std::string(name)] = buffer;
    return true;
}
bool ShadingState::SetRWStructuredBuffer(std::string_view name, DeviceStructuredBuffer* buffer) {
    if (!Known(mShader ? mShader->RWStructuredBuffers : std::vector<std::string_view>{}, name, "RW structured buffer")) return false;
    mBuffers[std::string(name)] = buffer;
    return true;
}
void ShadingState::ClearResourceBinding() {
    mTextures.clear();
    mRWTextures.clear();
    mBuffers.clear();
}
const TextureBinding& ShadingState::Texture(std::string_view name) const {
    auto it = mTextures.find(std::string(name));
    if (it == mTextures.end() || !it->second.Texture) throw HipException(std::string(File()) + ": texture '" + std::string(name) + "' is not bound");
    return it->second;
}
const TextureBinding& ShadingState::RWTexture(std::string_view name) const {
    auto it = mRWTextures.find(std::string(name));
    if (it == mRWTextures.end() || !it->second.Texture) throw HipException(std::string(File()) + ": RW texture '" + std::string(name) + "' is not bound");
    return it->second;
}
DeviceStructuredBuffer* ShadingState::Buffer(std::string_view name) const {
    auto it = mBuffers.find(std::string(name));
    if (it == mBuffers.end() || !it->second) throw HipException(std::string(File()) + ": buffer '" + std::string(name) + "' is not bound");
    return it->second;
}

// --------------------------------------------------------------------------------------------- HipCommandList
HipCommandList::HipCommandList(int hip_device) {
    pbr_status st = pbr_ctx_create(hip_device, &mCtx);
    if (st != PBR_OK) throw HipException("pbr_ctx_create failed (status " + std::to_string(st) + ")");
}
HipCommandList::~HipCommandList() {
    (void)pbr_sync(mCtx);
    for (hipEvent_t e : mFrameFence) (void)hipEventDestroy(e);
    pbr_ctx_destroy(mCtx);
}

void HipCommandList::Check(pbr_status st, const char* what) {
    if (st != PBR_OK) throw HipException(std::string(what) + ": " + pbr_last_error(mCtx));
}
void HipCommandList::WaitIdle() {
    FlushPendingBloom();
    EndTail();
    Check(pbr_sync(mCtx), "pbr_sync");   // the context's stream and the side stream's un-joined tail
    if (mTailOverlap) Check(pbr_ctx_side_join(mCtx), "pbr_ctx_side_join");
}
void HipCommandList::SetFramesInFlight(uint32 k) {
    if (k > 8) throw HipException("SetFramesInFlight: at most 8");   // refused BEFORE any state is touched: the renderer keeps its mode
    WaitIdle();
    for (hipEvent_t e : mFrameFence) (void)hipEventDestroy(e);
    mFrameFence.clear();
    mFrameIndex = 0;
    if (k <= 1) {          // the reference's per-frame fence wait: there is no next frame to overlap a tail with
        mTailOverlap = 0;
        return;
    }
    mFrameFence.resize(k, nullptr);
    for (hipEvent_t& e : mFrameFence) ThrowIfFailed(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreate");
}
void HipCommandList::SetTailOverlap(int mode) {
    if (mode < 0 || mode > 2) throw HipException("SetTailOverlap: mode 0 (off), 1 (from the average-luminance dispatch) or 2 (from the bloom pass)");
    if (mode && mFrameFence.empty()) throw HipException("SetTailOverlap: throughput mode only (SetFramesInFlight(k > 1) first): with the per-frame fence there is no next frame to overlap");
    if (mode == 2 && !mHaloPlan.empty()) throw HipException("SetTailOverlap: mode 2 is for frames without a halo exchange");
    WaitIdle();
    mTailOverlap = mode;
}
void HipCommandList::BeginTail() {
    if (!mTailOverlap || mInTail) return;
    Check(pbr_ctx_side_join(mCtx), "pbr_ctx_side_join");     // the tail of the frame BEFORE last is ordered ahead of this point (long finished)
    Check(pbr_ctx_side_begin(mCtx), "pbr_ctx_side_begin");   // the side stream waits for everything enqueued so far (this frame's bloom + histogram)
    mInTail = true;
}
void HipCommandList::EndTail() {
    if (!mInTail) return;
    Check(pbr_ctx_side_end(mCtx), "pbr_ctx_side_end");
    mInTail = false;
}
void HipCommandList::EndFrame() {
    FlushPendingBloom();
    EndTail();
    if (mFrameFence.empty()) {   // D3D12Device::EndFrame: signal + wait, every frame
        Check(pbr_sync(mCtx), "pbr_sync");
        return;
    }
    const size_t k = mFrameFence.size();
    ThrowIfFailed(hipEventRecord(mFrameFence[mFrameIndex % k], (hipStream_t)pbr_ctx_get_stream(mCtx)), "hipEventRecord");
    mFrameIndex++;
    if (mFrameIndex >= k) ThrowIfFailed(hipEventSynchronize(mFrameFence[mFrameIndex % k]), "hipEventSynchronize");   // frame i - k + 1
}

void HipCommandList::SetLayout(const TileLayout& l) {
    // tail mode 2 moves the bloom pass onto the side stream, which a halo exchange in the middle of that pass rules out: SetTailOverlap(2)
    // refuses a halo plan, and a halo layout installed afterwards falls back to mode 1 (average + tone-map only) instead of leaving the
    // combination in place (reachable through the C++ API only: the C ABI fixes the layout at creation)
    if (mTailOverlap == 2 && !l.HaloPlan().empty()) mTailOverlap = 1;
    mLayout = l;
    mTile = pbr_tile{l.Shaded.x, l.Shaded.y, l.Shaded.w, l.Shaded.h, l.FullW, l.FullH};
    mInterior = l.Tiled() ? l.InteriorInShaded() : Rect{};
    mHaloPlan = l.HaloPlan();
    mHaloStaging.reset();
    if (!mHaloPlan.empty()) {
        if (mHaloPlan.size() > 16) throw HipException("SetLayout: more than 16 halo peers");
        const size_t bytes = pbr_halo_staging_bytes(mHaloPlan.data(), (uint32_t)mHaloPlan.size());
        mHaloStaging = std::make_unique<DeviceStructuredBuffer>((uint32)bytes, 8);
    }
}

bool HipCommandList::HaloStripOffset(int rank, bool recv, size_t* offset, size_t* bytes) const {
    size_t off = 0;
    for (int pass = 0; pass < 2; pass++)
        for (const pbr_halo_peer& p : mHaloPlan) {
            const uint32_t* q = pass == 0 ? p.send : p.recv;
            const size_t n = (size_t)q[2] * q[3] * 8;
            if (p.rank == rank && (pass == 1) == recv) {
                if (!n) return false;
                *offset = off;
                *bytes = n;
                return true;
            }
            off += n;
        }
    return false;
}

// roctx (libroctx64.so.4: roctxRangePushA / roctxRangePop), resolved once; a copy that is already mapped — PyTorch
// ships one, rocprofv3 preloads one — is preferred over opening another
namespace {
struct Roctx {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    Roctx() {
        void* h = nullptr;
        const char* names[] = {"libroctx64.so.4", "libroctx64.so", "/opt/rocm/lib/libroctx64.so.4"};
        for (const char* n : names) if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
        if (!h) for (const char* n : names) if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
        if (!h) return;
        push = (int (*)(const char*))dlsym(h, "roctxRangePushA");
        pop = (int (*)())dlsym(h, "roctxRangePop");
        if (!push || !pop) push = nullptr, pop = nullptr;
    }
};
const Roctx& roctx() { static const Roctx r; return r; }
}  // namespace
void HipCommandList::BeginEvent(const char* name) {
    mEventLog.emplace_back(name);
    if (roctx().push) roctx().push(name);
}
void HipCommandList::EndEvent() {
    if (roctx().pop) roctx().pop();
}
void HipCommandList::SetExternalHistogram(const uint32* counts256) {
    if (counts256) mExternalHistogram.assign(counts256, counts256 + 256);
    else mExternalHistogram.clear();
}

namespace {
struct Mip {
    pbr_half* ptr;
    uint32 w, h;
};
Mip MipOf(const TextureBinding& b) {
    auto* t = dynamic_cast<DeviceTexture2D*>(b.Texture);
    if (!t) throw HipException("expected a 2D texture binding");
    uint32 m = b.MipSlice < 0 ? 0 : (uint32)b.MipSlice;
    return Mip{(pbr_half*)t->MipPtr(m), t->Width() >> m, t->Height() >> m};
}
uint32 Groups(uint32 size, uint32 group) { return (size + group - 1) / group; }
void ExpectGroups(std::string_view file, uint32 gx, uint32 gy, uint32 gz, uint32 ex, uint32 ey, uint32 ez) {
    if (gx != ex || gy != ey || gz != ez)
        throw HipException(std::string(file) + ": dispatch shape (" + std::to_string(gx) + "," + std::to_string(gy) + "," + std::to_string(gz) +
                           ") does not cover the bound output (" + std::to_string(ex) + "," + std::to_string(ey) + "," + std::to_string(ez) + ")");
}
}  // namespace

void HipCommandList::Dispatch(ShadingState* s, uint32 gx, uint32 gy, uint32 gz) {
    if (!s || !s->GetShader() || !s->IsCompute()) throw HipException("Dispatch: shading state has no compute shader");
    const std::string_view f = s->File();
    mDispatchCount++;
    if (f != "hdr_luminance_histogram.hlsl") FlushPendingBloom();
    if (f == "precompute_brdf.hlsl") {
        const auto& c = s->Constants<PrecomputeBRDFConstant>();
        auto* out = dynamic_cast<DeviceTexture2D*>(s->RWTexture("PrecomputeBRDF").Texture);
        if (!out || out->Width() != c.TextureResolution || out->Height() != c.TextureResolution) throw HipException("precompute_brdf: LUT size != TextureResolution");
        ExpectGroups(f, gx, gy, gz, Groups(c.TextureResolution, 8), Groups(c.TextureResolution, 8), 1);
        Check(pbr_brdf_lut(mCtx, c.TextureResolution, (pbr_half*)out->DevicePtr()), "pbr_brdf_lut");
    } else if (f == "env_map_gen.hlsl") {
        const auto& c = s->Constants<PreFilterEnvMapConstant>();
        auto* sky = dynamic_cast<DeviceTexture2DArray*>(s->Texture("SkyBox").Texture);
        auto* out = dynamic_cast<DeviceTexture2DArray*>(s->RWTexture("PrefilterEnvMap").Texture);
        if (!sky || !out || out->Size() != c.EnvMapSize || c.MipLevel >= out->MipLevels()) throw HipException("env_map_gen: bad bindings");
        const uint32 ms = c.EnvMapSize >> c.MipLevel;
        ExpectGroups(f, gx, gy, gz, Groups(ms, 8), Groups(ms, 8), 6);   // z = NumCubeMapFaces (quirk Q7: only 6 of the 30 z-threads are useful)
        pbr_cube_f32 cube{(const float*)sky->DevicePtr(), sky->Size(), sky->MipLevels()};
        pbr_half* dst = (pbr_half*)out->DevicePtr() + 4 * pbr_cube_mip_offset(out->Size(), c.MipLevel);
        Check(pbr_prefilter_env_mip(mCtx, &cube, c.EnvMapSize, c.MipLevel, c.Roughness, dst), "pbr_prefilter_env_mip");
        mPaddedEnv.erase(out);   // the padded copy the shade samples is stale now
    } else if (f == "clustered_compute.hlsl") {
        ExpectGroups(f, gx, gy, gz, 1, 1, 1);
        Check(pbr_cluster_build(mCtx, &mGlobal, (pbr_cluster*)s->Buffer("Clusters")->DevicePtr()), "pbr_cluster_build");
    } else if (f == "clustered_culling.hlsl") {
        ExpectGroups(f, gx, gy, gz, 1, 1, 1);
        const auto& c = s->Constants<ClusteredShaderConstant>();
        mNumLights = c.NumLight;
        Check(pbr_cluster_cull(mCtx, &mGlobal, (const pbr_light*)s->Buffer("PointLights")->DevicePtr(), c.NumLight,
                               (pbr_cluster*)s->Buffer("Clusters")->DevicePtr()), "pbr_cluster_cull");
    } else if (f == "bloom_prefilter.hlsl") {
        if (mTailOverlap == 2) BeginTail();   // the first dispatch of BloomPass::Execute issued one by one
        const auto& c = s->Constants<BloomPrefilterConstant>();
        Mip in = MipOf(s->Texture("InputTexture")), out = MipOf(s->RWTexture("OutputTexture"));
        // the reference sizes this grid by the FULL resolution (quirk Q9); the kernel covers the half-res output
        ExpectGroups(f, gx, gy, gz, Groups(in.w, 16), Groups(in.h, 16), 1);
        if (out.w != (in.w >> 1) || out.h != (in.h >> 1)) throw HipException("bloom_prefilter: output is not the half-res mip");
        Check(pbr_bloom_prefilter(mCtx, in.ptr, in.w, in.h, in.w, out.ptr, c.Threshold, c.Knee), "pbr_bloom_prefilter");
    } else if (f == "blur_horizontal.hlsl" || f == "blur_vertical.hlsl") {
        Mip in = MipOf(s->Texture("InputTexture")), out = MipOf(s->RWTexture("OutputTexture"));
        (void)s->Constants<BlurConstant>();   // TexelSize = 1/output size; recomputed identically inside the C ABI
        if (f == "blur_horizontal.hlsl") {
            ExpectGroups(f, gx, gy, gz, Groups(out.w, 256), out.h, 1);
            Check(pbr_blur_h(mCtx, in.ptr, in.w, in.h, out.ptr, out.w, out.h), "pbr_blur_h");
        } else {
            ExpectGroups(f, gx, gy, gz, out.w, Groups(out.h, 256), 1);
            Check(pbr_blur_v(mCtx, in.ptr, in.w, in.h, out.ptr, out.w, out.h), "pbr_blur_v");
        }
    } else if (f == "bloom_upsample_add.hlsl") {
        Mip up = MipOf(s->Texture("UpperLevel")), lo = MipOf(s->Texture("LowerLevel")), out = MipOf(s->RWTexture("OutputTexture"));
        ExpectGroups(f, gx, gy, gz, Groups(out.w, 256), out.h, 1);
        if (out.w != up.w || out.h != up.h) throw HipException("bloom_upsample_add: output size != upper level");
        Check(pbr_bloom_upsample_add(mCtx, up.ptr, up.w, up.h, lo.ptr, lo.w, lo.h, out.ptr), "pbr_bloom_upsample_add");
    } else if (f == "bloom_merge.hlsl") {
        Mip in = MipOf(s->Texture("InputTexture")), out = MipOf(s->RWTexture("OutputTexture"));
        ExpectGroups(f, gx, gy, gz, Groups(out.w, 16), Groups(out.h, 16), 1);
        Check(pbr_bloom_merge(mCtx, out.ptr, out.w, in.ptr, out.w, out.h), "pbr_bloom_merge");
    } else if (f == "hdr_luminance_histogram.hlsl") {
        const auto& c = s->Constants<LuminanceHistogramConstant>();
        Mip in = MipOf(s->Texture("LuminanceTexture"));
        ExpectGroups(f, gx, gy, gz, Groups(c.TextureWidth, 16), Groups(c.TextureHeight, 16), 1);
        uint32_t* hist = (uint32_t*)s->Buffer("LuminanceHistogram")->DevicePtr();
        // multi-GPU: only the interior rectangle of the (apron-extended) target is counted
        const Rect r = mInterior.w ? mInterior : Rect{0, 0, in.w, in.h};
        if (r.x + r.w > in.w || r.y + r.h > in.h || c.TextureWidth != r.w || c.TextureHeight != r.h)
            throw HipException("hdr_luminance_histogram: TextureWidth/Height must be the interior rectangle's size");
        const TextureBinding& lum = s->Texture("LuminanceTexture");
        if (mPendingBloom.What != PendingBloom::None && lum.Texture == mPendingBloom.Hdr && lum.MipSlice <= 0) {
            // the bloom held back by Bloom / BloomHalo wrote this very texture: count the histogram in its last kernel
            FlushPendingBloom(hist, c.MinLogLuminance, c.InvLogLuminanceRange);
        } else {
            FlushPendingBloom();
            Check(pbr_lum_histogram(mCtx, in.ptr + 4 * ((size_t)r.y * in.w + r.x), r.w, r.h, in.w, c.MinLogLuminance, c.InvLogLuminanceRange, hist),
                  "pbr_lum_histogram");
        }
        if (mCaptureHistogram) {
            Check(pbr_sync(mCtx), "pbr_sync");
            mCapturedHistogram.resize(256);
            ThrowIfFailed(hipMemcpy(mCapturedHistogram.data(), hist, 1024, hipMemcpyDeviceToHost), "read histogram");
        }
    } else if (f == "hdr_average_histogram.hlsl") {
        const auto& c = s->Constants<AverageLuminanceConstant>();
        ExpectGroups(f, gx, gy, gz, 1, 1, 1);
        uint32_t* hist = (uint32_t*)s->Buffer("LuminanceHistogram")->DevicePtr();
        BeginTail();   // overlapped frame tail: all-reduce + average (+ the tone-map that follows) on the side stream
        // new step: with several GPUs the tile histograms are summed first; no-op on one GPU
        Check(pbr_allreduce_hist(mCtx, hist), "pbr_allreduce_hist");
        if (!mExternalHistogram.empty()) {   // the same sum where the other tiles' counts arrive through the host
            Check(pbr_sync(mCtx), "pbr_sync");
            std::vector<uint32> mine(256);
            ThrowIfFailed(hipMemcpy(mine.data(), hist, 1024, hipMemcpyDeviceToHost), "read histogram");
            for (int i = 0; i < 256; i++) mine[i] += mExternalHistogram[i];
            ThrowIfFailed(hipMemcpy(hist, mine.data(), 1024, hipMemcpyHostToDevice), "write histogram");
        }
        Check(pbr_lum_average(mCtx, hist, c.PixelCount, c.MinLogLuminance, c.LogLuminanceRange, mGlobal.DeltaTime,
                              (float*)s->Buffer("AverageLuminance")->DevicePtr()), "pbr_lum_average");
    } else {
        throw HipException("Dispatch: " + std::string(f) + " is not a compute kernel of this build");
    }
}

void HipCommandList::DrawScreen(ShadingState* s) {
    if (!s || !s->GetShader() || s->IsCompute()) throw HipException("DrawScreen: shading state has no pixel shader");
    const std::string_view f = s->File();
    mDispatchCount++;
    FlushPendingBloom();
    if (!mRenderTarget) throw HipException("DrawScreen: no render target bound (FrameGraph::PreparePass)");
    if (f == "deferred_shading.hlsl") {
        auto tex = [&](const char* n) { return dynamic_cast<DeviceTexture2D*>(s->Texture(n).Texture); };
        DeviceTexture2D *a = tex("GBufferA"), *b = tex("GBufferB"), *c = tex("GBufferC"), *ds = tex("DepthStencil"), *lut = tex("PrecomputeBRDF");
        auto* env = dynamic_cast<DeviceTexture2DArray*>(s->Texture("PrefilterEnvMap").Texture);
        if (!a || !b || !c || !ds || !lut || !env) throw HipException("deferred_shading: bad texture bindings");
        const uint32 w = mRenderTarget->Width(), h = mRenderTarget->Height();
        pbr_gbuffer gb{(const uint32_t*)a->DevicePtr(), (const uint32_t*)b->DevicePtr(), (const uint32_t*)c->DevicePtr(),
                       ds->DepthPlane(), ds->StencilPlane(), w};
        pbr_tile tile = mTile.w ? mTile : pbr_tile{0, 0, w, h, w, h};
        // the shade samples the padded layout of the env chain (seamless-cube addressing resolved once)
        auto& padded = mPaddedEnv[env];
        if (!padded) {
            padded = std::make_unique<DeviceStructuredBuffer>((uint32)(pbr_env_padded_texels(env->Size(), env->MipLevels()) * 8), 8);
            Check(pbr_env_pad(mCtx, (const pbr_half*)env->DevicePtr(), env->Size(), env->MipLevels(), (pbr_half*)padded->DevicePtr()), "pbr_env_pad");
        }
        // stencil ref 0, compare LESS: shade where 0 < stencil (DeferredPipeline.h:176-181, .cpp:203)
        if (mStencilRef != 0) throw HipException("deferred_shading: only stencil ref 0 is supported");
        Check(pbr_deferred_shade(mCtx, &mGlobal, &tile, &gb, (const pbr_half*)lut->DevicePtr(), lut->Width(),
                                 (const pbr_half*)padded->DevicePtr(), env->Size(), env->MipLevels(),
                                 (const pbr_cluster*)s->Buffer("Clusters")->DevicePtr(), (const pbr_light*)s->Buffer("PointLights")->DevicePtr(),
                                 mNumLights, (pbr_half*)mRenderTarget->DevicePtr(), w), "pbr_deferred_shade");
    } else if (f == "hdr_tone_mapping.hlsl") {
        Mip in = MipOf(s->Texture("LuminanceTexture"));
        // multi-GPU: only the interior rectangle is tone-mapped (the apron belongs to the neighbours)
        const Rect r = mInterior.w ? mInterior : Rect{0, 0, in.w, in.h};
        if (r.x + r.w > in.w || r.y + r.h > in.h || mRenderTarget->Width() < in.w) throw HipException("hdr_tone_mapping: interior rectangle outside the target");
        Check(pbr_tonemap(mCtx, in.ptr + 4 * ((size_t)r.y * in.w + r.x), r.w, r.h, in.w, (const float*)s->Buffer("AverageLuminance")->DevicePtr(),
                          (uint32_t*)mRenderTarget->DevicePtr() + (size_t)r.y * mRenderTarget->Width() + r.x, mRenderTarget->Width()), "pbr_tonemap");
        EndTail();
    } else {
        throw HipException("DrawScreen: " + std::string(f) + " is not a full-screen kernel of this build");
    }
}

void HipCommandList::PrefilterEnv(DeviceTexture2DArray* sky, DeviceTexture2DArray* out) {
    if (!sky || !out) throw HipException("PrefilterEnv: null texture");
    mDispatchCount++;
    FlushPendingBloom();
    pbr_cube_f32 cube{(const float*)sky->DevicePtr(), sky->Size(), sky->MipLevels()};
    Check(pbr_prefilter_env(mCtx, &cube, out->Size(), out->MipLevels(), (pbr_half*)out->DevicePtr()), "pbr_prefilter_env");
    mPaddedEnv.erase(out);   // the padded copy the shade samples is stale now
}

void HipCommandList::Clustered(DeviceStructuredBuffer* clusters, DeviceStructuredBuffer* point_lights, int32 num_lights) {
    if (!clusters || !point_lights) throw HipException("Clustered: null buffer");
    mDispatchCount++;
    FlushPendingBloom();
    mNumLights = num_lights;
    Check(pbr_clustered(mCtx, &mGlobal, (const pbr_light*)point_lights->DevicePtr(), num_lights, (pbr_cluster*)clusters->DevicePtr()), "pbr_clustered");
}

void HipCommandList::Bloom(DeviceTexture2D* hdr, DeviceTexture2D* mip_chain, DeviceTexture2D* temp, float threshold, float knee) {
    if (!hdr || !mip_chain || !temp) throw HipException("Bloom: null texture");
    if (mip_chain->Width() != hdr->Width() || mip_chain->Height() != hdr->Height() || temp->Width() != hdr->Width() || temp->Height() != hdr->Height())
        throw HipException("Bloom: the mip chains must have the HDR target's size");
    mDispatchCount++;
    FlushPendingBloom();
    if (mTailOverlap == 2) BeginTail();
    mPendingBloom = PendingBloom{PendingBloom::Whole, hdr, mip_chain, temp, threshold, knee};
    if (!mFusedPasses) FlushPendingBloom();
}

void HipCommandList::FlushPendingBloom(uint32_t* hist, float min_log, float inv_range) {
    const PendingBloom p = mPendingBloom;
    mPendingBloom = PendingBloom{};
    if (p.What == PendingBloom::Whole) {
        pbr_half *hdr = (pbr_half*)p.Hdr->DevicePtr(), *a = (pbr_half*)p.MipChain->DevicePtr(), *b = (pbr_half*)p.Temp->DevicePtr();
        const uint32 w = p.Hdr->Width(), h = p.Hdr->Height();
        if (hist) {
            const Rect r = mInterior.w ? mInterior : Rect{0, 0, w, h};
            const uint32_t rect[4] = {r.x, r.y, r.w, r.h};
            Check(pbr_bloom_histogram(mCtx, hdr, w, h, w, a, b, p.Threshold, p.Knee, rect, min_log, inv_range, hist), "pbr_bloom_histogram");
        } else {
            Check(pbr_bloom(mCtx, hdr, w, h, w, a, b, p.Threshold, p.Knee), "pbr_bloom");
        }
    } else if (p.What == PendingBloom::Tiled) {
        IssueBloomTiled(p.Hdr, p.MipChain, p.Temp, hist, min_log, inv_range);
    }
}

void HipCommandList::IssueBloomTiled(DeviceTexture2D* hdr, DeviceTexture2D* mip_chain, DeviceTexture2D* temp, uint32_t* hist, float min_log, float inv_range) {
    const PixelRect s = mLayout.ShadedInBloom(), i = mLayout.InteriorInBloom();
    const uint32_t hdr_rect[4] = {s.x, s.y, s.w, s.h}, merge_rect[4] = {i.x, i.y, i.w, i.h};
    Check(pbr_bloom_tiled(mCtx, (pbr_half*)hdr->DevicePtr(), hdr->Width(), hdr_rect, mLayout.Bloom.w, mLayout.Bloom.h, (pbr_half*)mip_chain->DevicePtr(),
                          (pbr_half*)temp->DevicePtr(), merge_rect, min_log, inv_range, hist), "pbr_bloom_tiled");
}

void HipCommandList::HaloExchange(DeviceTexture2D* mip_chain) {
    if (mHaloPlan.empty()) return;
    pbr_half* level1 = (pbr_half*)mip_chain->MipPtr(1);
    const uint32 pitch = mLayout.Bloom.w / 2, rows = mLayout.Bloom.h / 2;
    const uint32_t n = (uint32_t)mHaloPlan.size();
    void* st = mHaloStaging->DevicePtr();
    const size_t bytes = mHaloStaging->Bytes();
    if (mHaloTransport == HaloTransport::Rccl) {
        Check(pbr_halo_exchange(mCtx, level1, pitch, rows, mHaloPlan.data(), n, st, bytes), "pbr_halo_exchange");
    } else {   // the strips are moved between staging areas by the host program (pbrh_halo_copy_from)
        Check(pbr_halo_pack(mCtx, level1, pitch, rows, mHaloPlan.data(), n, st, bytes, 0), "pbr_halo_pack");
        Check(pbr_halo_pack(mCtx, level1, pitch, rows, mHaloPlan.data(), n, st, bytes, 1), "pbr_halo_pack(unpack)");
    }
}

void HipCommandList::BloomHalo(DeviceTexture2D* hdr, DeviceTexture2D* mip_chain, DeviceTexture2D* temp, float threshold, float knee) {
    if (!hdr || !mip_chain || !temp) throw HipException("BloomHalo: null texture");
    if (!mLayout.Halo) throw HipException("BloomHalo: the command list has no halo layout (SetLayout)");
    const TileLayout& l = mLayout;
    if (hdr->Width() != l.Shaded.w || hdr->Height() != l.Shaded.h) throw HipException("BloomHalo: the HDR target must cover the shaded rectangle");
    if (mip_chain->Width() != l.Bloom.w || mip_chain->Height() != l.Bloom.h || temp->Width() != l.Bloom.w || temp->Height() != l.Bloom.h)
        throw HipException("BloomHalo: the mip chains must cover the extended rectangle");
    mDispatchCount++;
    FlushPendingBloom();
    // level-1 texels of the interior, written into the level-1 plane of E
    const PixelRect s = l.ShadedInBloom(), i = l.InteriorInShaded();
    const uint32_t rect[4] = {i.x / 2, i.y / 2, i.w / 2, i.h / 2};
    Check(pbr_bloom_prefilter_rect(mCtx, (pbr_half*)hdr->DevicePtr(), hdr->Width(), hdr->Height(), hdr->Width(), (pbr_half*)mip_chain->MipPtr(1),
                                   l.Bloom.w / 2, s.x / 2, s.y / 2, rect, threshold, knee), "pbr_bloom_prefilter_rect");
    HaloExchange(mip_chain);
    mPendingBloom = PendingBloom{PendingBloom::Tiled, hdr, mip_chain, temp, threshold, knee};
    if (!mFusedPasses) FlushPendingBloom();
}

void HipCommandList::DrawMesh(ShadingState* s) {   // D3D12CommandList.cpp DrawMesh; SkyboxPass::Execute :59-75
    if (!s || s->IsCompute()) throw HipException("DrawMesh: graphics shading state expected");
    const std::string_view f = s->File();
    mDispatchCount++;
    FlushPendingBloom();
    if (f != "skybox.hlsl") throw HipException("DrawMesh: " + std::string(f) + " is a raster shader without a kernel in this build");
    if (!mRenderTarget || !mDepthStencil) throw HipException("DrawMesh: render target / depth-stencil not bound (FrameGraph::PreparePass)");
    auto* sky = dynamic_cast<DeviceTexture2DArray*>(s->Texture("SkyBox").Texture);
    if (!sky) throw HipException("skybox: SkyBox is not bound to a cube texture");
    const uint32 w = mRenderTarget->Width(), h = mRenderTarget->Height();
    pbr_tile tile = mTile.w ? mTile : pbr_tile{0, 0, w, h, w, h};
    pbr_cube_f32 cube{(const float*)sky->DevicePtr(), sky->Size(), sky->MipLevels()};
    Check(pbr_skybox(mCtx, &mGlobal, &tile, &cube, mDepthStencil->StencilPlane(), w, (pbr_half*)mRenderTarget->DevicePtr(), w), "pbr_skybox");
}

void HipCommandList::EncodeGBuffer(ShadingState* s, const float* m0, const float* m1, const float* m2,
                                   DeviceTexture2D* a, DeviceTexture2D* b, DeviceTexture2D* c) {
    if (!s || s->File() != "gbuffer.hlsl") throw HipException("EncodeGBuffer: gbuffer.hlsl shading state expected");
    if (!a || !b || !c || a->Width() != b->Width() || a->Width() != c->Width() || a->Height() != b->Height() || a->Height() != c->Height())
        throw HipException("EncodeGBuffer: G-buffer planes of one size expected");
    mDispatchCount++;
    FlushPendingBloom();
    Check(pbr_gbuffer_encode(mCtx, m0, m1, m2, a->Width(), a->Height(), a->Width(), (uint32_t*)a->DevicePtr(),
                             (uint32_t*)b->DevicePtr(), (uint32_t*)c->DevicePtr()), "pbr_gbuffer_encode");
}

}  // namespace MRendererHip
