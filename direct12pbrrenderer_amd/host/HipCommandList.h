// HipCommandList.h — the dispatch seam.  Stands where D3D12CommandList stands in the reference
// (Engine/Include/Renderer/Device/Direct12/D3D12CommandList.h:83 DrawScreen, :103 Dispatch;
// .cpp:80-97, :116-133): a pass hands over a ShadingState (shader file + named bindings + POD
// constants) and a dispatch shape; here that becomes ONE call into the C ABI of
// include/pbr_hip.h on the context's HIP stream.  Work is recorded serially by one thread and
// executes in order on one stream, which gives the ordering the reference gets from its single
// DIRECT queue (it issues no UAV barriers between dependent dispatches).
#pragma once
#include <string>
#include <vector>

#include "IPipeline.h"
#include "TileLayout.h"

namespace MRendererHip {

class HipCommandList {
public:
    explicit HipCommandList(int hip_device);
    ~HipCommandList();
    HipCommandList(const HipCommandList&) = delete;
    HipCommandList& operator=(const HipCommandList&) = delete;

    void BeginFrame() { mDispatchCount = 0; mEventLog.clear(); }
    // D3D12Device::EndFrame blocks on the fence every frame (D3D12Device.cpp:993-1003): FramesInFlight() == 1, the
    // default.  With k > 1 (throughput mode, not in the reference) EndFrame returns once frame i - k + 1 is done, so
    // the host records frame i + 1 while the GPU still runs frame i; WaitIdle() drains.
    void EndFrame();
    void SetFramesInFlight(uint32 k);
    // Overlapped frame tail (not in the reference; throughput mode only): from the average-luminance dispatch on — histogram
    // all-reduce, average, tone-map — a frame's commands go to the context's high-priority side stream (pbr_ctx_side_*), so that
    // the collective's latency and the two small launches run beside the NEXT frame's cluster pass and shade.  The frame graph
    // must double-buffer what the tail reads (FrameGraph::DoubleBufferResources: HDR target, histogram).
    // mode 1: as above.  mode 2 (frames without a halo exchange): the side stream takes over at the bloom pass already — bloom chain,
    // histogram, average, tone-map beside the next frame's shade (the bloom is HBM / latency-bound, the shade FP32-issue-bound).
    void SetTailOverlap(int mode);
    int TailOverlap() const { return mTailOverlap; }
    uint32 FramesInFlight() const { return (uint32)mFrameFence.size() ? (uint32)mFrameFence.size() : 1; }
    void WaitIdle();

    // global constants (b2), RenderScheduler.cpp:22-41
    void SetGlobalConstant(const ConstantBufferGlobal& g) { mGlobal = g; }
    const ConstantBufferGlobal& GlobalConstant() const { return mGlobal; }

    void SetStencilRef(uint32 ref) { mStencilRef = ref; }
    // bound by FrameGraph::PreparePass for a GraphicsPass (Engine/Source/Renderer/FrameGraph.cpp:94-141)
    void SetRenderTarget(DeviceTexture2D* rt) { mRenderTarget = rt; }
    void SetDepthStencil(DeviceTexture2D* ds) { mDepthStencil = ds; }
    // multi-GPU: the region of the full frame this device renders; default = whole target
    void SetTile(const pbr_tile& tile) { mTile = tile; }
    // multi-GPU, everything at once from a tile layout: the render target is the layout's SHADED rectangle S (global-pixel
    // addressing, interior-only histogram / tone-map); in halo mode also the level-1 exchange plan and its staging area
    void SetLayout(const TileLayout& layout);
    const TileLayout& Layout() const { return mLayout; }
    // how HaloExchange moves the strips: over the context's RCCL communicator (pbr_halo_exchange; needs pbr_comm_init), or
    // not at all (Loopback: pack + unpack only — several tiles rendered one after the other on one device, the strips
    // copied between their staging areas by the host program: pbrh_halo_copy_from)
    enum class HaloTransport { Rccl, Loopback };
    void SetHaloTransport(HaloTransport t) { mHaloTransport = t; }
    const std::vector<pbr_halo_peer>& HaloPlan() const { return mHaloPlan; }
    DeviceStructuredBuffer* HaloStaging() const { return mHaloStaging.get(); }
    // byte offset of the strip sent to / received from `rank` inside the staging area (all send strips in plan order,
    // then all receive strips: the layout pbr_halo_exchange and pbr_halo_pack use); false if there is no such strip
    bool HaloStripOffset(int rank, bool recv, size_t* offset, size_t* bytes) const;
    // multi-GPU: the part of the render target this device OWNS (the rest is the apron it shades only to feed bloom):
    // the luminance histogram counts, and the tone-map writes, interior pixels only.  w == 0: the whole target.
    using Rect = PixelRect;
    void SetInterior(const Rect& r) { mInterior = r; }
    const Rect& Interior() const { return mInterior; }
    // multi-GPU without a communicator (several tiles rendered one after the other on one device: tests, a host that
    // moves the 1 KiB itself): counts of the OTHER tiles, added to this tile's histogram before the average — what
    // pbr_allreduce_hist does over RCCL when pbr_comm_init was called.  nullptr: none.
    void SetExternalHistogram(const uint32* counts256);
    // keep a host copy of the tile's own histogram (before external counts / the all-reduce) of the next frames
    void CaptureHistogram(bool on) { mCaptureHistogram = on; }
    const std::vector<uint32>& CapturedHistogram() const { return mCapturedHistogram; }

    // compute dispatch of `state`'s shader; (x,y,z) are thread-GROUP counts exactly as the reference passes them
    void Dispatch(ShadingState* state, uint32 thread_group_count_x, uint32 thread_group_count_y, uint32 thread_group_count_z);
    // full-screen triangle with `state`'s pixel shader
    void DrawScreen(ShadingState* state);
    // D3D12CommandList::DrawMesh (D3D12CommandList.h:75): the only mesh draw with a kernel in this build is the
    // sky sphere of skybox.hlsl (depth test on, depth write off => exactly the stencil == 0 pixels)
    void DrawMesh(ShadingState* state);
    // gbuffer.hlsl::ps_main on per-pixel material attributes already resolved by the rasterizer (three float4
    // device planes) -> GBufferA/B/C of the bound pass
    void EncodeGBuffer(ShadingState* state, const float* m0, const float* m1, const float* m2,
                       DeviceTexture2D* a, DeviceTexture2D* b, DeviceTexture2D* c);
    void Present(DeviceTexture2D* tex) { mPresented = tex; }

    // Pass-level entry points (not in the reference): a pass whose Execute body is a fixed sequence of dispatches can
    // hand the whole sequence over in one call — fewer launches, intermediates kept on chip; the per-frame passes (Clustered,
    // Bloom) bit for bit the same frame, the one-shot env prefilter within 1 fp16 ULP (below).  Passes use them when
    // FusedPasses() is on; off (default) every reference dispatch is issued one by one.
    void SetFusedPasses(bool on) { mFusedPasses = on; }
    bool FusedPasses() const { return mFusedPasses; }
    // PreFilterEnvMapPass::Execute's five env_map_gen.hlsl dispatches (DeferredPipeline.cpp:97-113) as ONE pbr_prefilter_env:
    // roughness = mip / (mips - 1) for every mip of `out` — what the five constant buffers say; <= 1 fp16 ULP from the
    // dispatch-by-dispatch chain, 6-10 x faster (the per-dispatch kernel keeps the shader's sequential sum)
    void PrefilterEnv(DeviceTexture2DArray* sky, DeviceTexture2DArray* out);
    // ClusteredPass::Execute's two dispatches (pbr_clustered)
    void Clustered(DeviceStructuredBuffer* clusters, DeviceStructuredBuffer* point_lights, int32 num_lights);
    // BloomPass::Execute's sixteen dispatches (pbr_bloom); mip_chain / temp are scratch afterwards.  With fused passes the
    // call is held back until the next command: when that is the luminance histogram of the same texture, both become ONE
    // pbr_bloom_histogram (the histogram is counted by the bloom's last kernel: one full read of the HDR target saved) —
    // a command list is free to merge adjacent commands as long as every consumer sees the same results.
    void Bloom(DeviceTexture2D* hdr, DeviceTexture2D* mip_chain, DeviceTexture2D* temp, float threshold, float knee);
    // Halo mode (SURVEY 8e option 2) — what BloomPass::Execute issues on a tile whose bloom pyramid runs on the extended
    // rectangle E: level-1 texels of the interior (pbr_bloom_prefilter_rect) -> the rest of E's level 1 from the neighbours
    // (HaloExchange) -> levels 1..4 on E and the merge into the interior (pbr_bloom_tiled; held back like Bloom when fused).
    // hdr covers S; mip_chain / temp are E-sized.
    void BloomHalo(DeviceTexture2D* hdr, DeviceTexture2D* mip_chain, DeviceTexture2D* temp, float threshold, float knee);

    // Named ranges around pass bodies, where the reference has PIXScopedEvent (DeferredPipeline.cpp:8 `PIXScope`): roctx
    // ranges here, visible to rocprofv3 --marker-trace.  libroctx64 is looked up at run time; absent = no-op.
    void BeginEvent(const char* name);
    void EndEvent();
    const std::vector<std::string>& EventLog() const { return mEventLog; }   // names pushed since BeginFrame (tests)

    pbr_ctx* Context() const { return mCtx; }
    uint32 DispatchCount() const { return mDispatchCount; }
    DeviceTexture2D* Presented() const { return mPresented; }
    // the light count the last clustered_culling dispatch was given (the HIP shade stages exactly that many records)
    int32 NumLights() const { return mNumLights; }

private:
    void Check(pbr_status st, const char* what);
    void HaloExchange(DeviceTexture2D* mip_chain);
    // the bloom call held back by Bloom / BloomHalo: issued with the histogram (hist != nullptr) or on its own
    struct PendingBloom {
        enum Kind { None, Whole, Tiled } What = None;
        DeviceTexture2D *Hdr = nullptr, *MipChain = nullptr, *Temp = nullptr;
        float Threshold = 0, Knee = 0;
    };
    void FlushPendingBloom(uint32_t* hist = nullptr, float min_log = 0.0f, float inv_range = 0.0f);
    void IssueBloomTiled(DeviceTexture2D* hdr, DeviceTexture2D* mip_chain, DeviceTexture2D* temp, uint32_t* hist, float min_log, float inv_range);

    pbr_ctx* mCtx = nullptr;
    ConstantBufferGlobal mGlobal{};
    uint32 mStencilRef = 0;
    uint32 mDispatchCount = 0;
    int32 mNumLights = 0;
    DeviceTexture2D* mPresented = nullptr;
    DeviceTexture2D* mRenderTarget = nullptr;
    DeviceTexture2D* mDepthStencil = nullptr;
    bool mFusedPasses = false;
    pbr_tile mTile{};
    Rect mInterior{};
    TileLayout mLayout{};
    HaloTransport mHaloTransport = HaloTransport::Rccl;
    std::vector<pbr_halo_peer> mHaloPlan;
    std::unique_ptr<DeviceStructuredBuffer> mHaloStaging;
    PendingBloom mPendingBloom;
    std::vector<hipEvent_t> mFrameFence;   // ring of per-frame completion events (throughput mode)
    uint64_t mFrameIndex = 0;
    int mTailOverlap = 0;
    bool mInTail = false;
    void BeginTail();
    void EndTail();
    std::vector<uint32> mExternalHistogram, mCapturedHistogram;
    bool mCaptureHistogram = false;
    std::vector<std::string> mEventLog;
    // padded copies of prefiltered env chains (pbr_env_pad), keyed by the plain texture; rebuilt after
    // env_map_gen.hlsl rewrites the texture
    std::map<const DeviceTexture2DArray*, std::unique_ptr<DeviceStructuredBuffer>> mPaddedEnv;
};

// RAII twin of the reference's PIXScope(cmd, name) macro (DeferredPipeline.cpp:8)
class PixScope {
public:
    PixScope(HipCommandList* cmd, const char* name) : mCmd(cmd) { mCmd->BeginEvent(name); }
    ~PixScope() { mCmd->EndEvent(); }
    PixScope(const PixScope&) = delete;
    PixScope& operator=(const PixScope&) = delete;
private:
    HipCommandList* mCmd;
};
#define PBR_PIX_CAT2(a, b) a##b
#define PBR_PIX_CAT(a, b) PBR_PIX_CAT2(a, b)
#define PIXScope(cmd, name) ::MRendererHip::PixScope PBR_PIX_CAT(pix_scope_, __LINE__)((cmd), (name))

}  // namespace MRendererHip
