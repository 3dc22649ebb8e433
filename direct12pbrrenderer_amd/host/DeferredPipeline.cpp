// DeferredPipeline.cpp — Execute bodies: bind by shader resource name, set the POD constants,
// dispatch with the reference's group counts (Engine/Source/Renderer/Pipeline/DeferredPipeline.cpp).
#include "DeferredPipeline.h"

namespace MRendererHip {

static inline uint32 CalculateDispatchSize(uint32 texture_size, uint32 thread_group_size) {
    return (texture_size + thread_group_size - 1) / thread_group_size;
}
template <class T>
static T* As(IDeviceResource* r) {
    T* t = dynamic_cast<T*>(r);
    if (!t) throw HipException("frame-graph resource has an unexpected type");
    return t;
}

std::vector<IRenderPass*> DeferredRenderPipeline::Setup() {
    // construction order matters exactly as in the reference: BloomPass reads DeferredShadingRT's
    // description from the table, so DeferredShadingPass must exist first (DeferredPipeline.cpp:24-33)
    mPrefilterEnvMapPass = std::make_unique<PreFilterEnvMapPass>(mEnvSize);
    mPrecomputeBRDFPass = std::make_unique<PrecomputeBRDFPass>(mLutRes);
    mGBufferPass = std::make_unique<GBufferPass>(mSize);
    mDeferredShadingPass = std::make_unique<DeferredShadingPass>(mSize);
    mSkyboxPass = std::make_unique<SkyboxPass>();
    mAutoExposurePass = std::make_unique<AutoExposurePass>();
    mToneMappingPass = std::make_unique<ToneMappingPass>(mSize);
    mPresentPass = std::make_unique<PresentPass>();
    mBloomPass = std::make_unique<BloomPass>(mHaloChain);
    mClusteredPass = std::make_unique<ClusteredPass>();
    mPresentPass->SetFinalTexture(DeferredPipelineResource::ToneMappedTexture);
    return {mPrefilterEnvMapPass.get(), mPrecomputeBRDFPass.get(), mClusteredPass.get(), mGBufferPass.get(),
            mDeferredShadingPass.get(), mSkyboxPass.get(), mAutoExposurePass.get(), mToneMappingPass.get(), mBloomPass.get(), mPresentPass.get()};
}

// ----------------------------------------------------------------------------------- IBL precompute
PreFilterEnvMapPass::PreFilterEnvMapPass(uint32 size) : mSize(size), mReady(false) {
    if ((size >> (PreFilterEnvMapMipsLevel - 1)) < 1)   // the coarsest mip must still hold a texel
        throw HipException("PreFilterEnvMapPass: size too small for 5 mips");
    mPrefilterEnvMap = std::make_shared<DeviceTexture2DArray>(size, PreFilterEnvMapMipsLevel, ETextureFormat_R16G16B16A16_FLOAT);
    WritePersistentResource(DeferredPipelineResource::PrefilterEnvMap, mPrefilterEnvMap.get());
}

void PreFilterEnvMapPass::Execute(FGContext* context) {   // DeferredPipeline.cpp:77-115
    if (mReady) return;
    mReady = true;
    SkyBox* sky = context->Scene->GetSkyBox();
    if (!sky) return;
    PIXScope(context->CommandList, "Precompute PrefilterEnvMap Pass");
    if (context->CommandList->FusedPasses()) {   // the five dispatches below as one call (same roughness ladder: mip / 4)
        auto* cube = dynamic_cast<DeviceTexture2DArray*>(sky->Resource());
        if (!cube) throw HipException("PreFilterEnvMapPass: the sky box is not a cube texture");
        context->CommandList->PrefilterEnv(cube, mPrefilterEnvMap.get());
        return;
    }
    for (uint32 i = 0; i < PreFilterEnvMapMipsLevel; i++) {
        ShadingState& st = mShadingState[i];
        st.SetShader("env_map_gen.hlsl", true);
        st.SetRWTextureArray("PrefilterEnvMap", mPrefilterEnvMap.get());
        st.SetTexture("SkyBox", sky->Resource());
        st.SetConstantBuffer(PreFilterEnvMapConstant{(float)i / (float)(PreFilterEnvMapMipsLevel - 1), i, mSize});
    }
    for (uint32 i = 0; i < PreFilterEnvMapMipsLevel; i++) {
        const uint32 mip_size = mSize >> i;
        const uint32 groups = (mip_size + DispatchGroupSize - 1) / DispatchGroupSize;
        context->CommandList->Dispatch(&mShadingState[i], groups, groups, 6);
    }
}

PrecomputeBRDFPass::PrecomputeBRDFPass(uint32 res) : mRes(res), mReady(false) {
    mPrecomputeBRDF = std::make_shared<DeviceTexture2D>(res, res, 1, ETextureFormat_R16G16_FLOAT);
    WritePersistentResource(DeferredPipelineResource::PrecomputeBRDF, mPrecomputeBRDF.get());
    mShadingState.SetShader("precompute_brdf.hlsl", true);
    mShadingState.SetRWTexture("PrecomputeBRDF", mPrecomputeBRDF.get());
}

void PrecomputeBRDFPass::Execute(FGContext* context) {   // :117-136
    if (mReady) return;
    PIXScope(context->CommandList, "Precompute BRDF Pass");
    mReady = true;
    constexpr uint32 ThreadGroupSize = 8;
    mShadingState.SetConstantBuffer(PrecomputeBRDFConstant{mRes});
    context->CommandList->Dispatch(&mShadingState, (mRes + ThreadGroupSize - 1) / ThreadGroupSize, (mRes + ThreadGroupSize - 1) / ThreadGroupSize, 1);
}

// ----------------------------------------------------------------------------------- G-buffer (input)
GBufferPass::GBufferPass(RenderSize s) {
    WriteTransientTexture(DeferredPipelineResource::GBufferA, s.Width, s.Height, 1, ETextureFormat_R8G8B8A8_UNORM);
    WriteTransientTexture(DeferredPipelineResource::GBufferB, s.Width, s.Height, 1, ETextureFormat_R8G8B8A8_UNORM);
    WriteTransientTexture(DeferredPipelineResource::GBufferC, s.Width, s.Height, 1, ETextureFormat_R8G8B8A8_UNORM);
    WriteTransientTexture(DeferredPipelineResource::DepthStencil, s.Width, s.Height, 1, ETextureFormat_DepthStencil, ETexture2DFlag_AllowDepthStencil);
    mShadingState.SetShader("gbuffer.hlsl", false);
}

void GBufferPass::Execute(FGContext* context) {
    PIXScope(context->CommandList, "Gbuffer Pass");
    GBufferSource& src = context->Scene->GBuffer();
    if (!src.Dirty) return;   // the planes a rasterizer would have left in device memory are still there
    src.Dirty = false;
    if (context->CommandList->FramesInFlight() > 1) context->CommandList->WaitIdle();   // earlier frames still read the planes
    auto* a = As<DeviceTexture2D>(GetTransientResource(context, DeferredPipelineResource::GBufferA));
    if (src.Width != a->Width() || src.Height != a->Height()) throw HipException("GBufferPass: G-buffer source size != render size");
    const size_t n = (size_t)src.Width * src.Height;
    auto* b = As<DeviceTexture2D>(GetTransientResource(context, DeferredPipelineResource::GBufferB));
    auto* c = As<DeviceTexture2D>(GetTransientResource(context, DeferredPipelineResource::GBufferC));
    auto* ds = As<DeviceTexture2D>(GetTransientResource(context, DeferredPipelineResource::DepthStencil));
    if (src.HasMaterials()) {   // gbuffer.hlsl::ps_main on the rasterizer's per-pixel attributes
        if (!mMaterialPlanes || mMaterialPlanes->Size() != n * 48)
            mMaterialPlanes = std::make_unique<DeviceStructuredBuffer>((uint32)(n * 48), 16);
        float* m = (float*)mMaterialPlanes->DevicePtr();
        ThrowIfFailed(hipMemcpy(m, src.M0.data(), n * 16, hipMemcpyHostToDevice), "upload material plane 0");
        ThrowIfFailed(hipMemcpy(m + 4 * n, src.M1.data(), n * 16, hipMemcpyHostToDevice), "upload material plane 1");
        ThrowIfFailed(hipMemcpy(m + 8 * n, src.M2.data(), n * 16, hipMemcpyHostToDevice), "upload material plane 2");
        context->CommandList->EncodeGBuffer(&mShadingState, m, m + 4 * n, m + 8 * n, a, b, c);
    } else {
        ThrowIfFailed(hipMemcpy(a->DevicePtr(), src.A.data(), n * 4, hipMemcpyHostToDevice), "upload GBufferA");
        ThrowIfFailed(hipMemcpy(b->DevicePtr(), src.B.data(), n * 4, hipMemcpyHostToDevice), "upload GBufferB");
        ThrowIfFailed(hipMemcpy(c->DevicePtr(), src.C.data(), n * 4, hipMemcpyHostToDevice), "upload GBufferC");
    }
    ThrowIfFailed(hipMemcpy(ds->DepthPlane(), src.Depth.data(), n * 4, hipMemcpyHostToDevice), "upload depth");
    ThrowIfFailed(hipMemcpy(ds->StencilPlane(), src.Stencil.data(), n, hipMemcpyHostToDevice), "upload stencil");
}

SkyboxPass::SkyboxPass() {   // DeferredPipeline.cpp:46-57
    mShadingState.SetShader("skybox.hlsl", false);
    WriteResource(DeferredPipelineResource::DeferredShadingRT);
    WriteResource(DeferredPipelineResource::DepthStencil);
}
void SkyboxPass::Execute(FGContext* context) {   // DeferredPipeline.cpp:59-75
    SkyBox* sky_box = context->Scene->GetSkyBox();
    if (!sky_box) return;
    PIXScope(context->CommandList, "Skybox Pass");
    mShadingState.SetTexture("SkyBox", sky_box->Resource());
    context->CommandList->DrawMesh(&mShadingState);   // sky sphere, depth test on / write off
}

// ----------------------------------------------------------------------------------- deferred shading
DeferredShadingPass::DeferredShadingPass(RenderSize s) {   // DeferredPipeline.h:157-182
    ReadResource(DeferredPipelineResource::GBufferA);
    ReadResource(DeferredPipelineResource::GBufferB);
    ReadResource(DeferredPipelineResource::GBufferC);
    ReadResource(DeferredPipelineResource::DepthStencil);
    ReadResource(DeferredPipelineResource::PrefilterEnvMap);
    ReadResource(DeferredPipelineResource::PrecomputeBRDF);
    ReadResource(DeferredPipelineResource::PointLights);
    ReadResource(DeferredPipelineResource::FrustumCluster);
    WriteTransientTexture(DeferredPipelineResource::DeferredShadingRT, s.Width, s.Height, 1, ETextureFormat_R16G16B16A16_FLOAT,
                          (ETexture2DFlag)(ETexture2DFlag_AllowRenderTarget | ETexture2DFlag_AllowUnorderedAccess));
    WriteResource(DeferredPipelineResource::DepthStencil);   // stencil test only
    mShadingState.SetShader("deferred_shading.hlsl", false);
}

void DeferredShadingPass::Execute(FGContext* context) {   // DeferredPipeline.cpp:187-206
    PIXScope(context->CommandList, "Deferred Shading");
    auto tex = [&](FGResourceId id) { return As<DeviceTexture>(GetTransientResource(context, id)); };
    mShadingState.SetTexture("GBufferA", tex(DeferredPipelineResource::GBufferA));
    mShadingState.SetTexture("GBufferB", tex(DeferredPipelineResource::GBufferB));
    mShadingState.SetTexture("GBufferC", tex(DeferredPipelineResource::GBufferC));
    mShadingState.SetTexture("PrefilterEnvMap", tex(DeferredPipelineResource::PrefilterEnvMap));
    mShadingState.SetTexture("PrecomputeBRDF", tex(DeferredPipelineResource::PrecomputeBRDF));
    mShadingState.SetTexture("DepthStencil", tex(DeferredPipelineResource::DepthStencil));
    mShadingState.SetStructuredBuffer("Clusters", As<DeviceStructuredBuffer>(GetTransientResource(context, DeferredPipelineResource::FrustumCluster)));
    mShadingState.SetStructuredBuffer("PointLights", As<DeviceStructuredBuffer>(GetTransientResource(context, DeferredPipelineResource::PointLights)));
    context->CommandList->SetStencilRef(0);
    context->CommandList->DrawScreen(&mShadingState);
}

// ----------------------------------------------------------------------------------- clustered lights
ClusteredPass::ClusteredPass() {
    WriteTransientBuffer(DeferredPipelineResource::FrustumCluster, ClusterSizeX * ClusterSizeY * ClusterSizeZ * (uint32)sizeof(pbr_cluster), sizeof(pbr_cluster));
    WriteTransientBuffer(DeferredPipelineResource::PointLights, MaxSceneLights * (uint32)sizeof(pbr_light), sizeof(pbr_light));
    mClusteredCompute.SetShader("clustered_compute.hlsl", true);
    mClusteredCulling.SetShader("clustered_culling.hlsl", true);
}

void ClusteredPass::Execute(FGContext* context) {   // DeferredPipeline.cpp:208-258
    PIXScope(context->CommandList, "Clustered Pass");
    auto* sw_cluster = As<DeviceStructuredBuffer>(GetTransientResource(context, DeferredPipelineResource::FrustumCluster));
    auto* sw_point_light = As<DeviceStructuredBuffer>(GetTransientResource(context, DeferredPipelineResource::PointLights));
    mClusteredCompute.SetRWStructuredBuffer("Clusters", sw_cluster);
    mClusteredCulling.SetRWStructuredBuffer("Clusters", sw_cluster);
    mClusteredCulling.SetRWStructuredBuffer("PointLights", sw_point_light);
    if (context->Scene->GetLightCount() > (uint32)MaxSceneLights) throw HipException("ClusteredPass: more than MaxSceneLights lights");
    std::vector<pbr_light>& lights = mLights;
    lights.assign(MaxSceneLights, pbr_light{});
    // frustum culling point lights (DeferredPipeline.cpp:224-241): membership and buffer order come from the octree walk
    const int i = FillLightBuffer(context->Scene, context->Camera, lights.data(), MaxSceneLights);
    mClusteredCompute.SetConstantBuffer(ClusteredShaderConstant{i});
    mClusteredCulling.SetConstantBuffer(ClusteredShaderConstant{i});
    // the reference re-uploads the buffer every frame through its upload ring; here the device copy is rewritten only when
    // the culled list changed (a static camera and scene: never again) — and then only once no earlier frame that reads
    // it can still be in flight (throughput mode keeps several)
    if (i != mCommittedCount || mCommitted.size() != lights.size() || std::memcmp(mCommitted.data(), lights.data(), (size_t)i * sizeof(pbr_light)) != 0) {
        if (context->CommandList->FramesInFlight() > 1) context->CommandList->WaitIdle();
        sw_point_light->Commit(lights.data(), lights.size() * sizeof(pbr_light));
        mCommitted = lights;
        mCommittedCount = i;
    }
    if (context->CommandList->FusedPasses()) {
        context->CommandList->Clustered(sw_cluster, sw_point_light, i);
        return;
    }
    context->CommandList->Dispatch(&mClusteredCompute, 1, 1, 1);
    context->CommandList->Dispatch(&mClusteredCulling, 1, 1, 1);
}

// ----------------------------------------------------------------------------------- auto exposure
AutoExposurePass::AutoExposurePass() : mAvarageLuminanceInitialized(false) {
    ReadResource(DeferredPipelineResource::DeferredShadingRT);
    WriteTransientBuffer(DeferredPipelineResource::LuminanceHistogram, HistogramBinSize * (uint32)sizeof(uint32), sizeof(uint32));
    WriteTransientBuffer(DeferredPipelineResource::AverageLuminance, 1 * (uint32)sizeof(float), sizeof(float));
    mLuminanceHistogramCompute.SetShader("hdr_luminance_histogram.hlsl", true);
    mAvarageLuminanceCompute.SetShader("hdr_average_histogram.hlsl", true);
}

void AutoExposurePass::Execute(FGContext* context) {   // DeferredPipeline.cpp:260-318
    PIXScope(context->CommandList, "Auto Exposure Pass");
    auto* input_tex = As<DeviceTexture2D>(GetTransientResource(context, DeferredPipelineResource::DeferredShadingRT));
    auto* histogram = As<DeviceStructuredBuffer>(GetTransientResource(context, DeferredPipelineResource::LuminanceHistogram));
    auto* avg_luminance = As<DeviceStructuredBuffer>(GetTransientResource(context, DeferredPipelineResource::AverageLuminance));
    if (!mAvarageLuminanceInitialized) {
        mAvarageLuminanceInitialized = true;
        if (context->CommandList->FramesInFlight() > 1) context->CommandList->WaitIdle();
        avg_luminance->Commit(&mInitialLuminance, sizeof(float));
    }
    {
    PIXScope(context->CommandList, "Luminance Histogram Pass");
    mLuminanceHistogramCompute.SetRWStructuredBuffer("LuminanceHistogram", histogram);
    mLuminanceHistogramCompute.SetTexture("LuminanceTexture", input_tex);
    // multi-GPU: a device counts the interior of its tile only; one GPU: the whole texture, as the reference
    const HipCommandList::Rect& in = context->CommandList->Interior();
    const uint32 hw = in.w ? in.w : input_tex->Width(), hh = in.w ? in.h : input_tex->Height();
    mLuminanceHistogramCompute.SetConstantBuffer(LuminanceHistogramConstant{hw, hh, MinLogLuminance, InvLogLuminanceRange});
    context->CommandList->Dispatch(&mLuminanceHistogramCompute, CalculateDispatchSize(hw, HistogramComputeThreadGroupSize),
                                   CalculateDispatchSize(hh, HistogramComputeThreadGroupSize), 1);
    }
    PIXScope(context->CommandList, "Average Luminance Pass");
    mAvarageLuminanceCompute.SetRWStructuredBuffer("LuminanceHistogram", histogram);
    mAvarageLuminanceCompute.SetRWStructuredBuffer("AverageLuminance", avg_luminance);
    const uint32 pixels = mFullFramePixels ? mFullFramePixels : input_tex->Width() * input_tex->Height();
    mAvarageLuminanceCompute.SetConstantBuffer(AverageLuminanceConstant{pixels, MinLogLuminance, LogLuminanceRange});
    context->CommandList->Dispatch(&mAvarageLuminanceCompute, 1, 1, 1);
}

// ----------------------------------------------------------------------------------- tone mapping
ToneMappingPass::ToneMappingPass(RenderSize s) {
    ReadResource(DeferredPipelineResource::DeferredShadingRT);
    ReadResource(DeferredPipelineResource::AverageLuminance);
    WriteTransientTexture(DeferredPipelineResource::ToneMappedTexture, s.Width, s.Height, 1, ETextureFormat_R8G8B8A8_UNORM);
    mToneMappingRender.SetShader("hdr_tone_mapping.hlsl", false);
}

void ToneMappingPass::Execute(FGContext* context) {   // DeferredPipeline.cpp:320-336
    PIXScope(context->CommandList, "Tone Mapping Pass");
    auto* input_tex = As<DeviceTexture2D>(GetTransientResource(context, DeferredPipelineResource::DeferredShadingRT));
    auto* avg_luminance = As<DeviceStructuredBuffer>(GetTransientResource(context, DeferredPipelineResource::AverageLuminance));
    mToneMappingRender.SetRWStructuredBuffer("AverageLuminance", avg_luminance);
    mToneMappingRender.SetTexture("LuminanceTexture", input_tex);
    context->CommandList->DrawScreen(&mToneMappingRender);
}

// ----------------------------------------------------------------------------------- bloom
BloomPass::BloomPass(RenderSize halo_chain) : mHalo(halo_chain.Width != 0) {   // DeferredPipeline.cpp:338-374
    const FGTransientTextureDescription& d = FGResourceDescriptionTable::Instance()->GetTransientTexture(DeferredPipelineResource::DeferredShadingRT);
    const uint32 cw = mHalo ? halo_chain.Width : d.Width, ch = mHalo ? halo_chain.Height : d.Height;
    if ((cw >> (MipmapLevel - 1)) == 0 || (ch >> (MipmapLevel - 1)) == 0) throw HipException("BloomPass: render size too small for the mip chain");
    WriteTransientTexture(DeferredPipelineResource::BloomMipchain, cw, ch, MipmapLevel, d.Format, ETexture2DFlag_AllowUnorderedAccess);
    WriteTransientTexture(DeferredPipelineResource::BloomTempTexture, cw, ch, MipmapLevel, d.Format, ETexture2DFlag_AllowUnorderedAccess);
    WriteResource(DeferredPipelineResource::DeferredShadingRT);
    mPrefilter.SetShader("bloom_prefilter.hlsl", true);
    mUpsampleBlurH.SetShader("blur_horizontal.hlsl", true);
    mUpsampleBlurV.SetShader("blur_vertical.hlsl", true);
    mUpsampleMerge.SetShader("bloom_merge.hlsl", true);
    for (auto& s : mDownsampleH) s.SetShader("blur_horizontal.hlsl", true);
    for (auto& s : mDownsampleV) s.SetShader("blur_vertical.hlsl", true);
    for (auto& s : mUpsampleH) s.SetShader("bloom_upsample_add.hlsl", true);
    for (auto& s : mUpsampleV) s.SetShader("blur_vertical.hlsl", true);
}

void BloomPass::Execute(FGContext* context) {   // DeferredPipeline.cpp:400-570, 16 dispatches
    PIXScope(context->CommandList, "Bloom Pass");
    auto* original_tex = As<DeviceTexture2D>(GetTransientResource(context, DeferredPipelineResource::DeferredShadingRT));
    auto* mip_chain = As<DeviceTexture2D>(GetTransientResource(context, DeferredPipelineResource::BloomMipchain));
    auto* temp_tex = As<DeviceTexture2D>(GetTransientResource(context, DeferredPipelineResource::BloomTempTexture));
    HipCommandList* cmd = context->CommandList;
    auto texel = [](uint32 w, uint32 h) { return Vector2{1.0f / (float)w, 1.0f / (float)h}; };
    if (mHalo) {
        // multi-GPU halo mode (SURVEY 8e option 2; not in the reference): the pyramid runs on the tile's extended rectangle,
        // whose level 1 outside the interior arrives from the neighbouring devices — prefilter the interior, exchange,
        // levels 1..4 + merge.  Bit-identical to the sixteen dispatches below on the whole frame in the interior.
        cmd->BloomHalo(original_tex, mip_chain, temp_tex, 1.0f, 0.5f);
        return;
    }
    if (cmd->FusedPasses()) {   // the sixteen dispatches below as one call: same HDR result
        cmd->Bloom(original_tex, mip_chain, temp_tex, 1.0f, 0.5f);
        return;
    }

    {
    PIXScope(cmd, "Bloom Prefilter");
    mPrefilter.SetConstantBuffer(BloomPrefilterConstant{texel(original_tex->Width() >> 1, original_tex->Height() >> 1), 1.0f, 0.5f});
    mPrefilter.SetTexture("InputTexture", original_tex);
    mPrefilter.SetRWTexture("OutputTexture", mip_chain, 1);
    cmd->Dispatch(&mPrefilter, CalculateDispatchSize(temp_tex->Width(), 16), CalculateDispatchSize(temp_tex->Height(), 16), 1);   // full-res grid (Q9)
    }
    {
    PIXScope(cmd, "Bloom Downsample");
    for (uint32 i = 0; i < BloomStep; i++) {   // downsample
        const uint32 upper = i + 1;
        const uint32 lw = temp_tex->Width() >> (upper + 1), lh = temp_tex->Height() >> (upper + 1);
        mDownsampleH[i].SetConstantBuffer(BlurConstant{texel(lw, lh)});
        mDownsampleH[i].SetTexture("InputTexture", mip_chain, upper);
        mDownsampleH[i].SetRWTexture("OutputTexture", temp_tex, upper + 1);
        { PIXScope(cmd, "Blur Horizontal"); cmd->Dispatch(&mDownsampleH[i], CalculateDispatchSize(lw, 256), CalculateDispatchSize(lh, 1), 1); }
        mDownsampleV[i].SetConstantBuffer(BlurConstant{texel(lw, lh)});
        mDownsampleV[i].SetTexture("InputTexture", temp_tex, i + 2);
        mDownsampleV[i].SetRWTexture("OutputTexture", mip_chain, i + 2);
        { PIXScope(cmd, "Blur Vertical"); cmd->Dispatch(&mDownsampleV[i], CalculateDispatchSize(lw, 1), CalculateDispatchSize(lh, 256), 1); }
    }
    }
    {
    PIXScope(cmd, "Bloom Upsample");
    for (int i = (int)BloomStep - 1; i >= 0; i--) {   // upsample: V(H(t1) + H(t2))
        const uint32 upper = (uint32)i + 1;
        const uint32 uw = temp_tex->Width() >> upper, uh = temp_tex->Height() >> upper;
        mUpsampleH[i].SetConstantBuffer(BlurConstant{texel(uw, uh)});
        mUpsampleH[i].SetTexture("UpperLevel", mip_chain, upper);
        mUpsampleH[i].SetTexture("LowerLevel", mip_chain, upper + 1);
        mUpsampleH[i].SetRWTexture("OutputTexture", temp_tex, i + 1);
        { PIXScope(cmd, "Upsample Horizontal Add"); cmd->Dispatch(&mUpsampleH[i], CalculateDispatchSize(uw, 256), CalculateDispatchSize(uh, 1), 1); }
        mUpsampleV[i].SetConstantBuffer(BlurConstant{texel(uw, uh)});
        mUpsampleV[i].SetTexture("InputTexture", temp_tex, upper);
        mUpsampleV[i].SetRWTexture("OutputTexture", mip_chain, upper);
        { PIXScope(cmd, "Blur Vertical"); cmd->Dispatch(&mUpsampleV[i], CalculateDispatchSize(uw, 1), CalculateDispatchSize(uh, 256), 1); }
    }
    }
    PIXScope(cmd, "Upsample Merge");
    const uint32 w = temp_tex->Width(), h = temp_tex->Height();   // merge
    mUpsampleBlurH.SetConstantBuffer(BlurConstant{texel(w, h)});
    mUpsampleBlurH.SetTexture("InputTexture", mip_chain, 1);
    mUpsampleBlurH.SetRWTexture("OutputTexture", temp_tex, 0);
    { PIXScope(cmd, "Blur Horizontal"); cmd->Dispatch(&mUpsampleBlurH, CalculateDispatchSize(w, 256), CalculateDispatchSize(h, 1), 1); }
    mUpsampleBlurV.SetConstantBuffer(BlurConstant{texel(w, h)});
    mUpsampleBlurV.SetTexture("InputTexture", temp_tex, 0);
    mUpsampleBlurV.SetRWTexture("OutputTexture", mip_chain, 0);
    { PIXScope(cmd, "Blur Vertical"); cmd->Dispatch(&mUpsampleBlurV, CalculateDispatchSize(w, 1), CalculateDispatchSize(h, 256), 1); }
    mUpsampleMerge.SetTexture("InputTexture", mip_chain, 0);
    mUpsampleMerge.SetRWTexture("OutputTexture", original_tex, 0);
    { PIXScope(cmd, "Merge"); cmd->Dispatch(&mUpsampleMerge, CalculateDispatchSize(original_tex->Width(), 16), CalculateDispatchSize(original_tex->Height(), 16), 1); }
}

}  // namespace MRendererHip
