// DeferredPipeline.h — the deferred pipeline's pass classes over the HIP kernels.
//
// Same classes, resource names, constants and declared reads/writes as
// Engine/Include/Renderer/Pipeline/DeferredPipeline.h (line references per class); Execute bodies
// in DeferredPipeline.cpp bind by the reference's shader resource names and dispatch with the
// reference's thread-group counts.  GBufferPass / SkyboxPass are raster passes; rasterization is out
// of scope, their per-pixel work is not: GBufferPass uploads either the encoded G-buffer
// planes or the rasterizer's per-pixel material attributes and encodes them with gbuffer.hlsl's
// pixel-shader math; SkyboxPass resolves the sky on the pixels geometry left uncovered.
#pragma once
#include "FrameGraph.h"
#include "Scene.h"
#include "ShaderConstants.h"

namespace MRendererHip {

struct DeferredPipelineResource {   // DeferredPipeline.h:9-32
    inline static FGResourceId PrefilterEnvMap = FGResourceIDs::Instance()->NameToID("PrefilterEnvMap");
    inline static FGResourceId PrecomputeBRDF = FGResourceIDs::Instance()->NameToID("PrecomputeBRDF");
    inline static FGResourceId GBufferA = FGResourceIDs::Instance()->NameToID("GBufferA");
    inline static FGResourceId GBufferB = FGResourceIDs::Instance()->NameToID("GBufferB");
    inline static FGResourceId GBufferC = FGResourceIDs::Instance()->NameToID("GBufferC");
    inline static FGResourceId DepthStencil = FGResourceIDs::Instance()->NameToID("GBufferDepthStencil");
    inline static FGResourceId DeferredShadingRT = FGResourceIDs::Instance()->NameToID("DeferredShadingRT");
    inline static FGResourceId BloomMipchain = FGResourceIDs::Instance()->NameToID("BloomMipchain");
    inline static FGResourceId BloomTempTexture = FGResourceIDs::Instance()->NameToID("BloomTempTexture");
    inline static FGResourceId ToneMappedTexture = FGResourceIDs::Instance()->NameToID("ToneMappedTexture");
    inline static FGResourceId FrustumCluster = FGResourceIDs::Instance()->NameToID("FrustumCluster");
    inline static FGResourceId PointLights = FGResourceIDs::Instance()->NameToID("ClusteredLights");
    inline static FGResourceId LuminanceHistogram = FGResourceIDs::Instance()->NameToID("LuminanceHistogram");
    inline static FGResourceId AverageLuminance = FGResourceIDs::Instance()->NameToID("AverageLuminance");
};

// render size (GD3D12Device->Width()/Height() in the reference; App.h:77-78 defaults 1440x960)
struct RenderSize { uint32 Width, Height; };

class PreFilterEnvMapPass : public ComputePass {   // DeferredPipeline.h:35-70
public:
    static constexpr uint32 PreFilterEnvMapSize = 512;
    static constexpr uint32 PreFilterEnvMapMipsLevel = 5;
    static constexpr uint32 DispatchGroupSize = 8;
    explicit PreFilterEnvMapPass(uint32 size = PreFilterEnvMapSize);
    const char* Name() const override { return "PreFilterEnvMap"; }
    void Execute(FGContext* context) override;
    void Invalidate() { mReady = false; }
protected:
    uint32 mSize;
    std::array<ShadingState, PreFilterEnvMapMipsLevel> mShadingState;
    std::shared_ptr<DeviceTexture2DArray> mPrefilterEnvMap;
    bool mReady;
};

class PrecomputeBRDFPass : public ComputePass {   // :72-99
public:
    static constexpr uint32 TextureResolution = 512;
    explicit PrecomputeBRDFPass(uint32 res = TextureResolution);
    const char* Name() const override { return "PrecomputeBRDF"; }
    void Execute(FGContext* context) override;
protected:
    uint32 mRes;
    ShadingState mShadingState;
    std::shared_ptr<DeviceTexture2D> mPrecomputeBRDF;
    bool mReady;
};

class GBufferPass : public GraphicsPass {   // :101-137 (rasterization out of scope; ps_main's encode is pbr_gbuffer_encode)
public:
    explicit GBufferPass(RenderSize s);
    const char* Name() const override { return "GBuffer"; }
    void Execute(FGContext* context) override;
protected:
    ShadingState mShadingState;
    std::unique_ptr<DeviceStructuredBuffer> mMaterialPlanes;   // device copy of GBufferSource::M0..M2
};

class DeferredShadingPass : public GraphicsPass {   // :139-190
public:
    explicit DeferredShadingPass(RenderSize s);
    const char* Name() const override { return "DeferredShading"; }
protected:
    void Execute(FGContext* context) override;
    ShadingState mShadingState;
};

class SkyboxPass : public GraphicsPass {   // :192-206, DeferredPipeline.cpp:46-75 (sky resolve on stencil == 0)
public:
    SkyboxPass();
    const char* Name() const override { return "Skybox"; }
    void Execute(FGContext* context) override;
protected:
    ShadingState mShadingState;
};

class BloomPass : public ComputePass {   // :208-298
public:
    static constexpr uint32 BloomStep = 3;
    static constexpr uint32 MipmapLevel = BloomStep + 2;
    // halo_chain: {0, 0} = the mip chains have DeferredShadingRT's size (the reference); multi-GPU halo mode:
    // the size of the tile's extended rectangle E, on which the pyramid runs (the HDR target covers S = interior + 4 px)
    explicit BloomPass(RenderSize halo_chain = RenderSize{0, 0});
    const char* Name() const override { return "Bloom"; }
    void Execute(FGContext* context) override;
protected:
    ShadingState mDownsampleH[BloomStep], mDownsampleV[BloomStep], mUpsampleH[BloomStep], mUpsampleV[BloomStep];
    ShadingState mUpsampleBlurH, mUpsampleBlurV, mUpsampleMerge, mPrefilter;
    bool mHalo = false;
};

class ClusteredPass : public ComputePass {   // :300-369
public:
    static constexpr int32 ClusterSizeX = 24, ClusterSizeY = 16, ClusterSizeZ = 8;
    static constexpr int32 MaxSceneLights = 1024, MaxClusterLights = 32;
    ClusteredPass();
    const char* Name() const override { return "Clustered"; }
    void Execute(FGContext* context) override;
protected:
    ShadingState mClusteredCompute, mClusteredCulling;
    std::vector<pbr_light> mLights, mCommitted;   // this frame's light buffer / what the device buffer holds
    int mCommittedCount = -1;
};

class AutoExposurePass : public ComputePass {   // :371-429
public:
    static constexpr float MinLogLuminance = -10.0f, MaxLogLuminance = 2.0f;
    static constexpr float LogLuminanceRange = MaxLogLuminance - MinLogLuminance;
    static constexpr float InvLogLuminanceRange = 1.0f / (MaxLogLuminance - MinLogLuminance);
    static constexpr uint32 HistogramComputeThreadGroupSize = 16, HistogramBinSize = 256;
    AutoExposurePass();
    const char* Name() const override { return "AutoExposure"; }
    void Execute(FGContext* context) override;
    void SetInitialLuminance(float v) { mInitialLuminance = v; mAvarageLuminanceInitialized = false; }
    // multi-GPU: PixelCount of the WHOLE frame; 0 = this device's texture
    void SetFullFramePixelCount(uint32 n) { mFullFramePixels = n; }
protected:
    ShadingState mLuminanceHistogramCompute, mAvarageLuminanceCompute;
    bool mAvarageLuminanceInitialized;
    float mInitialLuminance = 0.0f;   // DeferredPipeline.cpp:268-274 initialises to 0 (quirk Q20)
    uint32 mFullFramePixels = 0;
};

class ToneMappingPass : public GraphicsPass {   // :431-461
public:
    explicit ToneMappingPass(RenderSize s);
    const char* Name() const override { return "ToneMapping"; }
    void Execute(FGContext* context) override;
protected:
    ShadingState mToneMappingRender;
};

class DeferredRenderPipeline : public IRenderPipeline {   // :463-481, DeferredPipeline.cpp:17-44
public:
    DeferredRenderPipeline(RenderSize size, uint32 env_size = PreFilterEnvMapPass::PreFilterEnvMapSize, uint32 lut_res = PrecomputeBRDFPass::TextureResolution)
        : mSize(size), mEnvSize(env_size), mLutRes(lut_res) {}
    // one device's tile of a larger frame: every target covers the layout's shaded rectangle S; in halo mode the
    // bloom chains cover the extended rectangle E
    DeferredRenderPipeline(const TileLayout& layout, uint32 env_size = PreFilterEnvMapPass::PreFilterEnvMapSize, uint32 lut_res = PrecomputeBRDFPass::TextureResolution)
        : mSize(RenderSize{layout.Shaded.w, layout.Shaded.h}), mEnvSize(env_size), mLutRes(lut_res),
          mHaloChain(layout.Halo ? RenderSize{layout.Bloom.w, layout.Bloom.h} : RenderSize{0, 0}) {}
    std::vector<IRenderPass*> Setup() override;

    std::unique_ptr<GBufferPass> mGBufferPass;
    std::unique_ptr<DeferredShadingPass> mDeferredShadingPass;
    std::unique_ptr<SkyboxPass> mSkyboxPass;
    std::unique_ptr<BloomPass> mBloomPass;
    std::unique_ptr<PreFilterEnvMapPass> mPrefilterEnvMapPass;
    std::unique_ptr<PrecomputeBRDFPass> mPrecomputeBRDFPass;
    std::unique_ptr<AutoExposurePass> mAutoExposurePass;
    std::unique_ptr<ToneMappingPass> mToneMappingPass;
    std::unique_ptr<ClusteredPass> mClusteredPass;
private:
    RenderSize mSize;
    uint32 mEnvSize, mLutRes;
    RenderSize mHaloChain{0, 0};
};

}  // namespace MRendererHip
