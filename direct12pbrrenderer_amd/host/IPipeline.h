// IPipeline.h — pass API of the host pass graph (HIP build).
//
// Same surface as Engine/Include/Renderer/Pipeline/IPipeline.h: ConstantBufferGlobal (:38-62),
// ShadingState (:123-168: SetShader / SetTexture / SetRWTexture / SetRWTextureArray /
// Set[RW]StructuredBuffer / SetConstantBuffer<T>, bool result = "name known to the shader"),
// IRenderPass (:170-230: ReadResource / WriteResource / WriteTransientTexture /
// WriteTransientBuffer / WritePersistentResource / GetTransientResource / Execute(FGContext*)),
// PresentPass, GraphicsPass, ComputePass, IRenderPipeline (:232-286).  What changes is below
// the seam: a "shader" is one of the HIP kernels behind include/pbr_hip.h and
// HipCommandList::Dispatch / DrawScreen (HipCommandList.h) call it.
#pragma once
#include <algorithm>
#include <array>
#include <cassert>
#include <cstdio>
#include <map>

#include "FrameGraphResource.h"

namespace MRendererHip {

using ConstantBufferGlobal = pbr_global;   // IPipeline.h:38-62 field for field (412 bytes)
static_assert(sizeof(ConstantBufferGlobal) == 412);

// limits of the reference's binding model (Engine/Include/Fundation.h:28-30)
constexpr uint32 MaxShaderResourceViews = 8;
constexpr uint32 MaxUnorderedAccessViews = 8;

// What DXC reflection gives the reference (Shader.cpp): the resource names a shader file declares.
struct ShaderReflection {
    std::string_view File;
    bool IsCompute;
    std::vector<std::string_view> Textures;            // SRV textures
    std::vector<std::string_view> RWTextures;          // UAV textures
    std::vector<std::string_view> StructuredBuffers;   // SRV buffers
    std::vector<std::string_view> RWStructuredBuffers; // UAV buffers
    uint32 ConstantBufferSize;                          // bytes of CONSTANT_BUFFER_SHADER (0 = none)
};
const ShaderReflection* FindShader(std::string_view file);   // nullptr if the file is not a kernel of this build

struct TextureBinding {
    DeviceTexture* Texture = nullptr;
    int32 MipSlice = -1;   // -1 = whole resource
};

class ShadingState {
public:
    ShadingState() = default;
    ShadingState(const ShadingState&) = delete;
    ShadingState(ShadingState&&) = default;
    ShadingState& operator=(ShadingState&&) = default;

    void SetShader(std::string_view shader_file_path, bool is_compute);
    bool SetTexture(std::string_view semantic_name, DeviceTexture* texture);
    bool SetTexture(std::string_view semantic_name, DeviceTexture2D* texture, uint32 mip_slice);
    bool SetRWTexture(std::string_view semantic_name, DeviceTexture2D* texture);
    bool SetRWTexture(std::string_view semantic_name, DeviceTexture2D* texture, uint32 mip_slice);
    bool SetRWTextureArray(std::string_view semantic_name, DeviceTexture2DArray* texture);
    bool SetStructuredBuffer(std::string_view semantic_name, DeviceStructuredBuffer* buffer);
    bool SetRWStructuredBuffer(std::string_view semantic_name, DeviceStructuredBuffer* buffer);
    void ClearResourceBinding();

    template <typename T>
    void SetConstantBuffer(const T& t) {
        static_assert(std::is_trivially_copyable_v<T>);
        mConstants.resize(sizeof(T));
        std::memcpy(mConstants.data(), &t, sizeof(T));   // POD memcpy, HLSL packing is the struct's
    }
    template <typename T>
    const T& Constants() const {
        if (mConstants.size() != sizeof(T)) throw HipException("ShadingState: constant buffer not set / wrong size for " + std::string(File()));
        return *reinterpret_cast<const T*>(mConstants.data());
    }

    const ShaderReflection* GetShader() const { return mShader; }
    std::string_view File() const { return mShader ? mShader->File : std::string_view("<none>"); }
    bool IsCompute() const { return mIsCompute; }

    // lookups used by HipCommandList (throw when an expected binding is missing)
    const TextureBinding& Texture(std::string_view name) const;
    const TextureBinding& RWTexture(std::string_view name) const;
    DeviceStructuredBuffer* Buffer(std::string_view name) const;
    bool HasTexture(std::string_view name) const { return mTextures.count(std::string(name)) != 0; }

private:
    bool Known(const std::vector<std::string_view>& names, std::string_view semantic_name, const char* kind) const;

    const ShaderReflection* mShader = nullptr;
    bool mIsCompute = false;
    std::map<std::string, TextureBinding> mTextures, mRWTextures;
    std::map<std::string, DeviceStructuredBuffer*> mBuffers;
    std::vector<uint8_t> mConstants;
};

class IRenderPass {
    friend class FrameGraph;
    friend class RenderScheduler;

public:
    IRenderPass() = default;
    virtual ~IRenderPass() {}
    IRenderPass(const IRenderPass&) = delete;
    IRenderPass& operator=(const IRenderPass&) = delete;

    const std::vector<FGResourceId>& GetInputResources() const { return mInputResources; }
    const std::vector<FGResourceId>& GetOutputResources() const { return mOutputResources; }
    virtual const char* Name() const = 0;

protected:
    void ReadResource(FGResourceId id) {
        assert(std::find(mInputResources.begin(), mInputResources.end(), id) == mInputResources.end());
        mInputResources.push_back(id);
    }
    void WriteResource(FGResourceId id) {
        assert(std::find(mOutputResources.begin(), mOutputResources.end(), id) == mOutputResources.end());
        mOutputResources.push_back(id);
    }
    void WriteTransientTexture(FGResourceId id, uint32 width, uint32 height, uint32 mip_levels, ETextureFormat format,
                               ETexture2DFlag flag = ETexture2DFlag_AllowRenderTarget) {
        FGResourceDescriptionTable::Instance()->DeclareTransientTexture(id, width, height, mip_levels, format, flag);
        WriteResource(id);
    }
    void WriteTransientBuffer(FGResourceId id, uint32 size, uint32 stride) {
        FGResourceDescriptionTable::Instance()->DeclareTransientBuffer(id, size, stride);
        WriteResource(id);
    }
    void WritePersistentResource(FGResourceId id, IDeviceResource* res) {
        FGResourceDescriptionTable::Instance()->DeclarePersistentResource(id, res);
        WriteResource(id);
    }
    IDeviceResource* GetTransientResource(FGContext* context, FGResourceId id);

    virtual void Execute(FGContext* context) = 0;

    std::vector<FGResourceId> mInputResources;
    std::vector<FGResourceId> mOutputResources;
};

class PresentPass : public IRenderPass {
public:
    PresentPass() : mFinalTexture(InvalidFGResourceId) {}
    const char* Name() const override { return "Present"; }
    void Execute(FGContext* context) override;
    void SetFinalTexture(FGResourceId id) {
        mFinalTexture = id;
        ReadResource(id);
    }
    FGResourceId FinalTexture() const { return mFinalTexture; }
protected:
    FGResourceId mFinalTexture;
};

class GraphicsPass : public IRenderPass {};   // full-screen passes of the reference; compute kernels here
class ComputePass : public IRenderPass {};

class IRenderPipeline {
    friend class FrameGraph;

public:
    IRenderPipeline() { mPresentPass = std::make_unique<PresentPass>(); }
    virtual ~IRenderPipeline() {}
    IRenderPipeline(const IRenderPipeline&) = delete;
    IRenderPipeline& operator=(const IRenderPipeline&) = delete;
    virtual std::vector<IRenderPass*> Setup() = 0;
protected:
    std::unique_ptr<PresentPass> mPresentPass;
};

}  // namespace MRendererHip
