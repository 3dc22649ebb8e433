// FrameGraphResource.h — resource side of the host pass graph (HIP build).
//
// Mirrors the API surface of Engine/Include/Renderer/FrameGraphResource.h (FGResourceId,
// FGResourceIDs::NameToID :69-102, FGResourceDescriptionTable :130-213, FGResourceAllocator
// :215-270, FGContext :272-278) and the few device-resource classes the deferred pipeline binds
// (Engine/Include/Renderer/Device/Direct12/DeviceResource.h), with the D3D12 heaps replaced by
// linear HBM allocations: a "texture" is a row-major plane (mips concatenated), a structured
// buffer is a flat array.  Ownership follows the reference: transient resources belong to the
// FGResourceAllocator and are handed out as raw IDeviceResource* valid for the frame;
// persistent resources are owned (shared_ptr) by the pass that declares them.
#pragma once
#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <string_view>
#include <unordered_map>
#include <variant>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "../../include/pbr_hip.h"

namespace MRendererHip {

using uint32 = uint32_t;
using uint16 = uint16_t;
using int32 = int32_t;

// throw-on-failure like ThrowIfFailed/DxException (Engine/Include/Renderer/Device/Direct12/D3DUtils.h:12-41)
struct HipException : std::runtime_error {
    using std::runtime_error::runtime_error;
};
inline void ThrowIfFailed(hipError_t e, const char* what) {
    if (e != hipSuccess) throw HipException(std::string(what) + ": " + hipGetErrorString(e));
}

enum ETextureFormat : uint16 {
    ETextureFormat_None = 0,
    ETextureFormat_R8G8B8A8_UNORM,
    ETextureFormat_R16G16_FLOAT,
    ETextureFormat_R16G16B16A16_FLOAT,
    ETextureFormat_R32G32B32A32_FLOAT,
    ETextureFormat_DepthStencil,   // D32_FLOAT plane + S8 plane
};
enum ETexture2DFlag : uint16 {
    ETexture2DFlag_None = 0,
    ETexture2DFlag_AllowRenderTarget = 1,
    ETexture2DFlag_AllowDepthStencil = 2,
    ETexture2DFlag_AllowUnorderedAccess = 4,
};
inline uint32 BytesPerTexel(ETextureFormat f) {
    switch (f) {
        case ETextureFormat_R8G8B8A8_UNORM: return 4;
        case ETextureFormat_R16G16_FLOAT: return 4;
        case ETextureFormat_R16G16B16A16_FLOAT: return 8;
        case ETextureFormat_R32G32B32A32_FLOAT: return 16;
        case ETextureFormat_DepthStencil: return 5;   // 4 (depth plane) + 1 (stencil plane)
        default: return 0;
    }
}

class IDeviceResource {
public:
    virtual ~IDeviceResource() = default;
    virtual void* DevicePtr() const = 0;
    virtual size_t Bytes() const = 0;
};

class DeviceMemory {
public:
    // Dry run: build and sort a pass graph on a machine without a GPU (no allocation, null pointers);
    // nothing can be dispatched in this mode.  Used by the CPU tests of the graph logic only.
    static bool& DryRun() {
        static bool dry = false;
        return dry;
    }
    explicit DeviceMemory(size_t bytes) : mBytes(bytes) {
        if (DryRun()) return;
        ThrowIfFailed(hipMalloc(&mPtr, bytes ? bytes : 1), "hipMalloc");
        ThrowIfFailed(hipMemset(mPtr, 0, bytes ? bytes : 1), "hipMemset");
        // hipMemset of device memory is ASYNCHRONOUS to the host and runs on the null stream, which the contexts' non-blocking
        // streams do not wait for: without this wait a kernel enqueued next on a context's stream can write the buffer BEFORE the
        // zero-fill lands and lose part of what it wrote (round 4: the padded env chain built right after its allocation came out
        // with zeroed texels once in ~10 renderer creations — a few hundred wrong pixels per frame; found by
        // test_host_graph_throughput_mode_renders_the_same_frames).  Allocation is rare: a host wait here costs nothing per frame.
        ThrowIfFailed(hipStreamSynchronize(nullptr), "hipStreamSynchronize(null stream) after zero-fill");
    }
    ~DeviceMemory() { if (mPtr) (void)hipFree(mPtr); }
    DeviceMemory(const DeviceMemory&) = delete;
    DeviceMemory& operator=(const DeviceMemory&) = delete;
    void* Ptr() const { return mPtr; }
    size_t Bytes() const { return mBytes; }
private:
    void* mPtr = nullptr;
    size_t mBytes;
};

class DeviceTexture : public IDeviceResource {};

// Row-major 2D texture with a mip chain (level l is (w>>l) x (h>>l), levels concatenated).
class DeviceTexture2D : public DeviceTexture {
public:
    DeviceTexture2D(uint32 w, uint32 h, uint32 mips, ETextureFormat fmt)
        : mWidth(w), mHeight(h), mMips(mips), mFormat(fmt), mMem(TexelCount(w, h, mips) * BytesPerTexel(fmt)) {}
    static size_t TexelCount(uint32 w, uint32 h, uint32 mips) {
        size_t n = 0;
        for (uint32 l = 0; l < mips; l++) n += (size_t)(w >> l) * (h >> l);
        return n;
    }
    uint32 Width() const { return mWidth; }
    uint32 Height() const { return mHeight; }
    uint32 MipLevels() const { return mMips; }
    ETextureFormat Format() const { return mFormat; }
    void* DevicePtr() const override { return mMem.Ptr(); }
    size_t Bytes() const override { return mMem.Bytes(); }
    void* MipPtr(uint32 mip) const {
        return (char*)mMem.Ptr() + TexelCount(mWidth, mHeight, mip) * BytesPerTexel(mFormat);
    }
    // depth-stencil: depth plane first, stencil plane after it
    float* DepthPlane() const { return (float*)mMem.Ptr(); }
    uint8_t* StencilPlane() const { return (uint8_t*)mMem.Ptr() + (size_t)mWidth * mHeight * 4; }
private:
    uint32 mWidth, mHeight, mMips;
    ETextureFormat mFormat;
    DeviceMemory mMem;
};

// Cube / 2D array: mips concatenated, 6 faces per mip (the pbr_cube_f32 layout of include/pbr_hip.h)
class DeviceTexture2DArray : public DeviceTexture {
public:
    DeviceTexture2DArray(uint32 size, uint32 mips, ETextureFormat fmt)
        : mSize(size), mMips(mips), mFormat(fmt), mMem(pbr_cube_texels(size, mips) * BytesPerTexel(fmt)) {}
    uint32 Size() const { return mSize; }
    uint32 MipLevels() const { return mMips; }
    ETextureFormat Format() const { return mFormat; }
    void* DevicePtr() const override { return mMem.Ptr(); }
    size_t Bytes() const override { return mMem.Bytes(); }
private:
    uint32 mSize, mMips;
    ETextureFormat mFormat;
    DeviceMemory mMem;
};

class DeviceStructuredBuffer : public IDeviceResource {
public:
    DeviceStructuredBuffer(uint32 size, uint32 stride) : mSize(size), mStride(stride), mMem(size) {}
    uint32 Size() const { return mSize; }
    uint32 Stride() const { return mStride; }
    void* DevicePtr() const override { return mMem.Ptr(); }
    size_t Bytes() const override { return mMem.Bytes(); }
    // DeviceStructuredBuffer::Commit: host -> device upload (synchronous like the reference's upload ring + fence)
    void Commit(const void* data, size_t bytes) {
        if (bytes > mSize) throw HipException("DeviceStructuredBuffer::Commit: size overflow");
        ThrowIfFailed(hipMemcpy(mMem.Ptr(), data, bytes, hipMemcpyHostToDevice), "hipMemcpy");
    }
private:
    uint32 mSize, mStride;
    DeviceMemory mMem;
};

// ------------------------------------------------------------------------------------ resource ids
using FGResourceId = int32;
constexpr FGResourceId InvalidFGResourceId = -1;

class FGResourceIDs {
public:
    static FGResourceIDs* Instance() {
        static FGResourceIDs instance;
        return &instance;
    }
    FGResourceId NameToID(const std::string& name) {
        auto [it, inserted] = mTable.try_emplace(name, (FGResourceId)mNames.size());
        if (inserted) mNames.push_back(name);
        return it->second;
    }
    std::string_view IdToName(FGResourceId id) const { return mNames[id]; }
    uint32 NumResources() const { return (uint32)mNames.size(); }
private:
    std::vector<std::string> mNames;
    std::unordered_map<std::string, FGResourceId> mTable;
};

struct FGTransientTextureDescription {
    uint16 Width = 0, Height = 0, MipLevels = 0;
    ETextureFormat Format = ETextureFormat_None;
    ETexture2DFlag Flag = ETexture2DFlag_None;
    bool operator==(const FGTransientTextureDescription& o) const {
        return Width == o.Width && Height == o.Height && MipLevels == o.MipLevels && Format == o.Format && Flag == o.Flag;
    }
    bool Empty() const { return Width == 0 && Height == 0; }
};
struct FGTransientBufferDescription {
    uint32 Size = 0, Stride = 0;
    bool operator==(const FGTransientBufferDescription& o) const { return Size == o.Size && Stride == o.Stride; }
    bool Empty() const { return Size == 0 && Stride == 0; }
};
struct FGPersistentResourceDescription {
    IDeviceResource* Resource = nullptr;
    bool operator==(const FGPersistentResourceDescription& o) const { return Resource == o.Resource; }
    bool Empty() const { return Resource == nullptr; }
};

class FGResourceDescriptionTable {
public:
    using Description = std::variant<FGTransientTextureDescription, FGTransientBufferDescription, FGPersistentResourceDescription>;
    static FGResourceDescriptionTable* Instance() {
        static FGResourceDescriptionTable instance;
        return &instance;
    }
    void DeclareTransientTexture(FGResourceId id, uint32 w, uint32 h, uint32 mips, ETextureFormat fmt, ETexture2DFlag flag) {
        // TextureFormatKey stores 16-bit extents (FrameGraphResource.h:8-18, quirk Q24)
        if (w > 65535 || h > 65535) throw HipException("DeclareTransientTexture: extent exceeds 65535");
        Declare(id, FGTransientTextureDescription{(uint16)w, (uint16)h, (uint16)mips, fmt, flag});
    }
    void DeclareTransientBuffer(FGResourceId id, uint32 size, uint32 stride) { Declare(id, FGTransientBufferDescription{size, stride}); }
    void DeclarePersistentResource(FGResourceId id, IDeviceResource* res) { Declare(id, FGPersistentResourceDescription{res}); }
    const Description& Get(FGResourceId id) const { return mDescriptions.at(id); }
    const FGTransientTextureDescription& GetTransientTexture(FGResourceId id) const { return std::get<FGTransientTextureDescription>(mDescriptions.at(id)); }
    const FGTransientBufferDescription& GetTransientBuffer(FGResourceId id) const { return std::get<FGTransientBufferDescription>(mDescriptions.at(id)); }
    const FGPersistentResourceDescription& GetPersistentResource(FGResourceId id) const { return std::get<FGPersistentResourceDescription>(mDescriptions.at(id)); }
    // a new renderer instance re-declares everything (the reference has one pipeline per process)
    void Reset() { mDescriptions.clear(); }
    // the declarations of the pipeline that was just set up: a FrameGraph keeps its own copy (FrameGraph::Compile), so that
    // several renderers can live in one process although this table — like the reference's — is a process-wide singleton
    const std::vector<Description>& All() const { return mDescriptions; }
private:
    template <class T>
    void Declare(FGResourceId id, const T& desc) {
        if ((size_t)id >= mDescriptions.size()) mDescriptions.resize(FGResourceIDs::Instance()->NumResources(), Description{});
        // re-declaring with a different description is a contract violation (CheckDescription :189-192)
        const Description& cur = mDescriptions[id];
        bool empty = std::holds_alternative<FGTransientTextureDescription>(cur) && std::get<FGTransientTextureDescription>(cur).Empty();
        if (!empty && !(std::holds_alternative<T>(cur) && std::get<T>(cur) == desc))
            throw HipException("FGResourceDescriptionTable: conflicting declaration of " + std::string(FGResourceIDs::Instance()->IdToName(id)));
        mDescriptions[id] = desc;
    }
    std::vector<Description> mDescriptions;
};

class FGResourceAllocator {
public:
    void Reset() {
        mTransient.assign(FGResourceIDs::Instance()->NumResources(), nullptr);
        mSecond.assign(FGResourceIDs::Instance()->NumResources(), nullptr);
        mParity = 0;
    }
    void AllocateTransientResource(FGResourceId id, const FGResourceDescriptionTable::Description& d) { mTransient[id] = Make(d); }
    // A second instance of a transient resource: frames alternate between the two (SetParity), so that work of frame i that is
    // still in flight on another stream — the overlapped frame tail — does not see frame i + 1 write the same memory.  Not in
    // the reference (its frames are strictly serial).
    void DoubleBuffer(FGResourceId id, const FGResourceDescriptionTable::Description& d) {
        if (!mSecond.at(id)) mSecond[id] = Make(d);
    }
    void SetParity(uint32 p) { mParity = p & 1u; }
    uint32 Parity() const { return mParity; }
    IDeviceResource* GetResource(FGResourceId id) const {
        IDeviceResource* r = (mParity && mSecond.at(id)) ? mSecond[id].get() : mTransient.at(id).get();
        if (!r) throw HipException("FGResourceAllocator: resource not allocated: " + std::string(FGResourceIDs::Instance()->IdToName(id)));
        return r;
    }
private:
    static std::shared_ptr<IDeviceResource> Make(const FGResourceDescriptionTable::Description& d) {
        if (auto* t = std::get_if<FGTransientTextureDescription>(&d)) return std::make_shared<DeviceTexture2D>(t->Width, t->Height, t->MipLevels, t->Format);
        if (auto* b = std::get_if<FGTransientBufferDescription>(&d)) return std::make_shared<DeviceStructuredBuffer>(b->Size, b->Stride);
        return nullptr;
    }
    std::vector<std::shared_ptr<IDeviceResource>> mTransient, mSecond;
    uint32 mParity = 0;
};

class Scene;
class Camera;
class HipCommandList;
class FrameGraph;

// member names as in the reference (FrameGraphResource.h:272-278); qualified types keep GCC's
// "changes meaning" rule quiet where MSVC is permissive
struct FGContext {
    MRendererHip::HipCommandList* CommandList;
    MRendererHip::Scene* Scene;
    MRendererHip::Camera* Camera;
    MRendererHip::FrameGraph* FrameGraph;
};

}  // namespace MRendererHip
