// FrameGraph.h — pass graph: dependency sort by resource id, transient allocation, execution.
// API of Engine/Include/Renderer/FrameGraph.h:8-76 (FGExecutionParser::Parse / IsDependsOn /
// GetExecutionOrder / GetResourceLifecycle; FrameGraph::Setup / Compile / Execute / GetFGResource).
#pragma once
#include "IPipeline.h"

namespace MRendererHip {

class FGExecutionParser {
public:
    struct FGResourceLifecycle {
        FGResourceId ResourceId;
        uint32 StartPass, EndPass;
        bool Valid;   // is this resource ever touched by a pass
    };
    const std::vector<IRenderPass*>& GetExecutionOrder() const { return mExecutionOrder; }
    const std::vector<FGResourceLifecycle>& GetResourceLifecycle() const { return mResourceLifecycle; }
    void Parse(const std::vector<IRenderPass*>& passes, IRenderPass* present_pass);
    static bool IsDependsOn(const IRenderPass* lhs, const IRenderPass* rhs);
private:
    std::vector<IRenderPass*> mExecutionOrder;
    std::vector<FGResourceLifecycle> mResourceLifecycle;
};

class FrameGraph {
public:
    explicit FrameGraph(IRenderPipeline* pipeline) : mRenderPipeline(pipeline), mExecutionPass(0) {}
    FrameGraph(const FrameGraph&) = delete;
    void Setup();     // construct the passes (IRenderPipeline::Setup)
    void Compile();   // execution order + transient resource allocation
    void Execute(HipCommandList* cmd, Scene* scene, Camera* camera);
    // Frames alternate between two instances of these transient resources (the overlapped frame tail: the HDR target and the
    // luminance histogram of frame i are still read on the side stream while frame i + 1 is shaded).  Call after Compile.
    void DoubleBufferResources(const std::vector<FGResourceId>& ids);
    IRenderPipeline* GetPipeline() const { return mRenderPipeline; }
    IDeviceResource* GetFGResource(IRenderPass* pass, FGResourceId id);
    // by id, outside Execute (read-back by the host program / tests)
    IDeviceResource* FindResource(FGResourceId id);
    const std::vector<IRenderPass*>& ExecutionOrder() const { return mParser.GetExecutionOrder(); }
private:
    void PreparePass(HipCommandList* cmd, uint32 pass_index);
    const FGResourceDescriptionTable::Description& Describe(FGResourceId id) const { return mDescriptions.at(id); }
    FGExecutionParser mParser;
    FGResourceAllocator mFGResourceAllocator;
    std::vector<FGResourceDescriptionTable::Description> mDescriptions;   // this pipeline's declarations (copied at Compile)
    std::vector<IRenderPass*> mPipelinePasses;
    IRenderPipeline* mRenderPipeline;
    uint32 mExecutionPass;
    bool mDoubleBuffered = false;
    uint64_t mFrameCount = 0;
};

}  // namespace MRendererHip
