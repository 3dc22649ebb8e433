// FrameGraph.cpp — see FrameGraph.h.  The ordering rule is the reference's
// (Engine/Source/Renderer/FrameGraph.cpp:191-250): pass L depends on pass R when an input of L
// is an output of R; starting from the present pass, a pass is emitted once every pass that
// depends on it has been emitted (stack order), and the emitted list is reversed.  Reproducing
// the rule — not just "a" topological order — matters: BloomPass declares no reads, so its slot
// between DeferredShading and AutoExposure is decided by this traversal (quirk Q21).
#include "FrameGraph.h"

#include <stack>

#include "HipCommandList.h"

namespace MRendererHip {

bool FGExecutionParser::IsDependsOn(const IRenderPass* lhs, const IRenderPass* rhs) {
    if (lhs == rhs) return false;
    for (FGResourceId in : lhs->GetInputResources())
        for (FGResourceId out : rhs->GetOutputResources())
            if (in == out) return true;
    return false;
}

void FGExecutionParser::Parse(const std::vector<IRenderPass*>& passes, IRenderPass* present_pass) {
    struct Node {
        IRenderPass* Pass;
        std::vector<size_t> Inputs;   // passes this one depends on
        uint32 RefCount = 0;          // passes depending on this one that are not emitted yet
        bool Visited = false;
    };
    mExecutionOrder.clear();
    std::vector<Node> nodes;
    for (IRenderPass* p : passes) nodes.push_back(Node{p, {}, 0, false});
    size_t final_pass = nodes.size();
    for (size_t l = 0; l < nodes.size(); l++) {
        for (size_t r = 0; r < nodes.size(); r++)
            if (IsDependsOn(nodes[l].Pass, nodes[r].Pass)) {
                nodes[r].RefCount++;
                nodes[l].Inputs.push_back(r);
            }
        if (nodes[l].Pass == present_pass) final_pass = l;
    }
    if (final_pass == nodes.size() || nodes[final_pass].RefCount != 0) throw HipException("FrameGraph: present pass missing or depended upon");
    std::stack<size_t> dfs;
    dfs.push(final_pass);
    while (!dfs.empty()) {
        size_t n = dfs.top();
        dfs.pop();
        mExecutionOrder.push_back(nodes[n].Pass);
        for (size_t d : nodes[n].Inputs) {
            nodes[d].RefCount -= 1;
            if (!nodes[d].Visited && nodes[d].RefCount == 0) {
                nodes[d].Visited = true;
                dfs.push(d);
            }
        }
    }
    if (passes.size() != mExecutionOrder.size()) throw HipException("FrameGraph: unused pass or circular reference in the frame graph");
    std::reverse(mExecutionOrder.begin(), mExecutionOrder.end());

    const uint32 n_res = FGResourceIDs::Instance()->NumResources();
    mResourceLifecycle.assign(n_res, FGResourceLifecycle{0, 0, 0, false});
    for (uint32 i = 0; i < n_res; i++) mResourceLifecycle[i].ResourceId = (FGResourceId)i;
    for (uint32 i = 0; i < mExecutionOrder.size(); i++) {
        auto extend = [&](FGResourceId res) {
            auto& lc = mResourceLifecycle[res];
            if (lc.Valid) {
                lc.StartPass = std::min(lc.StartPass, i);
                lc.EndPass = std::max(lc.EndPass, i);
            } else {
                lc.Valid = true;
                lc.StartPass = lc.EndPass = i;
            }
        };
        for (FGResourceId r : mExecutionOrder[i]->GetInputResources()) extend(r);
        for (FGResourceId r : mExecutionOrder[i]->GetOutputResources()) extend(r);
    }
}

void FrameGraph::Setup() { mPipelinePasses = mRenderPipeline->Setup(); }

void FrameGraph::Compile() {
    mParser.Parse(mPipelinePasses, mRenderPipeline->mPresentPass.get());
    mDescriptions = FGResourceDescriptionTable::Instance()->All();
    mDescriptions.resize(FGResourceIDs::Instance()->NumResources());
    mFGResourceAllocator.Reset();
    for (auto& lc : mParser.GetResourceLifecycle())
        if (lc.Valid) mFGResourceAllocator.AllocateTransientResource(lc.ResourceId, Describe(lc.ResourceId));
}

void FrameGraph::DoubleBufferResources(const std::vector<FGResourceId>& ids) {
    for (FGResourceId id : ids) mFGResourceAllocator.DoubleBuffer(id, Describe(id));
    mDoubleBuffered = !ids.empty();
}

void FrameGraph::Execute(HipCommandList* cmd, Scene* scene, Camera* camera) {
    if (mDoubleBuffered) mFGResourceAllocator.SetParity((uint32)(mFrameCount++ & 1u));
    FGContext context{cmd, scene, camera, this};
    const auto& order = mParser.GetExecutionOrder();
    for (mExecutionPass = 0; mExecutionPass < order.size(); mExecutionPass++) {
        PreparePass(cmd, mExecutionPass);
        order[mExecutionPass]->Execute(&context);
    }
}

IDeviceResource* FrameGraph::FindResource(FGResourceId id) {
    const auto& d = Describe(id);
    if (auto* p = std::get_if<FGPersistentResourceDescription>(&d)) return p->Resource;
    return mFGResourceAllocator.GetResource(id);
}

IDeviceResource* FrameGraph::GetFGResource(IRenderPass* pass, FGResourceId id) {
    assert(mParser.GetExecutionOrder()[mExecutionPass] == pass);
    const auto& in = pass->GetInputResources();
    const auto& out = pass->GetOutputResources();
    if (std::find(in.begin(), in.end(), id) == in.end() && std::find(out.begin(), out.end(), id) == out.end())
        throw HipException(std::string(pass->Name()) + " accesses undeclared resource " + std::string(FGResourceIDs::Instance()->IdToName(id)));
    return FindResource(id);
}

// Bind the render target a graphics pass writes (the reference binds + clears RTs here,
// FrameGraph.cpp:94-141; clearing is the raster passes' business and they are out of scope).
void FrameGraph::PreparePass(HipCommandList* cmd, uint32 pass_index) {
    auto* pass = dynamic_cast<GraphicsPass*>(mParser.GetExecutionOrder()[pass_index]);
    if (!pass) return;
    DeviceTexture2D *rt = nullptr, *ds = nullptr;
    for (FGResourceId id : pass->GetOutputResources()) {
        const auto& d = Describe(id);
        if (auto* t = std::get_if<FGTransientTextureDescription>(&d)) {
            if (t->Format != ETextureFormat_DepthStencil && !rt) rt = dynamic_cast<DeviceTexture2D*>(mFGResourceAllocator.GetResource(id));
            if (t->Format == ETextureFormat_DepthStencil && !ds) ds = dynamic_cast<DeviceTexture2D*>(mFGResourceAllocator.GetResource(id));
        }
    }
    cmd->SetRenderTarget(rt);
    cmd->SetDepthStencil(ds);
}

IDeviceResource* IRenderPass::GetTransientResource(FGContext* context, FGResourceId id) { return context->FrameGraph->GetFGResource(this, id); }

void PresentPass::Execute(FGContext* context) {
    assert(mFinalTexture != InvalidFGResourceId);
    context->CommandList->Present(dynamic_cast<DeviceTexture2D*>(GetTransientResource(context, mFinalTexture)));
}

}  // namespace MRendererHip
