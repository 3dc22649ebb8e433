// LightCull.h — CPU-side frustum culling of scene lights.
//
// The light buffer the GPU culls against is filled by ClusteredPass::Execute from
// Scene::CullLight(FrustumVolume, fn) (DeferredPipeline.cpp:224-241), which walks a loose octree of
// light AABBs.  Both WHICH lights arrive and IN WHAT ORDER matter downstream: the per-cluster lists keep
// the first 32 hits in buffer order and the shade sums in list order.  So this file keeps the rules that
// decide membership and order — not the reference's containers:
//   AABB / FrustumVolume      <- Engine/Include/Utils/MathLib.h:962-1082 (strict Contain; GL-style near
//                                plane row3+row2 on a [0,1]-depth projection: kept, it only widens the cull)
//   LightOctree               <- Engine/Include/Utils/LooseOctree.h:18-32 (MaxCapacityToSplit 2, MinNodeSize
//                                0.75^8), :133-182 (insert), :184-244 (SubDivide: child max = parent max or
//                                centre, min = max - half size), :246-254 (FindBestFitChild), :256-277 (cull:
//                                node elements first, then children 0..7, depth first)
//   light world bound         <- Engine/Source/Renderer/Scene.cpp:122-130 (radius * 1.81418 * sqrt(intensity))
// Elements of a node are visited in insertion order (the reference's pool allocator hands out blocks in
// ascending address order and this path never frees one); re-inserting a split node's elements keeps
// their relative order, as the reference's move-and-re-add does.
#pragma once
#include <cmath>
#include <cstdint>
#include <vector>

namespace MRendererHip {

struct Vector3 { float x, y, z; };

struct AABB {
    Vector3 Min{}, Max{};
    float Width() const { return Max.x - Min.x; }
    Vector3 Size() const { return Vector3{Max.x - Min.x, Max.y - Min.y, Max.z - Min.z}; }
    Vector3 Center() const { return Vector3{(Min.x + Max.x) * 0.5f, (Min.y + Max.y) * 0.5f, (Min.z + Max.z) * 0.5f}; }
    bool Contain(const AABB& b) const {   // strict on every side (MathLib.h:1003-1007)
        return b.Min.x > Min.x && b.Min.y > Min.y && b.Min.z > Min.z && b.Max.x < Max.x && b.Max.y < Max.y && b.Max.z < Max.z;
    }
};

struct FrustumVolume {
    float Planes[6][4];   // (N, D): inside when dot(N, P) + D >= 0 for all six
    // rows of a row-major 4x4 that maps column vectors (Projection * View)
    static FrustumVolume FromMatrix(const float m[16]) {
        FrustumVolume v{};
        for (int c = 0; c < 4; c++) {
            v.Planes[0][c] = m[12 + c] + m[0 + c];
            v.Planes[1][c] = m[12 + c] - m[0 + c];
            v.Planes[2][c] = m[12 + c] + m[4 + c];
            v.Planes[3][c] = m[12 + c] - m[4 + c];
            v.Planes[4][c] = m[12 + c] + m[8 + c];
            v.Planes[5][c] = m[12 + c] - m[8 + c];
        }
        return v;
    }
    bool Contains(const AABB& b) const {   // false only when the box is entirely below one plane
        const Vector3 c = b.Center();
        const Vector3 s = b.Size();
        const Vector3 e{s.x * 0.5f, s.y * 0.5f, s.z * 0.5f};
        for (int i = 0; i < 6; i++) {
            const float* p = Planes[i];
            const float half_diagonal_projection = std::fabs(p[0] * e.x) + std::fabs(p[1] * e.y) + std::fabs(p[2] * e.z);
            const float center_distance = p[0] * c.x + p[1] * c.y + p[2] * c.z + p[3] * 1.0f;
            if (center_distance < -half_diagonal_projection) return false;
        }
        return true;
    }
};

class LightOctree {
public:
    explicit LightOctree(float size) { Reset(size); }
    void Reset(float size) {
        const float h = size * 0.5f;
        mNodes.clear();
        mNodes.push_back(Node{AABB{{-h, -h, -h}, {h, h, h}}, -1, {}});
    }
    // false when the bound does not fit strictly inside the world box (the reference ASSERTs)
    bool AddObject(const AABB& bound, int object) { return Insert(0, Element{bound, object}); }

    template <class Fn>
    void FrustumCull(const FrustumVolume& volume, Fn&& fn) const { Cull(volume, 0, fn); }
    size_t NumNodes() const { return mNodes.size(); }

private:
    struct Element { AABB Bound; int Object; };
    struct Node { AABB Bound; int Children; std::vector<Element> Elements; };
    static constexpr size_t MaxCapacityToSplit = 2;
    static float MinNodeSize() { return 0.100112915f; }   // (0.5 * 1.5)^8

    bool Insert(int node, const Element& e) {
        if (!mNodes[node].Bound.Contain(e.Bound)) return false;
        if (mNodes[node].Children < 0) {
            if (mNodes[node].Elements.size() + 1 > MaxCapacityToSplit && mNodes[node].Bound.Width() > MinNodeSize()) {
                SubDivide(node);
                std::vector<Element> pool;
                pool.swap(mNodes[node].Elements);
                for (const Element& old : pool) Insert(node, old);
                return Insert(node, e);
            }
            mNodes[node].Elements.push_back(e);
            return true;
        }
        const int child = BestFitChild(node, e.Bound);
        if (!Insert(child, e)) mNodes[node].Elements.push_back(e);
        return true;
    }
    void SubDivide(int node) {
        const int first = (int)mNodes.size();
        const AABB bound = mNodes[node].Bound;
        const Vector3 c = bound.Center();
        const Vector3 s = bound.Size();
        const Vector3 half{s.x * 0.5f, s.y * 0.5f, s.z * 0.5f};
        for (int i = 0; i < 8; i++) {
            const Vector3 mx{(i & 1) ? bound.Max.x : c.x, (i & 2) ? bound.Max.y : c.y, (i & 4) ? bound.Max.z : c.z};
            mNodes.push_back(Node{AABB{{mx.x - half.x, mx.y - half.y, mx.z - half.z}, mx}, -1, {}});
        }
        mNodes[node].Children = first;
    }
    int BestFitChild(int node, const AABB& b) const {
        const Vector3 bc = b.Center(), nc = mNodes[node].Bound.Center();
        int i = 0;
        if (bc.x - nc.x >= 0) i |= 1;
        if (bc.y - nc.y >= 0) i |= 2;
        if (bc.z - nc.z >= 0) i |= 4;
        return i + mNodes[node].Children;
    }
    template <class Fn>
    void Cull(const FrustumVolume& v, int node, Fn& fn) const {
        const Node& n = mNodes[node];
        if (!v.Contains(n.Bound)) return;
        for (const Element& e : n.Elements)
            if (v.Contains(e.Bound)) fn(e.Object);
        if (n.Children >= 0)
            for (int i = 0; i < 8; i++) Cull(v, n.Children + i, fn);
    }
    std::vector<Node> mNodes;
};

}  // namespace MRendererHip
