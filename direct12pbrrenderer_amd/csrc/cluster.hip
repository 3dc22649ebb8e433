// cluster.hip — clustered-light set-up: frustum-cluster AABBs and per-cluster light lists.
//
// Reference: clustered_compute.hlsl:8-42 and clustered_culling.hlsl:11-41, both dispatched as ONE
// 24x16-thread group whose threads walk 8 z-slices x NumLight lights serially
// (DeferredPipeline.cpp:253-256).  On MI355X that shape is pure latency (one CU, 384 serial
// chains), so the cull is re-cut: one 64-lane wave per cluster tests 64 lights per step and
// compacts the hits with a ballot + prefix popcount, which keeps the reference's
// "first 32 hits in ascending light index" order exactly.
#include "pbr_internal.hpp"
#include "pbr_device.hpp"

using namespace pbr;

struct ClusterParams {
    float Near, Far, Ratio, Fov;
    float View[12];   // rows 0..2 of the row-major view matrix
};

__device__ __forceinline__ int cluster_index3(int x, int y, int z) {   // clustered.hlsli:40-43
    return z + x * PBR_CLUSTER_Z + y * PBR_CLUSTER_X * PBR_CLUSTER_Z;
}

// view-space AABB of cluster t (clustered_compute.hlsl:8-42)
__device__ __forceinline__ void cluster_bounds(const ClusterParams& p, int t, float mn[3], float mx[3]) {
    const int z = t % PBR_CLUSTER_Z, tx = (t / PBR_CLUSTER_Z) % PBR_CLUSTER_X, ty = t / (PBR_CLUSTER_Z * PBR_CLUSTER_X);
    const float htan = tanf(p.Fov / 2.0f);
    const float znear = p.Near * powf(p.Far / p.Near, (float)z / (float)PBR_CLUSTER_Z);
    const float zfar = p.Near * powf(p.Far / p.Near, (float)(z + 1) / (float)PBR_CLUSTER_Z);
    const float minx = 2.0f * (float)tx / (float)PBR_CLUSTER_X - 1.0f, miny = 2.0f * (float)ty / (float)PBR_CLUSTER_Y - 1.0f;
    const float maxx = 2.0f * (float)(tx + 1) / (float)PBR_CLUSTER_X - 1.0f, maxy = 2.0f * (float)(ty + 1) / (float)PBR_CLUSTER_Y - 1.0f;
    // zplane_intersection: ray = (ndc.x*Ratio*tan, ndc.y*tan, 1)*Near; return ray * (view_z / ray.z)
    auto zplane = [&](float nx, float ny, float vz) {
        V3 ray = v3(nx * p.Ratio * htan, ny * htan, 1.0f) * p.Near;
        float tt = vz / ray.z;
        return ray * tt;
    };
    V3 min_near = zplane(minx, miny, znear), min_far = zplane(minx, miny, zfar);
    V3 max_near = zplane(maxx, maxy, znear), max_far = zplane(maxx, maxy, zfar);
    mn[0] = fminf(min_near.x, min_far.x); mn[1] = fminf(min_near.y, min_far.y); mn[2] = fminf(min_near.z, min_far.z);
    mx[0] = fmaxf(max_near.x, max_far.x); mx[1] = fmaxf(max_near.y, max_far.y); mx[2] = fmaxf(max_near.z, max_far.z);
}

// grid 12 x block 256: one thread per cluster (3 072); cluster t sits at index t (cluster_index3 is the same order)
__global__ __launch_bounds__(256) void k_cluster_build(ClusterParams p, pbr_cluster* __restrict__ clusters) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= PBR_NUM_CLUSTERS) return;
    float mn[3], mx[3];
    cluster_bounds(p, t, mn, mx);
    pbr_cluster* c = clusters + t;
    c->MinBound[0] = mn[0]; c->MinBound[1] = mn[1]; c->MinBound[2] = mn[2];
    c->MaxBound[0] = mx[0]; c->MaxBound[1] = mx[1]; c->MaxBound[2] = mx[2];
    c->NumLights = 0;
}

// The sphere/AABB test feeds an integer result (the light list), so it is evaluated with the
// exact operation sequence of the shader, un-contracted (no FMA) and with IEEE sqrt.
__device__ __forceinline__ bool light_hits(const ClusterParams& p, const pbr_light& l, const float* mn, const float* mx) {
#pragma clang fp contract(off)
    const float px = ((p.View[0] * l.Position[0] + p.View[1] * l.Position[1]) + p.View[2] * l.Position[2]) + p.View[3];
    const float py = ((p.View[4] * l.Position[0] + p.View[5] * l.Position[1]) + p.View[6] * l.Position[2]) + p.View[7];
    const float pz = ((p.View[8] * l.Position[0] + p.View[9] * l.Position[1]) + p.View[10] * l.Position[2]) + p.View[11];
    const float radius = l.Radius * 1.814f * sqrtf(l.Intensity);   // Q19: HLSL constant 1.814
    const float dx = px - fminf(fmaxf(px, mn[0]), mx[0]);
    const float dy = py - fminf(fmaxf(py, mn[1]), mx[1]);
    const float dz = pz - fminf(fmaxf(pz, mn[2]), mx[2]);
    return (dx * dx + dy * dy) + dz * dz < radius * radius;
}

// grid 3072/4 x block 256 (4 waves, one cluster per wave).  BUILD: both dispatches of ClusteredPass::Execute in one
// launch — the wave computes its cluster's bounds itself (every lane the same values) instead of reading them back.
template <bool BUILD>
__global__ __launch_bounds__(256) void k_cluster_cull(ClusterParams p, const pbr_light* __restrict__ lights, int n,
                                                        pbr_cluster* __restrict__ clusters) {
    const int lane = threadIdx.x & 63;
    const int ci = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ci >= PBR_NUM_CLUSTERS) return;   // wave-uniform
    pbr_cluster* c = clusters + ci;
    float mn[3], mx[3];
    int count = 0;
    if (BUILD) {
        cluster_bounds(p, ci, mn, mx);
        if (lane < 3) { c->MinBound[lane] = mn[lane]; c->MaxBound[lane] = mx[lane]; }
    } else {
        mn[0] = c->MinBound[0]; mn[1] = c->MinBound[1]; mn[2] = c->MinBound[2];
        mx[0] = c->MaxBound[0]; mx[1] = c->MaxBound[1]; mx[2] = c->MaxBound[2];
        count = c->NumLights;   // continues a partially filled list like the reference loop condition
        count = min(max(count, 0), PBR_MAX_LIGHTS_PER_CLUSTER);
    }
    for (int base = 0; base < n && count < PBR_MAX_LIGHTS_PER_CLUSTER; base += 64) {   // wave-uniform loop
        const int i = base + lane;
        bool hit = false;
        if (i < n) hit = light_hits(p, lights[i], mn, mx);
        const unsigned long long mask = __ballot(hit);
        const int pos = count + __popcll(mask & ((1ull << lane) - 1ull));
        if (hit && pos < PBR_MAX_LIGHTS_PER_CLUSTER) c->LightIndex[pos] = i;
        count = min(count + __popcll(mask), PBR_MAX_LIGHTS_PER_CLUSTER);
    }
    if (lane == 0) c->NumLights = count;
}

static ClusterParams make_params(const pbr_global* g) {
    ClusterParams p;
    p.Near = g->Near; p.Far = g->Far; p.Ratio = g->Ratio; p.Fov = g->Fov;
    for (int i = 0; i < 12; i++) p.View[i] = g->View[i];
    return p;
}

extern "C" {

pbr_status pbr_cluster_build(pbr_ctx* ctx, const pbr_global* g, pbr_cluster* clusters) {
    if (!ctx) return PBR_ERR_INVALID;
    PBR_REQUIRE(ctx, g && clusters, "pbr_cluster_build: null pointer");
    PBR_REQUIRE(ctx, g->Near > 0.0f && g->Far > g->Near, "pbr_cluster_build: need 0 < Near < Far");
    hipLaunchKernelGGL(k_cluster_build, dim3(PBR_NUM_CLUSTERS / 256), dim3(256), 0, ctx->stream, make_params(g), clusters);
    return launched(ctx, "k_cluster_build");
}

pbr_status pbr_cluster_cull(pbr_ctx* ctx, const pbr_global* g, const pbr_light* lights, int n, pbr_cluster* clusters) {
    if (!ctx) return PBR_ERR_INVALID;
    PBR_REQUIRE(ctx, g && clusters, "pbr_cluster_cull: null pointer");
    // ClusteredPass::Execute asserts GetLightCount() <= MaxSceneLights (DeferredPipeline.cpp:222)
    PBR_REQUIRE(ctx, n >= 0 && n <= PBR_MAX_SCENE_LIGHTS, "pbr_cluster_cull: light count out of [0, 1024]");
    PBR_REQUIRE(ctx, n == 0 || lights != nullptr, "pbr_cluster_cull: null lights");
    if (n == 0) return PBR_OK;
    hipLaunchKernelGGL(k_cluster_cull<false>, dim3(PBR_NUM_CLUSTERS / 4), dim3(256), 0, ctx->stream, make_params(g), lights, n, clusters);
    return launched(ctx, "k_cluster_cull");
}

pbr_status pbr_clustered(pbr_ctx* ctx, const pbr_global* g, const pbr_light* lights, int n, pbr_cluster* clusters) {
    if (!ctx) return PBR_ERR_INVALID;
    PBR_REQUIRE(ctx, g && clusters, "pbr_clustered: null pointer");
    PBR_REQUIRE(ctx, g->Near > 0.0f && g->Far > g->Near, "pbr_clustered: need 0 < Near < Far");
    PBR_REQUIRE(ctx, n >= 0 && n <= PBR_MAX_SCENE_LIGHTS, "pbr_clustered: light count out of [0, 1024]");
    PBR_REQUIRE(ctx, n == 0 || lights != nullptr, "pbr_clustered: null lights");
    hipLaunchKernelGGL(k_cluster_cull<true>, dim3(PBR_NUM_CLUSTERS / 4), dim3(256), 0, ctx->stream, make_params(g), lights, n, clusters);
    return launched(ctx, "k_cluster_cull<build>");
}

}  // extern "C"
