// Scene.h — the slice of Scene / Camera / RenderScheduler the shading path consumes.
//
// Camera        <- Engine/Include/Renderer/Camera.h:9-50, Engine/Source/Renderer/Camera.cpp:5-12,
//                  MathLib.cpp:35-68 (ProjectionMatrix1), MathLib.h:656-671 (FromEulerAngle), :786-811 (QuickInverse)
// SceneLight    <- Engine/Include/Renderer/Scene.h:115-184, Engine/Source/Renderer/Scene.cpp:122-165
// Scene         <- light list + sky box (cube + SH pack); meshes/materials are out of scope, the
//                  G-buffer a frame starts from is supplied as planes (synthetic or captured)
// RenderScheduler <- Engine/Source/Renderer/RenderScheduler.cpp:16-47
#pragma once
#include <cmath>
#include <memory>
#include <vector>

#include "HipCommandList.h"
#include "LightCull.h"

namespace MRendererHip {

struct Matrix4x4 {
    float m[16];   // row-major, M*v
    static Matrix4x4 Identity() {
        Matrix4x4 r{};
        r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.0f;
        return r;
    }
    float& At(int r, int c) { return m[r * 4 + c]; }
    float At(int r, int c) const { return m[r * 4 + c]; }
};

Matrix4x4 operator*(const Matrix4x4& a, const Matrix4x4& b);
Matrix4x4 ProjectionMatrix1(float fov, float ratio, float near_z, float far_z);
Matrix4x4 QuickInverse(const Matrix4x4& m);
Matrix4x4 Inverse(const Matrix4x4& m);

class Camera {
public:
    Camera(float fov, uint32 width, uint32 height, float near_plane, float far_plane)
        : mFov(fov), mRatio((float)width / (float)height), mNear(near_plane), mFar(far_plane),
          mRoll(0), mYaw(0), mPitch(0), mViewSpaceTransform(Matrix4x4::Identity()) {}
    void Move(const Vector3& d) {
        mViewSpaceTransform.m[3] += d.x;
        mViewSpaceTransform.m[7] += d.y;
        mViewSpaceTransform.m[11] += d.z;
    }
    void Rotate(float roll, float yaw, float pitch);
    Matrix4x4 GetWorldMatrix() const { return mViewSpaceTransform; }
    Matrix4x4 GetLocalSpaceMatrix() const { return QuickInverse(mViewSpaceTransform); }
    Matrix4x4 GetProjectionMatrix() const { return ProjectionMatrix1(mFov, mRatio, mNear, mFar); }
    Vector3 GetTranslation() const { return Vector3{mViewSpaceTransform.m[3], mViewSpaceTransform.m[7], mViewSpaceTransform.m[11]}; }
    float Near() const { return mNear; }
    float Far() const { return mFar; }
    float Fov() const { return mFov; }
    float Ratio() const { return mRatio; }
protected:
    float mFov, mRatio, mNear, mFar, mRoll, mYaw, mPitch;
    Matrix4x4 mViewSpaceTransform;
};

struct PointLightAttenuation { float Radius, ConstantCoefficent, LinearCoefficent, QuadraticCoefficent; };

class SceneLight {
public:
    SceneLight(const Vector3& pos, const Vector3& color, float radius, float intensity)
        : mTranslation(pos), mColor(color), mRadius(radius), mIntensity(intensity), mAttenuation(CaclAttenuationCoefficients(radius)) {}
    Vector3 GetTranslation() const { return mTranslation; }
    const Vector3& GetColor() const { return mColor; }
    float GetRadius() const { return mRadius; }
    float GetIntensity() const { return mIntensity; }
    const PointLightAttenuation& GetAttenuationCoefficients() const { return mAttenuation; }
    // at ~1.81418 * radius the preset attenuation falls below 1/256 (Scene.h:118); Scene.cpp:122-130
    static constexpr float CullingRadiusCoefficient = 1.81418f;
    AABB GetWorldBound() const {
        if (mHasBoundOverride) return mBoundOverride;
        const float r = mRadius * CullingRadiusCoefficient * std::sqrt(mIntensity);
        return AABB{{mTranslation.x - r, mTranslation.y - r, mTranslation.z - r}, {mTranslation.x + r, mTranslation.y + r, mTranslation.z + r}};
    }
    // a scene object with a rotation or a non-unit scale moves its bound with its matrix (Scene.h:31; SceneFile.cpp)
    void SetWorldBound(const AABB& b) { mBoundOverride = b; mHasBoundOverride = true; }
    static PointLightAttenuation CaclAttenuationCoefficients(float radius);   // Scene.cpp:132-165 (step function, quirk Q18)
protected:
    Vector3 mTranslation, mColor;
    float mRadius, mIntensity;
    PointLightAttenuation mAttenuation;
    AABB mBoundOverride{};
    bool mHasBoundOverride = false;
};

// CubeMapResource stand-in: fp32 RGBA cube with mips + the SH pack computed at import time
// (BasicStorage.cpp:201-209 -> SHBaker; here pbr_sh9_project on the GPU).
struct SkyBox {
    std::shared_ptr<DeviceTexture2DArray> Cube;
    pbr_sh_pack SH{};
    DeviceTexture2DArray* Resource() const { return Cube.get(); }
    const pbr_sh_pack& GetSHCoefficients() const { return SH; }
};

// What the rasterizer hands the frame (rasterization itself is out of scope): depth + stencil, and either
// the encoded G-buffer planes A/B/C, or the per-pixel material attributes gbuffer.hlsl::ps_main starts from
// (M0 = albedo.rgb (gamma space) + emission, M1 = normal_ws.xyz + roughness, M2 = metallic, ao, -, -; float4
// each), which GBufferPass then encodes on the GPU (pbr_gbuffer_encode).
struct GBufferSource {
    uint32 Width = 0, Height = 0;
    std::vector<uint32_t> A, B, C;
    std::vector<float> M0, M1, M2;
    std::vector<float> Depth;
    std::vector<uint8_t> Stencil;
    bool HasMaterials() const { return !M0.empty(); }
    bool Dirty = true;   // host copy changed since GBufferPass last uploaded / encoded it
};

class Scene {
public:
    static constexpr float WorldBound = 1000.0f;   // Scene.h:194
    // Scene::AddSceneLight -> AddOctreeElementInternal (Scene.h:246-258)
    void AddLight(const SceneLight& l) {
        if (!mOctreeSceneLight.AddObject(l.GetWorldBound(), (int)mLights.size()))
            throw HipException("Scene::AddLight: the light's culling bound leaves the world box (+-500)");
        mLights.push_back(l);
    }
    void ClearLights() { mLights.clear(); mOctreeSceneLight.Reset(WorldBound); }
    uint32 GetLightCount() const { return (uint32)mLights.size(); }
    const SceneLight* LightAt(size_t i) const { return &mLights[i]; }
    // call fn for each light whose bound intersects the frustum, in octree order (Scene.h:229-237)
    template <class Fn>
    void CullLight(const FrustumVolume& volume, Fn&& fn) {
        mOctreeSceneLight.FrustumCull(volume, [&](int index) { fn(&mLights[index]); });
    }
    void SetSkyBox(std::shared_ptr<SkyBox> s) { mSkyBox = std::move(s); }
    SkyBox* GetSkyBox() const { return mSkyBox.get(); }
    GBufferSource& GBuffer() { return mGBuffer; }
private:
    std::vector<SceneLight> mLights;
    LightOctree mOctreeSceneLight{WorldBound};
    std::shared_ptr<SkyBox> mSkyBox;
    GBufferSource mGBuffer;
};

// The PointLight[] ClusteredPass::Execute commits (DeferredPipeline.cpp:224-241): the lights Scene::CullLight hands out for
// the camera's frustum, in octree order, as {Position, Color, Intensity, Attenuation}.  Returns the count; throws when it
// exceeds `capacity` (the reference ASSERTs GetLightCount() <= MaxSceneLights).
int FillLightBuffer(Scene* scene, const Camera* camera, pbr_light* out, int capacity);

class FrameGraph;
class IRenderPipeline;

class RenderScheduler {
public:
    RenderScheduler(IRenderPipeline* pipeline, int hip_device, uint32 width, uint32 height);
    ~RenderScheduler();
    // fills ConstantBufferGlobal, then FrameGraph::Execute (RenderScheduler.cpp:16-47)
    HipCommandList* ExecutePipeline(Scene* scene, Camera* camera, float delta_time, float total_time);
    FrameGraph* GetFrameGraph() const { return mFrameGraph.get(); }
    HipCommandList* CommandList() const { return mCommandList.get(); }
private:
    std::unique_ptr<HipCommandList> mCommandList;
    std::unique_ptr<FrameGraph> mFrameGraph;
    uint32 mWidth, mHeight;
};

}  // namespace MRendererHip
