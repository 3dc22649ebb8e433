// Scene.cpp — Camera math, attenuation presets and the render scheduler (see Scene.h for citations).
#include "Scene.h"

#include "FrameGraph.h"

namespace MRendererHip {

Matrix4x4 operator*(const Matrix4x4& a, const Matrix4x4& b) {
    Matrix4x4 r{};
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            float acc = 0.0f;
            for (int k = 0; k < 4; k++) acc += a.At(i, k) * b.At(k, j);
            r.At(i, j) = acc;
        }
    return r;
}

Matrix4x4 ProjectionMatrix1(float fov, float ratio, float near_z, float far_z) {   // MathLib.cpp:35-68
    const float htan = std::tan(fov * 0.5f);
    const float r = near_z * ratio * htan, l = -r, t = near_z * htan, b = -t;
    Matrix4x4 ret{};
    ret.At(0, 0) = (2 * near_z) / (r - l);
    ret.At(0, 2) = (r + l) / (l - r);
    ret.At(1, 1) = (2 * near_z) / (t - b);
    ret.At(1, 2) = (t + b) / (b - t);
    ret.At(2, 2) = far_z / (far_z - near_z);
    ret.At(2, 3) = (near_z * far_z) / (near_z - far_z);
    ret.At(3, 2) = 1;
    return ret;
}

Matrix4x4 QuickInverse(const Matrix4x4& m) {   // MathLib.h:786-811: M = R*S + T
    float sc[3];
    for (int c = 0; c < 3; c++) sc[c] = std::sqrt(m.At(0, c) * m.At(0, c) + m.At(1, c) * m.At(1, c) + m.At(2, c) * m.At(2, c));
    float inv[3][3];
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) inv[r][c] = (m.At(c, r) / sc[r]) * (1.0f / sc[c]);   // transpose(R) scaled by 1/scale per column
    Matrix4x4 out{};
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) out.At(r, c) = inv[r][c];
        out.At(r, 3) = -((inv[r][0] * m.m[3] + inv[r][1] * m.m[7]) + inv[r][2] * m.m[11]);
    }
    out.At(3, 3) = 1;
    return out;
}

Matrix4x4 Inverse(const Matrix4x4& a) {   // general 4x4 (InvProjection only; not used by the shading kernels)
    double m[4][8];
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) { m[r][c] = a.At(r, c); m[r][4 + c] = r == c; }
    for (int c = 0; c < 4; c++) {
        int p = c;
        for (int r = c + 1; r < 4; r++) if (std::fabs(m[r][c]) > std::fabs(m[p][c])) p = r;
        for (int k = 0; k < 8; k++) std::swap(m[c][k], m[p][k]);
        double d = m[c][c];
        for (int k = 0; k < 8; k++) m[c][k] /= d;
        for (int r = 0; r < 4; r++)
            if (r != c) { double f = m[r][c]; for (int k = 0; k < 8; k++) m[r][k] -= f * m[c][k]; }
    }
    Matrix4x4 out{};
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) out.At(r, c) = (float)m[r][4 + c];
    return out;
}

void Camera::Rotate(float roll, float yaw, float pitch) {   // Camera.cpp:5-12 + MathLib.h:656-671
    mRoll += roll; mYaw += yaw; mPitch += pitch;
    // SetRotation(FromEulerAngle(mRoll, mYaw, mPitch)): the values land on FromEulerAngle's (yaw, pitch, roll) parameters
    const float ca = std::cos(mRoll), sa = std::sin(mRoll), cb = std::cos(mYaw), sb = std::sin(mYaw), cc = std::cos(mPitch), sc = std::sin(mPitch);
    const float rot[3][3] = {{ca * cb, ca * sb * sc - sa * cc, ca * sb * cc + sa * sc},
                             {sa * cb, sa * sb * sc + ca * cc, sa * sb * cc - ca * sc},
                             {-sb, cb * sc, cb * cc}};
    float scale[3];
    for (int c = 0; c < 3; c++)
        scale[c] = std::sqrt(mViewSpaceTransform.At(0, c) * mViewSpaceTransform.At(0, c) + mViewSpaceTransform.At(1, c) * mViewSpaceTransform.At(1, c) +
                             mViewSpaceTransform.At(2, c) * mViewSpaceTransform.At(2, c));
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) mViewSpaceTransform.At(r, c) = rot[r][c] * scale[c];
}

PointLightAttenuation SceneLight::CaclAttenuationCoefficients(float radius) {
    static constexpr PointLightAttenuation presets[] = {   // Scene.h:126-142
        {0.1f, 1.0f, 45.0f, 7500.0f}, {1.0f, 1.0f, 4.5f, 75.0f}, {7.0f, 1.0f, 0.7f, 1.8f}, {13.0f, 1.0f, 0.35f, 0.44f},
        {20.0f, 1.0f, 0.22f, 0.2f}, {32.0f, 1.0f, 0.14f, 0.07f}, {50.0f, 1.0f, 0.09f, 0.032f}, {65.0f, 1.0f, 0.07f, 0.017f},
        {100.0f, 1.0f, 0.045f, 0.0075f}, {160.0f, 1.0f, 0.027f, 0.0028f}, {200.0f, 1.0f, 0.022f, 0.0019f},
        {325.0f, 1.0f, 0.014f, 0.0007f}, {600.0f, 1.0f, 0.007f, 0.0002f}};
    constexpr int n = (int)(sizeof(presets) / sizeof(presets[0]));
    for (int i = 0; i < n - 1; i++) {
        // `radius >= P[i].Radius && radius <= P[i].Radius` only holds on equality, where k = 0:
        // both reachable branches return P[i]'s coefficients
        if (radius <= presets[i].Radius) return PointLightAttenuation{radius, presets[i].ConstantCoefficent, presets[i].LinearCoefficent, presets[i].QuadraticCoefficent};
    }
    return presets[n - 1];
}

int FillLightBuffer(Scene* scene, const Camera* camera, pbr_light* out, int capacity) {
    const Matrix4x4 view_projection = camera->GetProjectionMatrix() * camera->GetLocalSpaceMatrix();
    const FrustumVolume volume = FrustumVolume::FromMatrix(view_projection.m);
    int i = 0;
    scene->CullLight(volume, [&](SceneLight* light) {
        if (i >= capacity) throw HipException("light buffer: more lights pass the frustum cull than the buffer holds");
        const Vector3 p = light->GetTranslation(), c = light->GetColor();
        const PointLightAttenuation& a = light->GetAttenuationCoefficients();
        out[i++] = pbr_light{{p.x, p.y, p.z}, {c.x, c.y, c.z}, light->GetIntensity(), a.Radius, a.ConstantCoefficent, a.LinearCoefficent, a.QuadraticCoefficent};
    });
    return i;
}

RenderScheduler::RenderScheduler(IRenderPipeline* pipeline, int hip_device, uint32 width, uint32 height) : mWidth(width), mHeight(height) {
    mCommandList = std::make_unique<HipCommandList>(hip_device);
    mFrameGraph = std::make_unique<FrameGraph>(pipeline);
    mFrameGraph->Setup();
    mFrameGraph->Compile();
}
RenderScheduler::~RenderScheduler() = default;

HipCommandList* RenderScheduler::ExecutePipeline(Scene* scene, Camera* camera, float delta_time, float total_time) {
    mCommandList->BeginFrame();
    if (scene) {
        ConstantBufferGlobal g{};
        if (scene->GetSkyBox()) g.SkyBoxSH = scene->GetSkyBox()->GetSHCoefficients();
        const Matrix4x4 inv_view = camera->GetWorldMatrix(), view = camera->GetLocalSpaceMatrix(), proj = camera->GetProjectionMatrix(), inv_proj = Inverse(proj);
        std::memcpy(g.InvView, inv_view.m, 64);
        std::memcpy(g.View, view.m, 64);
        std::memcpy(g.Projection, proj.m, 64);
        std::memcpy(g.InvProjection, inv_proj.m, 64);
        const Vector3 p = camera->GetTranslation();
        g.CameraPos[0] = p.x; g.CameraPos[1] = p.y; g.CameraPos[2] = p.z;
        g.Ratio = camera->Ratio();
        g.Resolution[0] = (float)mWidth; g.Resolution[1] = (float)mHeight;
        g.Near = camera->Near(); g.Far = camera->Far(); g.Fov = camera->Fov();
        g.DeltaTime = delta_time; g.Time = total_time;
        mCommandList->SetGlobalConstant(g);
        mFrameGraph->Execute(mCommandList.get(), scene, camera);
    }
    mCommandList->EndFrame();
    return mCommandList.get();
}

}  // namespace MRendererHip
