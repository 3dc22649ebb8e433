"""Host-side mirrors of the reference's per-frame CPU logic that feeds the shading path.

* Camera            <- Engine/Include/Renderer/Camera.h:9-50, Engine/Source/Renderer/Camera.cpp:5-12
* projection_matrix1 <- Engine/Source/Utils/MathLib.cpp:35-68 (ndc.z in [0,1], left-handed)
* from_euler_angle  <- Engine/Include/Utils/MathLib.h:656-671
* quick_inverse     <- Engine/Include/Utils/MathLib.h:786-811
* make_global       <- RenderScheduler::ExecutePipeline, Engine/Source/Renderer/RenderScheduler.cpp:22-38
* attenuation presets / CaclAttenuationCoefficients <- Engine/Include/Renderer/Scene.h:126-142,
  Engine/Source/Renderer/Scene.cpp:132-165

Everything is evaluated in float32 like the reference's Vector/Matrix classes.
"""
import ctypes as C
import math

import numpy as np

from .structs import LIGHT_DTYPE, Global, ShPack

f32 = np.float32
PI = f32(3.14159265359)


def projection_matrix1(fov, ratio, near_z, far_z):
    fov, ratio, near_z, far_z = f32(fov), f32(ratio), f32(near_z), f32(far_z)
    htan = f32(math.tan(float(fov * f32(0.5))))
    r = near_z * ratio * htan
    l = -r
    t = near_z * htan
    b = -t
    m = np.zeros((4, 4), dtype=f32)
    m[0, 0] = (f32(2) * near_z) / (r - l)
    m[0, 2] = (r + l) / (l - r)
    m[1, 1] = (f32(2) * near_z) / (t - b)
    m[1, 2] = (t + b) / (b - t)
    m[2, 2] = far_z / (far_z - near_z)
    m[2, 3] = (near_z * far_z) / (near_z - far_z)
    m[3, 2] = f32(1)
    return m


def from_euler_angle(yaw, pitch, roll):
    """Matrix3x3::FromEulerAngle(yaw, pitch, roll) — parameter NAMES as in MathLib.h:656."""
    ca, sa = f32(math.cos(yaw)), f32(math.sin(yaw))
    cb, sb = f32(math.cos(pitch)), f32(math.sin(pitch))
    cc, sc = f32(math.cos(roll)), f32(math.sin(roll))
    return np.array([
        [ca * cb, ca * sb * sc - sa * cc, ca * sb * cc + sa * sc],
        [sa * cb, sa * sb * sc + ca * cc, sa * sb * cc - ca * sc],
        [-sb, cb * sc, cb * cc]], dtype=f32)


def quick_inverse(m):
    """Matrix4x4::QuickInverse: inverse of a rotation*scale + translation transform."""
    scale = np.sqrt((m[:3, :3].astype(f32) ** 2).sum(axis=0, dtype=f32)).astype(f32)
    rot = (m[:3, :3] / scale[None, :]).astype(f32).T
    inv_scale = (f32(1) / scale).astype(f32)
    inv_m = (rot * inv_scale[None, :]).astype(f32)
    tr = m[:3, 3]
    inv_t = np.array([(inv_m[i, 0] * tr[0] + inv_m[i, 1] * tr[1]) + inv_m[i, 2] * tr[2] for i in range(3)], dtype=f32)
    out = np.zeros((4, 4), dtype=f32)
    out[:3, :3] = inv_m
    out[:3, 3] = -inv_t
    out[3, 3] = f32(1)
    return out


class Camera:
    """Camera.h:9-50.  The transform is view-space -> world-space (row-major, M*v)."""

    def __init__(self, fov, width, height, near_plane, far_plane):
        self.fov = f32(fov)
        self.ratio = f32(width) / f32(height)
        self.near = f32(near_plane)
        self.far = f32(far_plane)
        self.roll = self.yaw = self.pitch = 0.0
        self.transform = np.eye(4, dtype=f32)

    def move(self, delta):
        self.transform[:3, 3] += np.asarray(delta, dtype=f32)

    def rotate(self, roll, yaw, pitch):
        # Camera.cpp:5-12: SetRotation(FromEulerAngle(mRoll, mYaw, mPitch)) — the arguments land on
        # FromEulerAngle's (yaw, pitch, roll) parameters in that order.
        self.roll += roll
        self.yaw += yaw
        self.pitch += pitch
        scale = np.sqrt((self.transform[:3, :3] ** 2).sum(axis=0, dtype=f32)).astype(f32)
        self.transform[:3, :3] = from_euler_angle(self.roll, self.yaw, self.pitch) * scale[None, :]

    def world_matrix(self):
        return self.transform.copy()

    def local_space_matrix(self):
        return quick_inverse(self.transform)

    def projection_matrix(self):
        return projection_matrix1(self.fov, self.ratio, self.near, self.far)

    def translation(self):
        return self.transform[:3, 3].copy()

    @staticmethod
    def reference_default(width, height):
        """App.cpp:99-101: Fov 0.333*PI, Near 0.1, Far 1000, Move(0,3,10), Rotate(0, PI, 0)."""
        cam = Camera(f32(0.333) * PI, width, height, 0.1, 1000.0)
        cam.move((0.0, 3.0, 10.0))
        cam.rotate(0.0, float(PI), 0.0)
        return cam


def make_global(camera, width, height, sh_pack=None, delta_time=1.0 / 60.0, time=0.0):
    """Fill ConstantBufferGlobal the way RenderScheduler::ExecutePipeline does (RenderScheduler.cpp:22-38)."""
    g = Global()
    if sh_pack is not None:
        arr = np.asarray(sh_pack, dtype=f32).reshape(28)
        C.memmove(C.byref(g.SkyBoxSH), arr.ctypes.data, 112)
    proj = camera.projection_matrix()
    mats = {
        "InvView": camera.world_matrix(),
        "View": camera.local_space_matrix(),
        "Projection": proj,
        "InvProjection": np.linalg.inv(proj.astype(np.float64)).astype(f32),
    }
    for name, m in mats.items():
        getattr(g, name)[:] = [float(v) for v in np.ascontiguousarray(m, dtype=f32).reshape(16)]
    g.CameraPos[:] = [float(v) for v in camera.translation()]
    g.Ratio = float(camera.ratio)
    g.Resolution[:] = [float(width), float(height)]
    g.Near = float(camera.near)
    g.Far = float(camera.far)
    g.Fov = float(camera.fov)
    g.DeltaTime = float(f32(delta_time))
    g.Time = float(f32(time))
    return g


# Scene.h:126-142
ATTENUATION_PRESETS = [
    (0.1, 1.0, 45.0, 7500.0), (1.0, 1.0, 4.5, 75.0), (7.0, 1.0, 0.7, 1.8), (13.0, 1.0, 0.35, 0.44),
    (20.0, 1.0, 0.22, 0.2), (32.0, 1.0, 0.14, 0.07), (50.0, 1.0, 0.09, 0.032), (65.0, 1.0, 0.07, 0.017),
    (100.0, 1.0, 0.045, 0.0075), (160.0, 1.0, 0.027, 0.0028), (200.0, 1.0, 0.022, 0.0019),
    (325.0, 1.0, 0.014, 0.0007), (600.0, 1.0, 0.007, 0.0002),
]


def attenuation_coefficients(radius):
    """SceneLight::CaclAttenuationCoefficients (Scene.cpp:132-165).

    The interpolation branch tests `radius >= P[i].Radius && radius <= P[i].Radius` on the SAME
    preset, so it only fires on exact equality (k = 0): the function is a step function (quirk Q18).
    Returns (Radius, C0, C1, C2).
    """
    radius = float(f32(radius))
    for i in range(len(ATTENUATION_PRESETS) - 1):
        lower = ATTENUATION_PRESETS[i]
        if radius < float(f32(lower[0])):
            return (radius, lower[1], lower[2], lower[3])
        if radius == float(f32(lower[0])):
            return (radius, lower[1], lower[2], lower[3])   # k == 0 -> lower
    last = ATTENUATION_PRESETS[-1]
    return (last[0], last[1], last[2], last[3])


def make_lights(positions, colors, radius, intensity):
    """PointLight records as ClusteredPass::Execute uploads them (DeferredPipeline.cpp:225-250)."""
    positions = np.asarray(positions, dtype=f32).reshape(-1, 3)
    n = positions.shape[0]
    lights = np.zeros(n, dtype=LIGHT_DTYPE)
    lights["Position"] = positions
    lights["Color"] = np.asarray(colors, dtype=f32).reshape(-1, 3)
    lights["Intensity"] = f32(intensity)
    r, c0, c1, c2 = attenuation_coefficients(radius)
    lights["Radius"], lights["C0"], lights["C1"], lights["C2"] = f32(r), f32(c0), f32(c1), f32(c2)
    return lights


def sh_pack_struct(arr28):
    p = ShPack()
    a = np.asarray(arr28, dtype=f32).reshape(28)
    C.memmove(C.byref(p), a.ctypes.data, 112)
    return p


def scene_file_text(recs, extra_members=True):
    """A scene file in the reference serializer's shape (Serialization.h:180-236: base class under "@<Base>", Vector3 as {x, y, z})
    holding these light records (dict-like of arrays: name, translation, rotation, scale, color, radius, intensity) — what
    pbrh_load_scene_lights reads; generated input, not the reference's asset."""
    import json

    def v(a):
        return {"x": float(a[0]), "y": float(a[1]), "z": float(a[2])}
    lights = [{"@SceneObject": {"mName": str(recs["name"][i]), "mTranslation": v(recs["translation"][i]),
                                "mRotation": v(recs["rotation"][i]), "mScale": v(recs["scale"][i])},
               "mColor": v(recs["color"][i]), "mRadius": float(recs["radius"][i]), "mIntensity": float(recs["intensity"][i])}
              for i in range(len(recs["radius"]))]
    doc = {"@IResource": None, "mSceneLight": lights}
    if extra_members:   # members of the file the light path must skip over
        doc["mSceneModel"] = [{"@SceneObject": {"mName": "m\u00e9sh \"0\"", "mTranslation": v([0, 0, 0]), "mRotation": v([0, 90, 0]),
                                                "mScale": v([0.1, 0.1, 0.1])}, "mModelFilePath": "Asset/Model/x.json"}]
        doc["mSkyBoxPath"] = "Asset/SkyBox/none"
    return json.dumps(doc, indent=1)
