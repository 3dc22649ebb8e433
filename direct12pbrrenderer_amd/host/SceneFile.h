// SceneFile.h — the light records of a reference scene file ("accept Asset/Scene/main.json lights verbatim").
//
// Reference: a Scene is a json resource (Scene.h:192 `ResourceFormat = EResourceFormat_Json`) whose reflected fields
// (ReflectionDef.h:119-149) are written by the generic serializer (Serialization.h:180-236): a class is an object of its
// serializable members, the base class sits under the key "@<Base>", a Vector3 is {"x","y","z"}.  One element of
// "mSceneLight" is therefore
//     {"@SceneObject": {"mName", "mTranslation", "mRotation", "mScale"}, "mColor", "mRadius", "mIntensity"}
// and SceneLight::PostDeserialized (Scene.cpp:115-120) derives the attenuation preset from mRadius; Scene::PostDeserialized
// (:83-99) then inserts the lights into the loose octree in file order.  Models, materials and the sky-box path of the file
// are the rasterizer's and the asset loader's business (out of scope); they are skipped, not interpreted.
//
// nlohmann/json is vendored by the reference, not by this repo: the reader below is a small recursive-descent parser of
// RFC 8259 json, enough for files the reference's serializer writes (and strict about what it does not understand).
#pragma once
#include <string>
#include <vector>

#include "Scene.h"

namespace MRendererHip {

struct SceneLightRecord {
    std::string Name;
    Vector3 Translation{0, 0, 0}, Rotation{0, 0, 0}, Scale{1, 1, 1}, Color{1, 1, 1};   // defaults: SceneObject() / SceneLight() (Scene.cpp:7-11, Scene.h:147-153)
    float Radius = 1.0f, Intensity = 1.0f;
};

// Parses a whole scene file held in memory and returns its "mSceneLight" records in file order.
// Throws HipException with position and reason on malformed json or a record of another shape.
std::vector<SceneLightRecord> ParseSceneLights(const char* text, size_t bytes);
std::vector<SceneLightRecord> LoadSceneLights(const std::string& path);

// Scene::PostDeserialized for the light list: clears the scene's lights and adds the records in file order.
// A light's culling bound is a cube around the origin (Scene.cpp:122-130) moved by the object's matrix (Scene.h:29 GetWorldBound =
// matrix * local bound; SceneObject::PostDeserialized, Scene.cpp:31-36: FromEulerAngle(mRotation in degrees) with its columns scaled
// by mScale, then the translation) — and `matrix * AABB` transforms the TWO CORNERS only and takes their component-wise min / max
// (MathLib.cpp:5-10), so a rotated light's bound is the box spanned by its two rotated corners, not the rotated cube's hull; restated
// as the reference computes it (tests/test_host.py compares with a numpy restatement, cull membership included).
// Stricter than the reference: the reader refuses radius <= 0 and intensity < 0 (the reference would take them and cull with a
// degenerate or NaN bound — sqrt of a negative intensity); every light of Asset/Scene/main.json passes.
void AddSceneLights(Scene* scene, const std::vector<SceneLightRecord>& records);

}  // namespace MRendererHip
