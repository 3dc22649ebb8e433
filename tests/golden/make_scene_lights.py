#!/usr/bin/env python3
"""Writes tests/golden/scene_lights.npz: the `mSceneLight` records of the reference's scene asset as DATA.

Build container only (it reads /root/reference, which does not travel to the GPU box):

    python tests/golden/make_scene_lights.py

Source: /root/reference/DeferredRendering/Asset/Scene/main.json — a json resource written by the reference's serializer
(Engine/Include/Renderer/Scene.h:192, Engine/Include/Utils/ReflectionDef.h:119-149): per light the object's name,
translation, rotation, scale and the light's colour, radius and intensity.  Nothing is computed here; the attenuation
preset and the culling order are what the tests derive from these records (tests/test_host.py, tests/test_host_graph.py).
"""
import json
import os

import numpy as np

SRC = "/root/reference/DeferredRendering/Asset/Scene/main.json"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "scene_lights.npz")


def vec(d):
    return [float(d["x"]), float(d["y"]), float(d["z"])]


def main():
    doc = json.load(open(SRC))
    recs = doc["mSceneLight"]
    obj = [r["@SceneObject"] for r in recs]
    np.savez(OUT,
             name=np.array([o["mName"] for o in obj]),
             translation=np.float32([vec(o["mTranslation"]) for o in obj]),
             rotation=np.float32([vec(o["mRotation"]) for o in obj]),
             scale=np.float32([vec(o["mScale"]) for o in obj]),
             color=np.float32([vec(r["mColor"]) for r in recs]),
             radius=np.float32([r["mRadius"] for r in recs]),
             intensity=np.float32([r["mIntensity"] for r in recs]),
             source=np.array("DeferredRendering/Asset/Scene/main.json: mSceneLight"))
    print(f"{OUT}: {len(recs)} lights")


if __name__ == "__main__":
    main()
