"""Test-side restatement (numpy float32) of the reference's CPU light cull — which lights reach the light
buffer and in what order — used to check direct12pbrrenderer_amd/host/LightCull.h.  TEST INFRASTRUCTURE ONLY.

Rules restated from the reference (read as text): Engine/Include/Utils/LooseOctree.h:18-32,133-277 (insert /
SubDivide / FindBestFitChild / FrustumCullInternal), Engine/Include/Utils/MathLib.h:962-1082 (AABB::Contain strict,
FrustumVolume::FromMatrix / Contains(AABB)), Engine/Source/Renderer/Scene.cpp:122-130 (light bound),
Engine/Source/Renderer/Pipeline/DeferredPipeline.cpp:224-241 (frustum of Projection * View).
"""
import numpy as np

F = np.float32
WORLD = F(1000.0)
MIN_NODE = F(0.75) ** 8
CULL_COEFF = F(1.81418)


def light_bound(pos, radius, intensity):
    r = F(radius) * CULL_COEFF * np.sqrt(F(intensity))
    p = np.asarray(pos, dtype=F)
    return p - r, p + r


def contain(nmin, nmax, bmin, bmax):
    return bool(np.all(bmin > nmin) and np.all(bmax < nmax))


class Octree:
    def __init__(self):
        h = WORLD * F(0.5)
        self.nodes = [dict(min=np.full(3, -h, F), max=np.full(3, h, F), children=-1, elems=[])]

    def add(self, bmin, bmax, obj):
        return self._insert(0, (bmin, bmax, obj))

    def _insert(self, ni, e):
        n = self.nodes[ni]
        if not contain(n["min"], n["max"], e[0], e[1]):
            return False
        if n["children"] < 0:
            if len(n["elems"]) + 1 > 2 and (n["max"][0] - n["min"][0]) > MIN_NODE:
                self._subdivide(ni)
                pool, n["elems"] = n["elems"], []
                for old in pool:
                    self._insert(ni, old)
                return self._insert(ni, e)
            n["elems"].append(e)
            return True
        c = (n["min"] + n["max"]) * F(0.5)
        bc = (e[0] + e[1]) * F(0.5)
        d = bc - c
        child = n["children"] + (1 if d[0] >= 0 else 0) + (2 if d[1] >= 0 else 0) + (4 if d[2] >= 0 else 0)
        if not self._insert(child, e):
            n["elems"].append(e)
        return True

    def _subdivide(self, ni):
        n = self.nodes[ni]
        c = (n["min"] + n["max"]) * F(0.5)
        half = (n["max"] - n["min"]) * F(0.5)
        first = len(self.nodes)
        for i in range(8):
            mx = np.array([n["max"][k] if (i >> k) & 1 else c[k] for k in range(3)], dtype=F)
            self.nodes.append(dict(min=mx - half, max=mx, children=-1, elems=[]))
        n["children"] = first

    def cull(self, planes, ni=0, out=None):
        out = [] if out is None else out
        n = self.nodes[ni]
        if not frustum_contains(planes, n["min"], n["max"]):
            return out
        for bmin, bmax, obj in n["elems"]:
            if frustum_contains(planes, bmin, bmax):
                out.append(obj)
        if n["children"] >= 0:
            for i in range(8):
                self.cull(planes, n["children"] + i, out)
        return out


def matmul44(a, b):
    r = np.zeros((4, 4), F)
    for i in range(4):
        for j in range(4):
            acc = F(0.0)
            for k in range(4):
                acc = F(acc + F(a[i, k] * b[k, j]))
            r[i, j] = acc
    return r


def frustum_planes(vp):
    vp = np.asarray(vp, dtype=F)
    return np.stack([vp[3] + vp[0], vp[3] - vp[0], vp[3] + vp[1], vp[3] - vp[1], vp[3] + vp[2], vp[3] - vp[2]]).astype(F)


def frustum_contains(planes, bmin, bmax):
    c = (bmin + bmax) * F(0.5)
    e = (bmax - bmin) * F(0.5)
    for p in planes:
        half = F(F(abs(F(p[0] * e[0])) + abs(F(p[1] * e[1]))) + abs(F(p[2] * e[2])))
        dist = F(F(F(F(p[0] * c[0]) + F(p[1] * c[1])) + F(p[2] * c[2])) + p[3])
        if dist < -half:
            return False
    return True


def cull_lights(camera, positions, radius, intensity):
    """Indices of the lights the reference hands to the light buffer, in buffer order; None if one leaves the world box."""
    tree = Octree()
    for i, p in enumerate(positions):
        bmin, bmax = light_bound(p, radius[i], intensity[i])
        if not tree.add(bmin, bmax, i):
            return None
    vp = matmul44(camera.projection_matrix().astype(F), camera.local_space_matrix().astype(F))
    return tree.cull(frustum_planes(vp))


def rotated_scaled_bound(pos, rotation_deg, scale, radius, intensity):
    """SceneObject::GetWorldBound of a light whose object carries a rotation / scale (Scene.h:29, Scene.cpp:31-36, MathLib.h:656-670,
    769-782, MathLib.cpp:5-10): matrix = FromEulerAngle(rotation in radians) with its columns scaled, translation in the last column;
    matrix * AABB = component-wise min / max of the two transformed CORNERS of the local cube."""
    a, b, c = (F(F(v) * F(3.14159265359 / 180.0)) for v in rotation_deg)
    ca, sa, cb, sb, cc, sc = F(np.cos(a)), F(np.sin(a)), F(np.cos(b)), F(np.sin(b)), F(np.cos(c)), F(np.sin(c))
    rot = np.array([[ca * cb, ca * sb * sc - sa * cc, ca * sb * cc + sa * sc],
                    [sa * cb, sa * sb * sc + ca * cc, sa * sb * cc - ca * sc],
                    [-sb, cb * sc, cb * cc]], dtype=F)
    m = rot * np.asarray(scale, dtype=F)[None, :]
    r = F(radius) * CULL_COEFF * np.sqrt(F(intensity))
    lo = (m @ np.full(3, -r, F)).astype(F) + np.asarray(pos, F)
    hi = (m @ np.full(3, r, F)).astype(F) + np.asarray(pos, F)
    return np.minimum(lo, hi), np.maximum(lo, hi)


def cull_bounds(camera, bounds):
    """cull_lights for explicit world bounds [(min, max), ...]"""
    tree = Octree()
    for i, (bmin, bmax) in enumerate(bounds):
        if not tree.add(np.asarray(bmin, F), np.asarray(bmax, F), i):
            return None
    vp = matmul44(camera.projection_matrix().astype(F), camera.local_space_matrix().astype(F))
    return tree.cull(frustum_planes(vp))
