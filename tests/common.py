"""Shared deterministic test scenes (inputs are regenerated from seeds; goldens hold outputs only)."""
import numpy as np

from direct12pbrrenderer_amd import scene, synth
from direct12pbrrenderer_amd.structs import Tile

SKY_SIZE = 16
SKY_MIPS = 5
ENV_SIZE = 16
ENV_MIPS = 5
LUT_RES = 32


def host_lib_path():
    """libpbr_host.so; PBR_TEST_HOST_LIB points the CPU tests at the sanitizer build (tools/asan_cpu.sh)."""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    return os.environ.get("PBR_TEST_HOST_LIB") or os.path.join(root, "direct12pbrrenderer_amd", "libpbr_host.so")


def small_ibl(orc):
    """sky cube (with box mips), prefiltered env, LUT and SH pack from the oracle (16^3 / 32^2)."""
    sky = synth.env_cube(SKY_SIZE, SKY_MIPS)
    orc.cube_gen_mips(sky, SKY_SIZE, SKY_MIPS)
    env = orc.prefilter_env(sky, SKY_SIZE, SKY_MIPS, ENV_SIZE, ENV_MIPS)
    lut = orc.brdf_lut(LUT_RES)
    sh = orc.sh9_project(sky, SKY_SIZE)
    return sky, env, lut, sh


def shade_scene(w, h, n_lights, sh, full=None, x0=0, y0=0, rough_min=0, coverage_mask=True):
    """Global constants, lights and a G-buffer tile for a w x h region of a full frame."""
    full_w, full_h = full if full else (w, h)
    cam = scene.Camera.reference_default(full_w, full_h)
    g = scene.make_global(cam, full_w, full_h, sh_pack=sh)
    if n_lights == 0:
        lights = scene.make_lights(np.zeros((0, 3)), np.zeros((0, 3)), 2.0, 10.0)
    elif n_lights == 1:
        lights = synth.reference_scene_light()
    else:
        lights = synth.lights_in_view_box(n_lights, cam)
    gb = synth.gbuffer_tile(x0, y0, w, h, full_w, full_h, rough_min=rough_min, coverage_mask=coverage_mask)
    tile = Tile(x0, y0, w, h, full_w, full_h)
    return cam, g, lights, gb, tile


def half_ulp_diff(a, b):
    """ULP distance between two float16 arrays (sign-magnitude -> monotone integer)."""
    def key(x):
        u = np.ascontiguousarray(x, dtype=np.float16).view(np.uint16).astype(np.int32)
        return np.where(u & 0x8000, 0x8000 - u, u)
    return np.abs(key(a) - key(b))


def oracle_bloom_from_level1(orc, a1):
    """BloomPass::Execute after its prefilter dispatch, composed from the oracle's stage functions
    (DeferredPipeline.cpp:428-570): a1 = level 1 [h/2, w/2, 4] half; returns A0 [h, w, 4] (what bloom_merge adds)."""
    a = {1: a1}
    for l in (1, 2, 3):                      # B(l+1) = H(A l); A(l+1) = V(B(l+1))
        oh, ow = a[l].shape[0] >> 1, a[l].shape[1] >> 1
        a[l + 1] = orc.blur_v(orc.blur_h(a[l], ow, oh), ow, oh)
    for l in (3, 2, 1):                      # B l = H(A l) + H(A(l+1)); A l = V(B l)
        oh, ow = a[l].shape[:2]
        a[l] = orc.blur_v(orc.bloom_upsample_add(a[l], a[l + 1]), ow, oh)
    h, w = a[1].shape[0] * 2, a[1].shape[1] * 2
    return orc.blur_v(orc.blur_h(a[1], w, h), w, h)


def reference_scene_lights():
    """The 8 `mSceneLight` records of the reference's Asset/Scene/main.json (tests/golden/scene_lights.npz, written by
    tests/golden/make_scene_lights.py in the build container)."""
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "scene_lights.npz"))


def scene_json_text(recs, extra_members=True):
    """A scene file in the reference serializer's shape holding these light records (direct12pbrrenderer_amd/scene.py)."""
    return scene.scene_file_text(recs, extra_members)
