"""Host logic and the C-ABI surface — no GPU needed (no compute call is made)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import common
from direct12pbrrenderer_amd import _lib, scene, structs, synth
from direct12pbrrenderer_amd.pipeline import TileSpec, grid_for_world, tile_for_rank

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "pbr_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"#ifdef PBR_DEBUG_KNOBS\n.*?#endif", "", text, flags=re.S)   # the knobs build's measurement entry points: not product exports (tests/test_runtime_cpu.py)
    return sorted(set(re.findall(r"\b(pbr_[a-z0-9_]+)\s*\(", text)))


def test_library_loads_and_exports_every_declared_symbol():
    names = declared_functions()
    assert len(names) >= 28 and "pbr_deferred_shade" in names and "pbr_allreduce_hist" in names
    lib = _lib.load()
    raw = C.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), f"{n} declared in include/pbr_hip.h but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature"
    assert sorted(_lib.SIGNATURES) == names
    assert lib.pbr_version().decode().startswith("pbr_hip")


def test_layout_helpers_match_python_mirrors():
    lib = _lib.load()
    for size, mips in ((512, 5), (16, 5), (8, 4), (1, 1)):
        assert lib.pbr_cube_texels(size, mips) == structs.cube_texels(size, mips)
        for m in range(mips):
            assert lib.pbr_cube_mip_offset(size, m) == structs.cube_mip_offset(size, m)
    assert structs.cube_texels(512, 5) == 2_095_104
    for w, h in ((3840, 2160), (1920, 1080), (128, 72), (300, 170)):
        assert lib.pbr_bloom_chain_texels(w, h) == structs.bloom_chain_texels(w, h)
        for l in range(5):
            assert lib.pbr_bloom_level_offset(w, h, l) == structs.bloom_level_offset(w, h, l)


def test_struct_sizes_match_reference_layouts():
    assert C.sizeof(structs.Global) == 412 and C.sizeof(structs.ShPack) == 112
    assert structs.LIGHT_DTYPE.itemsize == 44 and structs.CLUSTER_DTYPE.itemsize == 156
    assert C.sizeof(structs.Tile) == 24
    assert structs.NUM_CLUSTERS == 3072


def test_null_context_is_rejected_without_touching_a_gpu():
    lib = _lib.load()
    assert lib.pbr_sync(None) == -1
    assert lib.pbr_brdf_lut(None, 512, None) == -1
    assert lib.pbr_allreduce_hist(None, None) == -1
    assert lib.pbr_last_error(None) == b"null context"


def test_camera_matches_reference_defaults():
    cam = scene.Camera.reference_default(1440, 960)             # App.h:77-78, App.cpp:99-101
    assert np.allclose(cam.translation(), [0, 3, 10])
    W = cam.world_matrix()
    assert np.allclose(W[:3, :3], np.diag([-1, 1, -1]), atol=1e-6)   # yaw PI: looks down -z
    V = cam.local_space_matrix()
    assert np.allclose(V @ W, np.eye(4), atol=1e-5)
    P = cam.projection_matrix()                                  # MathLib.cpp:35-68
    t = np.tan(0.333 * np.pi / 2)
    assert P[0, 0] == pytest.approx(1 / (1.5 * t), rel=1e-5) and P[1, 1] == pytest.approx(1 / t, rel=1e-5)
    assert P[2, 2] == pytest.approx(1000 / 999.9, rel=1e-6) and P[2, 3] == pytest.approx(-100 / 999.9, rel=1e-5) and P[3, 2] == 1
    g = scene.make_global(cam, 1440, 960, delta_time=1 / 60)
    assert g.Ratio == pytest.approx(1.5) and g.Near == pytest.approx(0.1) and g.Far == 1000.0
    assert list(g.Resolution) == [1440.0, 960.0]
    assert np.allclose(np.array(g.InvProjection[:]).reshape(4, 4) @ P, np.eye(4), atol=1e-3)


def test_attenuation_presets_are_a_step_function():
    # Scene.cpp:132-165 (quirk Q18): radius 2 picks the 7.0 preset's coefficients
    assert scene.attenuation_coefficients(2.0) == (2.0, 1.0, 0.7, 1.8)
    assert scene.attenuation_coefficients(0.05)[1:] == (1.0, 45.0, 7500.0)
    assert scene.attenuation_coefficients(7.0)[1:] == (1.0, 0.7, 1.8)
    assert scene.attenuation_coefficients(1.0)[1:] == (1.0, 4.5, 75.0)
    assert scene.attenuation_coefficients(500.0) == (600.0, 1.0, 0.007, 0.0002)   # falls through to the last preset
    l = synth.reference_scene_light()
    assert l["C1"][0] == np.float32(0.7) and l["C2"][0] == np.float32(1.8) and l["Intensity"][0] == 10.0


def test_synthetic_tile_equals_region_of_full_frame():
    full = synth.gbuffer_tile(0, 0, 96, 64, 96, 64, coverage_mask=True)
    sub = synth.gbuffer_tile(32, 16, 40, 20, 96, 64, coverage_mask=True)
    for k in full:
        assert np.array_equal(full[k][16:36, 32:72], sub[k]), k
    assert 0.02 < (full["stencil"] == 0).mean() < 0.3
    assert full["depth"].min() > 0.89 and full["depth"].max() < 1.0
    rough = full["C"] & 255
    assert rough.min() >= 48 and (synth.gbuffer_tile(0, 0, 64, 64, 64, 64, rough_min=0)["C"] & 255).min() < 8


def test_tile_specs_cover_the_frame_once():
    from direct12pbrrenderer_amd.pipeline import halo_plan, parse_layout, tile_of_frame
    expect = {1: (1, 1), 2: (2, 1), 3: (3, 1), 4: (2, 2), 6: (3, 2), 8: (4, 2)}   # most square, cols >= rows
    for world, grid in expect.items():
        for layout in (None, (world, 1)):                     # default grid and one row of tiles
            cols, rows = grid_for_world(world, layout)
            assert (cols, rows) == (grid if layout is None else layout)
            for halo in (False, True):
                cover = np.zeros((rows * 48, cols * 64), dtype=np.int32)
                specs = [tile_for_rank(r, world, 64, 48, apron=16, layout=layout, halo=halo) for r in range(world)]
                for r, s in enumerate(specs):
                    cover[s.y0:s.y0 + s.h, s.x0:s.x0 + s.w] += 1
                    assert (s.full_w, s.full_h) == (cols * 64, rows * 48)
                    assert 0 <= s.ex0 <= s.x0 and s.ex1 <= s.full_w and s.ix == s.x0 - s.ex0
                    assert 0 <= s.ey0 <= s.y0 and s.ey1 <= s.full_h and s.iy == s.y0 - s.ey0
                    assert (s.apron == 0) == (world == 1)
                    assert s.ex0 % 16 == 0 and s.ey0 % 16 == 0      # extended origin stays on the coarsest mip grid
                    assert s.halo == (halo and world > 1)
                    if s.halo:      # shaded rectangle: interior + 4 px towards every neighbour, even origin
                        assert (s.sx0, s.sy0) == (max(s.x0 - 4, 0), max(s.y0 - 4, 0)) and s.sx0 % 2 == 0 and s.sy0 % 2 == 0
                        assert s.sx1 == min(s.x0 + s.w + 4, s.full_w) and s.sy1 == min(s.y0 + s.h + 4, s.full_h)
                    else:
                        assert (s.sx0, s.sy0, s.sw, s.sh) == (s.ex0, s.ey0, s.ew, s.eh)
                assert np.all(cover == 1)
                if halo and world > 1:
                    # halo plan: what r receives from n is what n sends to r, and the strips + the interior tile E/2 exactly
                    plans = [halo_plan(r, world, specs) for r in range(world)]
                    for r, s in enumerate(specs):
                        got = np.zeros((s.full_h // 2, s.full_w // 2), dtype=np.int32)
                        got[s.y0 // 2:(s.y0 + s.h) // 2, s.x0 // 2:(s.x0 + s.w) // 2] += 1
                        for n, snd, rcv in plans[r]:
                            back = [q for q in plans[n] if q[0] == r][0]
                            assert back[1] == rcv and back[2] == snd
                            if rcv:
                                got[rcv[1]:rcv[3], rcv[0]:rcv[2]] += 1
                        e = np.zeros_like(got)
                        e[s.ey0 // 2:s.ey1 // 2, s.ex0 // 2:s.ex1 // 2] = 1
                        assert np.array_equal(got, e)
    assert parse_layout("2x4") == (4, 2)                      # rows x cols, the way BASELINE cfg5 says "tiled 2x4"
    with pytest.raises(ValueError):
        grid_for_world(0)
    with pytest.raises(ValueError):
        grid_for_world(8, (3, 2))
    with pytest.raises(ValueError):
        tile_for_rank(0, 2, 64, 40, apron=16)
    # BASELINE cfg5: 7680x4320 as 2 rows x 4 cols of 1920x2160; an inner tile of the top row carries aprons on three sides
    s5 = tile_of_frame(1, 8, 7680, 4320, layout=parse_layout("2x4"))
    assert (s5.x0, s5.y0, s5.w, s5.h) == (1920, 0, 1920, 2160) and (s5.ew, s5.eh) == (1920 + 512, 2160 + 256)
    assert abs(s5.ew * s5.eh / (1920 * 2160) - 1.417) < 1e-3          # apron mode: 42 % extra shaded pixels ...
    h5 = tile_of_frame(1, 8, 7680, 4320, layout=parse_layout("2x4"), halo=True)
    assert (h5.sw, h5.sh) == (1928, 2164) and h5.sw * h5.sh / (1920 * 2160) < 1.007   # ... halo mode: 0.6 %
    # one row of eight 4K tiles: the busiest rank shades 13.3 % more pixels than its tile in apron mode
    s8 = tile_for_rank(3, 8, 3840, 2160, layout=(8, 1))
    assert (s8.ew, s8.eh) == (3840 + 512, 2160) and abs(s8.ew * s8.eh / (3840 * 2160) - 1.1333) < 1e-3
    s = TileSpec(0, 0, 64, 40, 128, 40, 16)
    assert (s.ew, s.eh) == (80, 40)


# ------------------------------------------------------------------------- CPU light cull
def _host_lib():
    import ctypes as C
    L = C.CDLL(common.host_lib_path())
    L.pbrh_cull_lights.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    return L


def _cull(L, cam_pos_yaw, pos, radius, intensity, w=1280, h=720):
    n = len(pos)
    packed = np.zeros((n, 8), np.float32)
    packed[:, :3] = pos
    packed[:, 3:6] = 1.0
    packed[:, 6] = radius
    packed[:, 7] = intensity
    idx = np.full(max(n, 1), -1, np.int32)
    cp = np.asarray(cam_pos_yaw, np.float32)
    cnt = L.pbrh_cull_lights(w, h, cp.ctypes.data, packed.ctypes.data, n, idx.ctypes.data, len(idx))
    return cnt, idx[:max(cnt, 0)].tolist()


def test_light_cull_octree_order_known_answers():
    # LooseOctree.h: a leaf splits when it would hold a third element; an element that straddles a child boundary
    # stays in the parent; the cull visits a node's own elements first, then children 0..7
    L = _host_lib()
    cam = (0.0, 3.0, 40.0, np.pi)                      # reference camera pulled back: looks down -z at the origin
    pos = np.float32([[10, 10, 10], [-10, 10, 10], [1, 1, 1]])
    one = np.ones(3, np.float32)
    assert _cull(L, cam, pos[:2], 2 * one[:2], one[:2]) == (2, [0, 1])          # root still a leaf: insertion order
    assert _cull(L, cam, pos, 2 * one, one) == (3, [2, 1, 0])                    # root(2), child 6 (x<0), child 7
    # behind the camera: culled; behind but with a culling bound that reaches the near plane: kept
    assert _cull(L, cam, np.float32([[0, 3, 60]]), [2.0], [1.0]) == (0, [])
    assert _cull(L, cam, np.float32([[0, 3, 43]]), [2.0], [1.0]) == (1, [0])     # 3.6 > 3 behind the eye
    # culling radius = radius * 1.81418 * sqrt(intensity) (Scene.cpp:122-130): the same light far off to the side
    assert _cull(L, cam, np.float32([[60, 3, 0]]), [2.0], [1.0])[0] == 0
    assert _cull(L, cam, np.float32([[60, 3, 0]]), [2.0], [100.0])[0] == 1
    # a bound that leaves the +-500 world box is an error (the reference ASSERTs)
    assert _cull(L, cam, np.float32([[499, 0, 0]]), [2.0], [1.0])[0] == -1


def test_light_cull_matches_restatement_on_random_scenes():
    import light_cull_ref
    from direct12pbrrenderer_amd import scene
    L = _host_lib()
    rng = np.random.default_rng(0x5EED0030)
    for n, spread, cam_xyzyaw in [(256, 60.0, (0.0, 3.0, 10.0, np.pi)), (1024, 200.0, (5.0, 2.0, -30.0, 0.7)), (40, 3.0, (0.0, 3.0, 10.0, np.pi))]:
        pos = rng.uniform(-spread, spread, size=(n, 3)).astype(np.float32)
        radius = rng.choice(np.float32([0.5, 2.0, 7.0, 13.0]), size=n).astype(np.float32)
        intensity = rng.uniform(0.5, 10.0, size=n).astype(np.float32)
        cam = scene.Camera(0.333 * 3.14159265359, 1280, 720, 0.1, 1000.0)
        cam.move(cam_xyzyaw[:3])
        cam.rotate(0.0, cam_xyzyaw[3], 0.0)
        want = light_cull_ref.cull_lights(cam, pos, radius, intensity)
        cnt, got = _cull(L, cam_xyzyaw, pos, radius, intensity)
        assert cnt == len(want) and got == want
        assert 0 < cnt <= n and got != sorted(got)          # octree order, not scene order
        assert cnt < n or spread < 10                       # the wide scenes really lose lights to the frustum


def test_reference_scene_lights_reach_the_light_buffer_verbatim():
    """SURVEY 8f row 4, last clause: the 8 lights of the reference's Asset/Scene/main.json (fixture: tests/golden/scene_lights.npz)
    -> scene-file reader -> SceneLight::CaclAttenuationCoefficients (Scene.cpp:132-165) -> Scene::CullLight (Scene.h:229-237)
    -> the PointLight[] ClusteredPass commits (DeferredPipeline.cpp:224-241)."""
    import ctypes as C
    import common
    import light_cull_ref
    L = _host_lib()
    L.pbrh_parse_scene_lights.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_int, C.c_char_p, C.c_size_t]
    L.pbrh_light_buffer.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    recs = common.reference_scene_lights()
    n = len(recs["radius"])
    assert n == 8 and list(recs["name"]) == [f"light_{i}" for i in range(1, 9)]
    assert np.all(recs["rotation"] == 0) and np.all(recs["scale"] == 1)
    # ---- the file: a document in the serializer's shape round-trips through the C++ reader bit for bit
    text = common.scene_json_text(recs).encode()
    got = np.zeros((n, 8), np.float32)
    err = C.create_string_buffer(256)
    assert L.pbrh_parse_scene_lights(text, len(text), got.ctypes.data, n, err, 256) == n, err.value
    want = np.concatenate([recs["translation"], recs["color"], recs["radius"][:, None], recs["intensity"][:, None]], axis=1).astype(np.float32)
    assert np.array_equal(got, want)
    assert np.array_equal(got[0], np.float32([-4.2, 1.0, 3.5, 0.9, 0.1, 0.3, 2.0, 10.0]))       # light_1, SURVEY 8d
    assert L.pbrh_parse_scene_lights(text, len(text), None, 0, err, 256) == n                   # count only
    assert L.pbrh_parse_scene_lights(b'{"mSceneModel": []}', 19, None, 0, err, 256) == 0        # a scene without lights
    for bad, why in ((text[:-2], b"unterminated"), (text.replace(b'"mRadius": 2.0', b'"mRadius": "2"', 1), b"mRadius"),
                     (text.replace(b'"@SceneObject"', b'"SceneObject"', 1), b"@SceneObject"), (b"[1, 2]", b"not an object"),
                     (text.replace(b'"mIntensity": 10.0', b'"mIntensity": 1e', 1), b"exponent"), (text + b"x", b"trailing")):
        assert L.pbrh_parse_scene_lights(bad, len(bad), None, 0, err, 256) == -1 and why in err.value, (why, err.value)
    # ---- attenuation: radius 2 falls into the 7-unit preset (step function, quirk Q18) = SURVEY 8d's (1, 0.7, 1.8)
    buf = np.zeros(16, structs.LIGHT_DTYPE)
    for (w, h, cam) in ((1440, 960, (0.0, 3.0, 10.0, np.pi)), (640, 360, (0.0, 3.0, 10.0, np.pi)), (1280, 720, (3.0, 1.0, -4.0, 0.4))):
        cp = np.float32(cam)
        cnt = L.pbrh_light_buffer(w, h, cp.ctypes.data, got.ctypes.data, n, buf.ctypes.data, len(buf))
        c = scene.Camera(0.333 * 3.14159265359, w, h, 0.1, 1000.0)
        c.move(cam[:3])
        c.rotate(0.0, cam[3], 0.0)
        order = light_cull_ref.cull_lights(c, recs["translation"], recs["radius"], recs["intensity"])
        assert cnt == len(order) and cnt >= 1
        out = buf[:cnt]
        assert np.array_equal(out["Position"], recs["translation"][order]) and np.array_equal(out["Color"], recs["color"][order])
        assert np.all(out["Intensity"] == 10.0) and np.all(out["Radius"] == 2.0)
        assert np.all(out["C0"] == np.float32(1.0)) and np.all(out["C1"] == np.float32(0.7)) and np.all(out["C2"] == np.float32(1.8))
        # the same records through the Python mirror the bench / parity tests build their light buffers with
        mirror = scene.make_lights(recs["translation"][order], recs["color"][order], 2.0, 10.0)
        assert mirror.tobytes() == out.tobytes()
    # the reference camera (App.cpp:99-101) sees all eight: culling radius 2 * 1.81418 * sqrt(10) = 11.5 around a 13-unit-wide ring
    cp = np.float32([0.0, 3.0, 10.0, np.pi])
    assert L.pbrh_light_buffer(1440, 960, cp.ctypes.data, got.ctypes.data, n, buf.ctypes.data, len(buf)) == 8
    assert L.pbrh_light_buffer(1440, 960, cp.ctypes.data, got.ctypes.data, n, buf.ctypes.data, 4) == -1    # a buffer too small is an error


def test_rotated_and_scaled_scene_lights_cull_like_the_reference_bound():
    """a light whose object carries mRotation / mScale.  The reference's world bound is matrix * local cube with only the two
    CORNERS transformed (MathLib.cpp:5-10) — AddSceneLights restates exactly that: bounds and frustum-cull membership + order against
    the numpy restatement (tests/light_cull_ref.py), on random rotations / scales and on the unrotated fixture lights."""
    import ctypes as C
    import common
    import light_cull_ref
    L = _host_lib()
    L.pbrh_scene_light_bounds.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p, C.c_char_p, C.c_size_t, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_char_p, C.c_size_t]
    rng = np.random.default_rng(0x5EED0051)
    n = 96
    recs = {"name": [f"l{i}" for i in range(n)], "translation": rng.uniform(-40, 40, (n, 3)).astype(np.float32),
            "rotation": np.where(rng.random((n, 1)) < 0.7, rng.uniform(-180, 180, (n, 3)), 0.0).astype(np.float32),
            "scale": np.where(rng.random((n, 1)) < 0.7, rng.uniform(0.25, 3.0, (n, 3)), 1.0).astype(np.float32),
            "color": rng.uniform(0, 1, (n, 3)).astype(np.float32), "radius": rng.choice(np.float32([0.5, 2.0, 7.0]), n).astype(np.float32),
            "intensity": rng.uniform(0.5, 10.0, n).astype(np.float32)}
    assert ((recs["rotation"] != 0).any(axis=1) | (recs["scale"] != 1).any(axis=1)).sum() > 60
    text = common.scene_json_text(recs).encode()
    err = C.create_string_buffer(256)
    for (w, h, cam) in ((1440, 960, (0.0, 3.0, 10.0, np.pi)), (1280, 720, (3.0, 1.0, -4.0, 0.4))):
        cp = np.float32(cam)
        bounds = np.zeros((n, 6), np.float32)
        vis = np.zeros(n, np.int32)
        nvis = C.c_int(0)
        assert L.pbrh_scene_light_bounds(w, h, cp.ctypes.data, text, len(text), bounds.ctypes.data, n, vis.ctypes.data, C.byref(nvis), err, 256) == n, err.value
        want = [light_cull_ref.rotated_scaled_bound(recs["translation"][i], recs["rotation"][i], recs["scale"][i], recs["radius"][i], recs["intensity"][i])
                if (recs["rotation"][i] != 0).any() or (recs["scale"][i] != 1).any()
                else light_cull_ref.light_bound(recs["translation"][i], recs["radius"][i], recs["intensity"][i]) for i in range(n)]
        wb = np.array([np.concatenate(b) for b in want], dtype=np.float32)
        assert np.abs(bounds - wb).max() <= 2e-5 * np.abs(wb).max(), np.abs(bounds - wb).max()      # libm cos / sin vs numpy's: a few ulps
        c = scene.Camera(0.333 * 3.14159265359, w, h, 0.1, 1000.0)
        c.move(cam[:3])
        c.rotate(0.0, cam[3], 0.0)
        # membership and visiting order from the C++ side's OWN bounds (so that an ulp of cos / sin cannot flip a boundary case)
        order = light_cull_ref.cull_bounds(c, [(b[:3], b[3:]) for b in bounds])
        assert order is not None and nvis.value == len(order) and list(vis[:nvis.value]) == order and 0 < len(order) < n
    # a rotated light's bound is NOT the hull of the rotated cube: 45 degrees about one axis collapses two extents to (almost) nothing
    one = {k: (v[:1] if not isinstance(v, list) else v[:1]) for k, v in recs.items()}
    one["rotation"] = np.float32([[0.0, 0.0, 45.0]]); one["scale"] = np.float32([[1.0, 1.0, 1.0]])
    t1 = common.scene_json_text(one).encode()
    b1 = np.zeros((1, 6), np.float32)
    assert L.pbrh_scene_light_bounds(640, 360, np.float32([0, 3, 10, np.pi]).ctypes.data, t1, len(t1), b1.ctypes.data, 1, None, None, err, 256) == 1
    ext = b1[0, 3:] - b1[0, :3]
    assert ext.min() < 1e-4 * ext.max()


def test_scene_file_reader_survives_mutated_input():
    """Property test (hypothesis): whatever bytes arrive, the scene-file reader of the C++ host answers with a count or with -1 and a
    reason — never a crash (tools/asan_cpu.sh runs this against the ASan + UBSan build); documents in the serializer's shape
    round-trip every float exactly."""
    import ctypes as C
    import json
    from hypothesis import given, settings, strategies as st
    L = _host_lib()
    L.pbrh_parse_scene_lights.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_int, C.c_char_p, C.c_size_t]
    err = C.create_string_buffer(256)
    f32 = st.floats(min_value=-400.0, max_value=400.0, width=32)
    pos32 = st.floats(min_value=0.125, max_value=64.0, width=32)
    vec = st.fixed_dictionaries({"x": f32, "y": f32, "z": f32})
    light = st.fixed_dictionaries({"@SceneObject": st.fixed_dictionaries({"mName": st.text(max_size=12), "mTranslation": vec,
                                                                          "mRotation": st.just({"x": 0.0, "y": 0.0, "z": 0.0}),
                                                                          "mScale": st.just({"x": 1.0, "y": 1.0, "z": 1.0})}),
                                   "mColor": vec, "mRadius": pos32, "mIntensity": pos32})
    base = common.scene_json_text(common.reference_scene_lights()).encode()

    @settings(max_examples=150, deadline=None)
    @given(st.lists(light, max_size=6), st.booleans())
    def roundtrip(lights, pretty):
        text = json.dumps({"mSceneLight": lights, "mSkyBoxPath": "x"}, indent=1 if pretty else None, ensure_ascii=pretty).encode()
        got = np.zeros((max(len(lights), 1), 8), np.float32)
        assert L.pbrh_parse_scene_lights(text, len(text), got.ctypes.data, len(lights), err, 256) == len(lights), err.value
        for i, l in enumerate(lights):
            t, c = l["@SceneObject"]["mTranslation"], l["mColor"]
            want = np.float32([t["x"], t["y"], t["z"], c["x"], c["y"], c["z"], l["mRadius"], l["mIntensity"]])
            assert np.array_equal(got[i], want)

    @settings(max_examples=300, deadline=None)
    @given(st.integers(0, len(base) - 1), st.integers(0, 255), st.integers(0, len(base)))
    def mutated(pos, byte, cut):
        blob = bytearray(base)
        blob[pos] = byte
        for data in (bytes(blob), bytes(blob[:cut]), base[:cut] + bytes([byte]) + base[cut:]):
            out = np.zeros((16, 8), np.float32)
            n = L.pbrh_parse_scene_lights(data, len(data), out.ctypes.data, 16, err, 256)
            assert n == -1 or 0 <= n <= 16, n
            assert n != -1 or err.value                    # a refusal names its reason

    roundtrip()
    mutated()


def test_hdr_parser_survives_mutated_input():
    """The same property for the Radiance .hdr reader: mutated and truncated files (flat and run-length coded) are parsed or refused with
    a reason, the declared size bounds the output, nothing crashes."""
    import ctypes as C
    import hdr_writer
    from hypothesis import given, settings, strategies as st
    L = _host_lib()
    L.pbrh_parse_hdr.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t]
    rng = np.random.default_rng(5)
    img = rng.uniform(0.0, 4.0, size=(9, 16, 3)).astype(np.float32)
    rgbe = hdr_writer.float_to_rgbe(img)
    bases = [hdr_writer.encode_hdr(rgbe, rle=False), hdr_writer.encode_hdr(rgbe, rle=True)]
    err = C.create_string_buffer(256)

    @settings(max_examples=300, deadline=None)
    @given(st.integers(0, 1), st.integers(0, 4000), st.integers(0, 255), st.integers(0, 4000))
    def mutated(which, pos, byte, cut):
        base = bases[which]
        blob = bytearray(base)
        blob[pos % len(blob)] = byte
        for data in (bytes(blob), bytes(blob[:cut % (len(blob) + 1)])):
            w, h = C.c_uint32(0), C.c_uint32(0)
            st_ = L.pbrh_parse_hdr(data, len(data), C.byref(w), C.byref(h), None, 0, err, 256)
            assert st_ in (0, -1)
            if st_ == 0 and 0 < w.value * h.value <= 1 << 16:
                out = np.zeros((h.value, w.value, 4), np.uint8)
                assert L.pbrh_parse_hdr(data, len(data), C.byref(w), C.byref(h), out.ctypes.data, out.nbytes, err, 256) in (0, -1)
            elif st_ == -1:
                assert err.value

    mutated()


# ------------------------------------------------------------------------- .hdr ingestion
def _parse_hdr(L, blob, cap=None):
    import ctypes as C
    L.pbrh_parse_hdr.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t]
    w, h = C.c_uint32(0), C.c_uint32(0)
    err = C.create_string_buffer(256)
    if L.pbrh_parse_hdr(blob, len(blob), C.byref(w), C.byref(h), None, 0, err, 256) != 0:
        return None, err.value.decode()
    out = np.zeros((h.value, w.value, 4), np.uint8)
    assert L.pbrh_parse_hdr(blob, len(blob), C.byref(w), C.byref(h), out.ctypes.data, out.nbytes if cap is None else cap, err, 256) == (0 if cap is None else -1)
    return out, err.value.decode()


def test_hdr_parse_flat_rle_and_malformed():
    import hdr_writer
    L = _host_lib()
    rng = np.random.default_rng(7)
    img = rng.uniform(0, 4, size=(12, 40, 3)).astype(np.float32)
    img[3, 5:30] = (0.25, 0.5, 1.0)            # long runs in every component
    img[7] = 0.0                                # a black row: exponent byte 0
    img[9, :, 1] = 1000.0
    rgbe = hdr_writer.float_to_rgbe(img)
    assert tuple(rgbe[3, 5]) == (32, 64, 128, 129) and tuple(rgbe[7, 0]) == (0, 0, 0, 0)
    for rle in (False, True):
        for crlf in (False, True):
            blob = hdr_writer.encode_hdr(rgbe, rle=rle, extra_header=("# made by tests", "EXPOSURE=1.0"), crlf=crlf)
            got, err = _parse_hdr(L, blob)
            assert got is not None, err
            assert np.array_equal(got, rgbe)
    assert len(hdr_writer.encode_hdr(rgbe, rle=True)) < len(hdr_writer.encode_hdr(rgbe, rle=False))
    # a 4-wide image is always flat (RLE needs 8 <= w < 32768)
    small = hdr_writer.float_to_rgbe(rng.uniform(0, 1, size=(4, 4, 3)))
    got, _ = _parse_hdr(L, hdr_writer.encode_hdr(small, rle=True))
    assert np.array_equal(got, small)
    good = hdr_writer.encode_hdr(rgbe, rle=True)
    for blob, why in [(b"#?RADIANCX" + good[10:], "signature"), (good[:-7], "truncated"), (good.replace(b"32-bit_rle_rgbe", b"32-bit_rle_xyze"), "unsupported"),
                      (good.replace(b"-Y 12 +X 40", b"+Y 12 +X 40"), "resolution"), (good.replace(b"-Y 12 +X 40", b"-Y 12 +X 48"), "width"),
                      (good.replace(b"FORMAT=32-bit_rle_rgbe\n", b"EXPOSURE=2.0\nFORMAT=32-bit_rle_rgbe\n"), "EXPOSURE"), (b"", "small")]:
        got, err = _parse_hdr(L, blob)
        assert got is None and why in err, (why, err)
    _, err = _parse_hdr(L, good, cap=16)
    assert "too small" in err


def test_rgbe_decode_oracle_known_answers(orc):
    # published rule: e == 0 -> 0, else mantissa * 2^(e - 136); alpha 1
    t = np.uint8([[128, 64, 32, 129], [255, 0, 1, 128], [9, 9, 9, 0], [1, 2, 3, 136], [255, 255, 255, 255], [1, 0, 0, 1]])
    d = orc.rgbe_decode(t).astype(np.float64)
    assert d[0].tolist() == [1.0, 0.5, 0.25, 1.0]
    assert d[1].tolist() == [255 / 256, 0.0, 1 / 256, 1.0]
    assert d[2].tolist() == [0.0, 0.0, 0.0, 1.0]
    assert d[3].tolist() == [1.0, 2.0, 3.0, 1.0]
    assert d[4, 0] == 255.0 * 2.0 ** 119 and d[5, 0] == 2.0 ** -135           # top of the range, a subnormal
    import hdr_writer
    x = np.exp(np.random.default_rng(3).uniform(-20, 20, size=(1000, 3))).astype(np.float32)
    back = orc.rgbe_decode(hdr_writer.float_to_rgbe(x))[:, :3]
    assert np.all(back <= x) and np.all(x - back <= x.max(axis=1, keepdims=True) / 128.0)   # truncating 8-bit mantissa


def test_ring_core_split_of_the_overlapped_halo_frame():
    """DeferredFrame._ring_core_split (no GPU needed): ring U core tile the shaded rectangle resp. the interior's level-1
    texels exactly once, every send rectangle of the halo plan lies in the ring, and the ring's level-1 texels read
    ring pixels only — so the exchange can start before the core is shaded."""
    from direct12pbrrenderer_amd import pipeline
    from direct12pbrrenderer_amd.pipeline import halo_plan, tile_of_frame

    class Stub:
        pass
    cases = [[tile_of_frame(r, 8, 7680, 4320, layout=(4, 2), halo=True) for r in range(8)],
             [tile_for_rank(r, 8, 2720, 3056, layout=(4, 2), halo=True) for r in range(8)],
             [tile_for_rank(r, 2, 2720, 3056, halo=True) for r in range(2)],
             [tile_for_rank(r, 4, 1024, 640, layout=(2, 2), halo=True) for r in range(4)]]
    for specs in cases:
        for rank, spec in enumerate(specs):
            f = Stub()
            f.spec = spec
            ring, core, l1_ring, l1_core = pipeline.DeferredFrame._ring_core_split(f)
            cov = np.zeros((spec.sh, spec.sw), dtype=np.int32)
            for x, y, w, h in ring + [core]:
                cov[y:y + h, x:x + w] += 1
            assert (cov == 1).all()
            c1 = np.zeros((spec.sh // 2, spec.sw // 2), dtype=np.int32)
            for x, y, w, h in l1_ring + [l1_core]:
                c1[y:y + h, x:x + w] += 1
            ox, oy = spec.six // 2, spec.siy // 2
            want = np.zeros_like(c1)
            want[oy:oy + spec.h // 2, ox:ox + spec.w // 2] = 1
            assert np.array_equal(c1, want)
            shaded = np.zeros((spec.sh, spec.sw), dtype=bool)
            for x, y, w, h in ring:
                shaded[y:y + h, x:x + w] = True
            in_ring = np.zeros_like(c1, dtype=bool)
            for x, y, w, h in l1_ring:      # prefilter of half-res texel x reads full-res 2x-3 .. 2x+2 (clamped at the image edge)
                in_ring[y:y + h, x:x + w] = True
                assert shaded[max(2 * y - 3, 0):min(2 * (y + h - 1) + 2, spec.sh - 1) + 1, max(2 * x - 3, 0):min(2 * (x + w - 1) + 2, spec.sw - 1) + 1].all()
            for n, snd, rcv in halo_plan(rank, len(specs), specs):
                if snd:                          # global half-res -> S-image half-res
                    sx, sy = snd[0] - spec.sx0 // 2, snd[1] - spec.sy0 // 2
                    assert in_ring[sy:sy + snd[3] - snd[1], sx:sx + snd[2] - snd[0]].all(), (rank, n)
    small = Stub()
    small.spec = tile_for_rank(0, 4, 384, 288, layout=(2, 2), halo=True)
    assert pipeline.DeferredFrame._ring_core_split(small) is None      # too small to split: plain sequence


def test_quick_slice_estimate_agrees_with_the_exact_index_outside_its_band():
    """csrc/shade.hip takes the cluster z-slice from t = slice_k * v_log_f32(zc * inv_near) wherever t is further than 1e-4
    from an integer, and from the shader's expression (int)(Z * logf(zc / Near) / log(Far / Near)) otherwise.  Numeric model of
    that rule in fp32 (hardware log2 modelled as the correctly rounded value +/- 1 ulp): no depth outside the band — random
    ones and ones within a few ppm of every slice boundary — may truncate differently; the two values stay within 1e-5."""
    import numpy as np
    f32 = np.float32
    rng = np.random.default_rng(0x511CE)
    for near, far in [(0.1, 1000.0), (0.01, 10000.0), (1.0, 100.0), (0.5, 1.0), (0.001, 100000.0)]:
        n = 400_000
        zb = near * (far / near) ** (rng.integers(0, 25, n) / 24.0)
        z = np.concatenate([zb * (1 + rng.normal(0, 3e-6, n)), near * (far / near) ** rng.random(n)]).astype(f32)
        zc = np.minimum(np.maximum(z, f32(near)), f32(far))
        exact = (f32(24.0) * np.log((zc / f32(near)).astype(f32)).astype(f32) / f32(np.log(f32(far) / f32(near)))).astype(f32)
        inv_near, slice_k = f32(1.0 / float(f32(near))), f32(24.0 / np.log2(float(f32(far)) / float(f32(near))))
        l2 = np.log2((zc * inv_near).astype(f32).astype(np.float64)).astype(f32)
        for l2p in (l2, np.nextafter(l2, f32(np.inf)), np.nextafter(l2, f32(-np.inf))):
            t = (slice_k * l2p).astype(f32)
            fr = t - np.floor(t)
            outside = (fr > 1e-4) & (fr < 1 - 1e-4)
            assert not (outside & (np.trunc(t) != np.trunc(exact))).any()
            assert np.abs(t.astype(np.float64) - exact.astype(np.float64)).max() < 1e-5
