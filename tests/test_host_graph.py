"""The C++ host pass graph (direct12pbrrenderer_amd/host): IRenderPass / ShadingState / FrameGraph
look-alikes of Engine/ that dispatch the HIP kernels through the C ABI."""
import ctypes as C
import os

import numpy as np
import pytest

import common
from direct12pbrrenderer_amd import synth
from direct12pbrrenderer_amd.structs import Global, bloom_chain_texels, cube_texels

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST_LIB = common.host_lib_path()

# the order FGExecutionParser::Parse (FrameGraph.cpp:191-250) derives
REFERENCE_ORDER = "PreFilterEnvMap>PrecomputeBRDF>Clustered>GBuffer>Skybox>DeferredShading>Bloom>AutoExposure>ToneMapping>Present"


@pytest.fixture(scope="module")
def host():
    assert os.path.exists(HOST_LIB), "build with make -C direct12pbrrenderer_amd/host"
    import torch  # noqa: F401  (before the HIP runtime the library links: see direct12pbrrenderer_amd/_lib.py)
    L = C.CDLL(HOST_LIB)
    L.pbrh_create.restype = C.c_void_p
    L.pbrh_create.argtypes = [C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_char_p, C.c_size_t]
    L.pbrh_destroy.argtypes = [C.c_void_p]
    L.pbrh_last_error.restype = C.c_char_p
    L.pbrh_last_error.argtypes = [C.c_void_p]
    L.pbrh_set_skybox.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
    L.pbrh_set_fused.argtypes = [C.c_void_p, C.c_int]
    L.pbrh_render_n.argtypes = [C.c_void_p, C.c_int, C.c_float, C.POINTER(C.c_double)]
    L.pbrh_load_skybox.argtypes = [C.c_void_p, C.c_char_p]
    L.pbrh_cull_lights.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    L.pbrh_set_materials.argtypes = [C.c_void_p] * 6
    L.pbrh_set_lights.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.pbrh_set_gbuffer.argtypes = [C.c_void_p] + [C.c_void_p] * 5
    L.pbrh_set_initial_luminance.argtypes = [C.c_void_p, C.c_float]
    L.pbrh_render.argtypes = [C.c_void_p, C.c_float]
    L.pbrh_execution_order.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
    L.pbrh_dispatch_count.argtypes = [C.c_void_p]
    L.pbrh_read.restype = C.c_long
    L.pbrh_read.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t]
    L.pbrh_get_global.argtypes = [C.c_void_p, C.c_void_p]
    L.pbrh_probe_binding.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int]
    L.pbrh_dry_run_execution_order.argtypes = [C.c_uint32, C.c_uint32, C.c_char_p, C.c_size_t]
    L.pbrh_event_log.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
    L.pbrh_set_tile.argtypes = [C.c_void_p] + [C.c_uint32] * 8
    L.pbrh_comm_init.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.pbrh_set_external_histogram.argtypes = [C.c_void_p, C.c_void_p]
    L.pbrh_capture_histogram.argtypes = [C.c_void_p, C.c_int]
    L.pbrh_captured_histogram.argtypes = [C.c_void_p, C.c_void_p]
    L.pbrh_create_tile.restype = C.c_void_p
    L.pbrh_create_tile.argtypes = [C.c_int] + [C.c_uint32] * 5 + [C.c_int, C.c_uint32, C.c_uint32, C.c_char_p, C.c_size_t]
    L.pbrh_set_halo_loopback.argtypes = [C.c_void_p, C.c_int]
    L.pbrh_halo_copy_from.argtypes = [C.c_void_p, C.c_void_p]
    L.pbrh_set_frames_in_flight.argtypes = [C.c_void_p, C.c_int]
    L.pbrh_set_tail_overlap.argtypes = [C.c_void_p, C.c_int]
    L.pbrh_load_scene_lights.argtypes = [C.c_void_p, C.c_char_p]
    L.pbrh_light_buffer.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    return L


def test_execution_order_matches_reference_graph(host):
    buf = C.create_string_buffer(512)
    for w, h in ((1440, 960), (3840, 2160), (7680, 4320)):
        assert host.pbrh_dry_run_execution_order(w, h, buf, 512) == 0, buf.value
        assert buf.value.decode() == REFERENCE_ORDER
    # BloomPass asserts BloomStep < max mip levels (DeferredPipeline.cpp:343): 8x8 cannot hold 5 mips
    assert host.pbrh_dry_run_execution_order(8, 8, buf, 512) == -1 and b"too small" in buf.value
    # TextureFormatKey stores 16-bit extents (quirk Q24)
    assert host.pbrh_dry_run_execution_order(70000, 64, buf, 512) == -1 and b"65535" in buf.value


def test_shading_state_binding_contract(host):
    # known name -> true; unknown name -> false + log (IPipeline.cpp:188-199); unknown shader file -> throws
    T, RWT, SB, RWSB = 0, 1, 2, 3
    assert host.pbrh_probe_binding(b"deferred_shading.hlsl", 0, b"GBufferA", T) == 1
    assert host.pbrh_probe_binding(b"deferred_shading.hlsl", 0, b"Clusters", SB) == 1
    assert host.pbrh_probe_binding(b"deferred_shading.hlsl", 0, b"GBufferZ", T) == 0
    assert host.pbrh_probe_binding(b"deferred_shading.hlsl", 0, b"GBufferA", RWT) == 0      # SRV, not UAV
    assert host.pbrh_probe_binding(b"bloom_upsample_add.hlsl", 1, b"LowerLevel", T) == 1
    assert host.pbrh_probe_binding(b"hdr_average_histogram.hlsl", 1, b"AverageLuminance", RWSB) == 1
    assert host.pbrh_probe_binding(b"clustered_culling.hlsl", 1, b"PointLights", RWSB) == 1
    assert host.pbrh_probe_binding(b"no_such_shader.hlsl", 1, b"x", T) == -1
    assert host.pbrh_probe_binding(b"blur_vertical.hlsl", 0, b"InputTexture", T) == -1        # compute shader bound as graphics


def to_half(t):
    import torch
    return t.cpu().view(torch.int16).numpy().view(np.float16)


@pytest.mark.gpu
def test_host_graph_frame_matches_c_abi_pipeline_and_oracle(host, ctx, orc):
    """Two frames through RenderScheduler/FrameGraph vs the same passes issued from Python
    (bit-identical: both are the same C-ABI calls) and vs the oracle (parity tolerances)."""
    import torch
    from direct12pbrrenderer_amd import scene
    from direct12pbrrenderer_amd.pipeline import DeferredFrame, TileSpec
    W, H, ENV, LUT = 320, 192, 32, 64
    err = C.create_string_buffer(256)
    r = host.pbrh_create(0, W, H, ENV, LUT, err, 256)
    assert r, err.value
    try:
        sky_np = synth.env_cube(ENV)
        n0 = 4 * 6 * ENV * ENV
        assert host.pbrh_set_skybox(r, sky_np[:n0].ctypes.data, ENV) == 0, host.pbrh_last_error(r)
        cam = scene.Camera.reference_default(W, H)
        lights = synth.lights_in_view_box(256, cam)
        packed = np.concatenate([lights["Position"], lights["Color"], np.full((256, 1), 2.0, np.float32), lights["Intensity"][:, None]], axis=1).astype(np.float32)
        packed = np.ascontiguousarray(packed)
        assert host.pbrh_set_lights(r, packed.ctypes.data, 256) == 0
        # ClusteredPass fills the light buffer in the order Scene::CullLight walks its octree:
        # the Python / oracle side of this test gets the lights in that order
        order = np.zeros(256, np.int32)
        cam4 = np.float32([0.0, 3.0, 10.0, 3.14159265359])
        assert host.pbrh_cull_lights(W, H, cam4.ctypes.data, packed.ctypes.data, 256, order.ctypes.data, 256) == 256
        assert sorted(order.tolist()) == list(range(256))   # (these 23-unit bounds all straddle y = 0: they stay in the root, in scene order)
        lights = lights[order]
        gb = synth.gbuffer_tile(0, 0, W, H, W, H, coverage_mask=True)
        assert host.pbrh_set_gbuffer(r, *[np.ascontiguousarray(gb[k]).ctypes.data for k in ("A", "B", "C", "depth", "stencil")]) == 0
        assert host.pbrh_set_initial_luminance(r, 0.18) == 0
        assert host.pbrh_render(r, 1.0 / 60.0) == 0, host.pbrh_last_error(r)
        assert host.pbrh_dispatch_count(r) == 5 + 1 + 2 + 1 + 1 + 16 + 2 + 1      # first frame incl. one-shot IBL; sky draw + shade
        buf = C.create_string_buffer(512)
        assert host.pbrh_execution_order(r, buf, 512) == 0 and buf.value.decode() == REFERENCE_ORDER
        # named ranges = the reference's PIXScope strings, in its nesting order (DeferredPipeline.cpp:65-562); roctx here
        ev = C.create_string_buffer(2048)
        assert host.pbrh_event_log(r, ev, 2048) == 0
        down = ["Blur Horizontal", "Blur Vertical"] * 3
        up = ["Upsample Horizontal Add", "Blur Vertical"] * 3
        assert ev.value.decode().split(">") == (
            ["Precompute PrefilterEnvMap Pass", "Precompute BRDF Pass", "Clustered Pass", "Gbuffer Pass", "Skybox Pass", "Deferred Shading",
             "Bloom Pass", "Bloom Prefilter", "Bloom Downsample"] + down + ["Bloom Upsample"] + up +
            ["Upsample Merge", "Blur Horizontal", "Blur Vertical", "Merge", "Auto Exposure Pass", "Luminance Histogram Pass",
             "Average Luminance Pass", "Tone Mapping Pass"])

        def read(name, shape, dtype):
            a = np.zeros(shape, dtype=dtype)
            n = host.pbrh_read(r, name.encode(), a.ctypes.data, a.nbytes)
            assert n == a.nbytes, (name, n, host.pbrh_last_error(r))
            return a
        hdr1 = read("DeferredShadingRT", (H, W, 4), np.float16)
        ldr1 = read("ToneMappedTexture", (H, W), np.uint32)
        avg1 = read("AverageLuminance", (1,), np.float32)[0]
        g_host = Global()
        assert host.pbrh_get_global(r, C.byref(g_host)) == 0

        # --- the same frame issued from Python through the same C ABI
        sky_mips = int(np.log2(ENV)) + 1
        sky = ctx.upload(sky_np)
        ctx.cube_gen_mips(sky, ENV, sky_mips)
        lut = ctx.brdf_lut(LUT)
        env = ctx.prefilter_env_dispatches(sky, ENV, sky_mips, ENV, 5)    # one dispatch per mip, like PreFilterEnvMapPass
        fast = to_half(ctx.prefilter_env(sky, ENV, sky_mips, ENV, 5))     # all mips, table-driven: <= 1 fp16 ULP or 1e-3 apart
        seq = to_half(env)
        assert ((common.half_ulp_diff(fast, seq) <= 1) | (np.abs(fast.astype(np.float32) - seq.astype(np.float32)) <= 1e-3 * np.abs(seq.astype(np.float32)))).all()
        sh = ctx.sh9_project(sky, ENV, sky_mips).cpu().numpy()
        g = scene.make_global(cam, W, H, sh_pack=sh, delta_time=1.0 / 60.0, time=1.0 / 60.0)
        for f in ("InvView", "View", "Projection", "CameraPos"):      # C++ Camera == Python Camera
            assert np.allclose(np.array(getattr(g_host, f)[:]), np.array(getattr(g, f)[:]), rtol=1e-6, atol=1e-7), f
        assert bytes(g_host.SkyBoxSH) == bytes(g.SkyBoxSH)
        fr = DeferredFrame(ctx, TileSpec(0, 0, W, H, W, H, 0), g_host, lights, lut, LUT, env, ENV, 5, sky=(sky, ENV, sky_mips))
        fr.upload_gbuffer(gb)
        fr.set_prev_luminance(0.18)
        fr.render()
        hdr_py = fr.hdr.cpu().view(torch.int16).numpy().view(np.float16)
        on = gb["stencil"] > 0
        assert (~on).sum() > 1000                                   # the sky pass had pixels to resolve
        assert np.array_equal(hdr1.view(np.uint16), hdr_py.view(np.uint16))
        assert np.array_equal(ldr1, fr.ldr_numpy())
        assert avg1 == fr.avg.cpu().numpy()[0]

        # --- second frame: one-shot passes latched (mReady), 22 dispatches, exposure keeps adapting
        assert host.pbrh_render(r, 1.0 / 60.0) == 0
        assert host.pbrh_dispatch_count(r) == 23
        avg2 = read("AverageLuminance", (1,), np.float32)[0]
        fr.render()
        assert avg2 == fr.avg.cpu().numpy()[0] and avg2 != avg1

        # --- oracle parity of the first frame's HDR (shade + bloom), IBL inputs taken from the GPU
        lut_np = lut.cpu().view(torch.int16).numpy().view(np.float16)
        env_np = env.cpu().view(torch.int16).numpy().view(np.float16)
        cl = orc.cluster_build(g_host)
        orc.cluster_cull(g_host, lights, cl)
        from direct12pbrrenderer_amd.structs import Tile
        want, _ = orc.deferred_shade(g_host, Tile(0, 0, W, H, W, H), gb, lut_np, env_np, ENV, 5, cl, lights)
        sky_full = sky.cpu().numpy()
        orc.skybox(g_host, Tile(0, 0, W, H, W, H), sky_full, ENV, sky_mips, gb["stencil"], want)
        orc.bloom(want)
        scale = np.abs(want.astype(np.float32)[:, :, :3]).max()
        d = common.half_ulp_diff(hdr1[:, :, :3], want[:, :, :3])
        assert (d > 2).mean() <= 1e-3, (d.max(), (d > 2).mean())   # rare slice/threshold flips move a single pixel
        assert np.abs(hdr1.astype(np.float32) - want.astype(np.float32))[:, :, :3].max() <= 5e-3 * scale
        assert bloom_chain_texels(W, H) == 320 * 192 + 160 * 96 + 80 * 48 + 40 * 24 + 20 * 12

        # --- third frame from the rasterizer's material attributes: GBufferPass encodes them on the GPU
        m0, m1, m2 = synth.material_tile(0, 0, W, H, W, H)
        assert host.pbrh_set_materials(r, m0.ctypes.data, m1.ctypes.data, m2.ctypes.data,
                                       np.ascontiguousarray(gb["depth"]).ctypes.data, np.ascontiguousarray(gb["stencil"]).ctypes.data) == 0
        assert host.pbrh_render(r, 1.0 / 60.0) == 0, host.pbrh_last_error(r)
        assert host.pbrh_dispatch_count(r) == 24
        wantA, wantB, wantC = orc.gbuffer_encode(m0, m1, m2)
        assert np.array_equal(read("GBufferB", (H, W), np.uint32), wantB)
        assert np.array_equal(read("GBufferC", (H, W), np.uint32), wantC)
        gotA = read("GBufferA", (H, W), np.uint32)
        assert np.abs(gotA.view(np.uint8).astype(np.int16) - wantA.view(np.uint8).astype(np.int16)).max() <= 1
        hdr3 = read("DeferredShadingRT", (H, W, 4), np.float16)
        assert np.isfinite(hdr3.astype(np.float32)[on]).mean() > 0.99 and not np.array_equal(hdr3, hdr1)
    finally:
        host.pbrh_destroy(r)


@pytest.mark.gpu
@pytest.mark.parametrize("W,H,ENV,LUT", [(640, 360, 32, 64), (1440, 960, 512, 512)])
def test_host_graph_renders_the_reference_scene_lights(host, orc, tmp_path, W, H, ENV, LUT):
    """the 8 lights of the reference's Asset/Scene/main.json (tests/golden/scene_lights.npz), read from a scene file
    by the C++ host (pbrh_load_scene_lights -> SceneLight presets -> Scene::CullLight -> ClusteredPass), light a frame; the HDR
    target after shade + bloom, the adapted luminance and the LDR image are compared with the oracle fed the PointLight[]
    records the CPU test pins (tests/test_host.py::test_reference_scene_lights_reach_the_light_buffer_verbatim).
    1440x960 / env 512^2 / LUT 512^2 = THE REFERENCE'S OWN OPERATING POINT (Engine/Include/App.h:77-78, DeferredPipeline.h:80,85): its
    default target, its table sizes, its scene's lights, the reference camera, the fence per frame (D3D12Device.cpp:993-1003), every
    pass dispatch by dispatch through libpbr_host.so — the whole 1.38-Mpixel frame against the oracle."""
    from direct12pbrrenderer_amd import scene
    from direct12pbrrenderer_amd.structs import LIGHT_DTYPE, Tile
    recs = common.reference_scene_lights()
    path = tmp_path / "main.json"
    path.write_text(common.scene_json_text(recs))
    err = C.create_string_buffer(256)
    r = host.pbrh_create(0, W, H, ENV, LUT, err, 256)
    assert r, err.value
    try:
        sky_np = synth.env_cube(ENV)
        assert host.pbrh_set_skybox(r, sky_np[:4 * 6 * ENV * ENV].ctypes.data, ENV) == 0, host.pbrh_last_error(r)
        assert host.pbrh_load_scene_lights(r, str(path).encode()) == 0, host.pbrh_last_error(r)
        assert host.pbrh_load_scene_lights(r, str(tmp_path / "missing.json").encode()) == -1 and b"cannot open" in host.pbrh_last_error(r)
        gb = synth.gbuffer_tile(0, 0, W, H, W, H, coverage_mask=True)
        assert host.pbrh_set_gbuffer(r, *[np.ascontiguousarray(gb[k]).ctypes.data for k in ("A", "B", "C", "depth", "stencil")]) == 0
        assert host.pbrh_set_initial_luminance(r, 0.18) == 0
        assert host.pbrh_render(r, 1.0 / 60.0) == 0, host.pbrh_last_error(r)

        def read(name, shape, dtype):
            a = np.zeros(shape, dtype=dtype)
            n = host.pbrh_read(r, name.encode(), a.ctypes.data, a.nbytes)
            assert n == a.nbytes, (name, n, host.pbrh_last_error(r))
            return a
        hdr = read("DeferredShadingRT", (H, W, 4), np.float16)
        ldr = read("ToneMappedTexture", (H, W), np.uint32)
        avg = read("AverageLuminance", (1,), np.float32)[0]
        lights_dev = read("ClusteredLights", (1024,), LIGHT_DTYPE)
        env = read("PrefilterEnvMap", (cube_texels(ENV, 5), 4), np.float16)
        lut = read("PrecomputeBRDF", (LUT, LUT, 2), np.float16)
        g = Global()
        assert host.pbrh_get_global(r, C.byref(g)) == 0
    finally:
        host.pbrh_destroy(r)
    # the light buffer on the device = the records the CPU path derives (octree order, 7-unit preset)
    packed = np.ascontiguousarray(np.concatenate([recs["translation"], recs["color"], recs["radius"][:, None], recs["intensity"][:, None]], axis=1), np.float32)
    lights = np.zeros(16, LIGHT_DTYPE)
    cam4 = np.float32([0.0, 3.0, 10.0, 3.14159265359])
    assert host.pbrh_light_buffer(W, H, cam4.ctypes.data, packed.ctypes.data, 8, lights.ctypes.data, 16) == 8
    lights = lights[:8]
    assert lights_dev[:8].tobytes() == lights.tobytes() and not lights_dev[8:].tobytes().strip(b"\0")
    assert sorted(map(tuple, lights["Position"])) == sorted(map(tuple, recs["translation"]))
    # oracle frame on the device's IBL tables (their own parity: tests/test_gpu_parity.py)
    cl = orc.cluster_build(g)
    orc.cluster_cull(g, lights, cl)
    assert cl["NumLights"].max() >= 1                                   # the scene's lights do reach clusters of this view
    tile = Tile(0, 0, W, H, W, H)
    want, _ = orc.deferred_shade(g, tile, gb, lut, env, ENV, 5, cl, lights)
    no_lights, _ = orc.deferred_shade(g, tile, gb, lut, env, ENV, 5, orc.cluster_cull(g, lights[:0], orc.cluster_build(g)), lights[:0])
    lit = np.abs(want.astype(np.float32) - no_lights.astype(np.float32))[..., :3].max(axis=-1) > 1e-3
    assert lit.mean() > 0.02                                            # ... and light a visible share of the pixels
    sky_full = sky_np.copy()
    orc.cube_gen_mips(sky_full, ENV, int(np.log2(ENV)) + 1)
    orc.skybox(g, tile, sky_full, ENV, int(np.log2(ENV)) + 1, gb["stencil"], want)
    orc.bloom(want)
    hist = orc.lum_histogram(want)
    avg_want = orc.lum_average(hist, W * H, 1.0 / 60.0, 0.18)
    ldr_want = orc.tonemap(want, avg_want)
    scale = np.abs(want.astype(np.float32)[..., :3]).max()
    d = common.half_ulp_diff(hdr[..., :3], want[..., :3])
    assert (d > 2).mean() <= 1e-3, (d.max(), (d > 2).mean())
    assert np.abs(hdr.astype(np.float32) - want.astype(np.float32))[..., :3].max() <= 5e-3 * scale
    assert abs(float(avg) - avg_want) <= 1e-4 * abs(avg_want) + 1e-7
    sh = np.array([0, 8, 16], dtype=np.uint32)
    dl = np.abs(((ldr[..., None] >> sh) & 255).astype(np.int32) - ((ldr_want[..., None] >> sh) & 255).astype(np.int32))
    assert (dl > 1).mean() < 1e-3


@pytest.mark.gpu
def test_host_loads_hdr_cube_faces(host, orc, tmp_path):
    """LoadCubeMap: six Radiance .hdr faces (flat and run-length coded) -> sky cube on the GPU.  A frame rendered with
    the loaded sky equals the frame rendered with the same texels handed over as floats (oracle-decoded)."""
    import hdr_writer
    W, H, ENV, LUT = 160, 96, 16, 32
    sky_np = synth.env_cube(ENV)
    faces = sky_np.reshape(-1, 4)[:6 * ENV * ENV, :3].reshape(6, ENV, ENV, 3)
    rgbe = hdr_writer.float_to_rgbe(faces)
    for i, name in enumerate(["px", "nx", "py", "ny", "pz", "nz"]):
        (tmp_path / f"{name}.hdr").write_bytes(hdr_writer.encode_hdr(rgbe[i], rle=(i % 2 == 0)))
    decoded = orc.rgbe_decode(rgbe.reshape(-1, 4))
    assert np.abs(decoded[:, :3] - faces.reshape(-1, 3)).max() <= faces.max() / 128

    def frame(setup):
        err = C.create_string_buffer(256)
        r = host.pbrh_create(0, W, H, ENV, LUT, err, 256)
        assert r, err.value
        try:
            assert setup(r) == 0, host.pbrh_last_error(r)
            gb = synth.gbuffer_tile(0, 0, W, H, W, H, coverage_mask=True)
            assert host.pbrh_set_gbuffer(r, *[np.ascontiguousarray(gb[k]).ctypes.data for k in ("A", "B", "C", "depth", "stencil")]) == 0
            assert host.pbrh_set_initial_luminance(r, 0.18) == 0
            assert host.pbrh_render(r, 1.0 / 60.0) == 0, host.pbrh_last_error(r)
            a = np.zeros((H, W, 4), np.float16)
            assert host.pbrh_read(r, b"DeferredShadingRT", a.ctypes.data, a.nbytes) == a.nbytes
            return a
        finally:
            host.pbrh_destroy(r)

    from_files = frame(lambda r: host.pbrh_load_skybox(r, str(tmp_path).encode()))
    from_floats = frame(lambda r: host.pbrh_set_skybox(r, np.ascontiguousarray(decoded).ctypes.data, ENV))
    assert np.isfinite(from_files.astype(np.float32)).all() and from_files.astype(np.float32).max() > 0
    assert np.array_equal(from_files.view(np.uint16), from_floats.view(np.uint16))
    # a missing face is an error, not a crash
    (tmp_path / "nz.hdr").unlink()
    err = C.create_string_buffer(256)
    r = host.pbrh_create(0, W, H, ENV, LUT, err, 256)
    try:
        assert host.pbrh_load_skybox(r, str(tmp_path).encode()) != 0 and b"nz.hdr" in host.pbrh_last_error(r)
    finally:
        host.pbrh_destroy(r)


@pytest.mark.gpu
def test_host_fused_passes_equal_dispatch_by_dispatch(host):
    """pbrh_set_fused: ClusteredPass and BloomPass hand their dispatch sequences over as one call each — fewer launches,
    bit-identical frame (HDR, LDR, adapted luminance), over several frames of adapting exposure."""
    W, H, ENV, LUT = 512, 288, 32, 64          # exact 2x pyramid: the fused bloom kernels run
    sky_np = synth.env_cube(ENV)
    from direct12pbrrenderer_amd import scene
    cam = scene.Camera.reference_default(W, H)
    lights = synth.lights_in_view_box(64, cam)
    packed = np.ascontiguousarray(np.concatenate([lights["Position"], lights["Color"], np.full((64, 1), 2.0, np.float32), lights["Intensity"][:, None]], axis=1).astype(np.float32))
    gb = synth.gbuffer_tile(0, 0, W, H, W, H, coverage_mask=True)

    def run(fused, fused_later=None):
        err = C.create_string_buffer(256)
        r = host.pbrh_create(0, W, H, ENV, LUT, err, 256)
        assert r, err.value
        try:
            assert host.pbrh_set_fused(r, fused) == 0
            assert host.pbrh_set_skybox(r, sky_np[:4 * 6 * ENV * ENV].ctypes.data, ENV) == 0
            assert host.pbrh_set_lights(r, packed.ctypes.data, 64) == 0
            assert host.pbrh_set_gbuffer(r, *[np.ascontiguousarray(gb[k]).ctypes.data for k in ("A", "B", "C", "depth", "stencil")]) == 0
            assert host.pbrh_set_initial_luminance(r, 0.18) == 0
            assert host.pbrh_render(r, 1.0 / 60.0) == 0, host.pbrh_last_error(r)
            first = host.pbrh_dispatch_count(r)
            if fused_later is not None:   # the one-shot passes (env prefilter, LUT) have run: only the per-frame passes change from here on
                assert host.pbrh_set_fused(r, fused_later) == 0
            ms = C.c_double(0.0)
            assert host.pbrh_render_n(r, 3, 1.0 / 60.0, C.byref(ms)) == 0 and ms.value > 0.0
            later = host.pbrh_dispatch_count(r)
            out = {}
            for name, shape, dt in (("DeferredShadingRT", (H, W, 4), np.float16), ("ToneMappedTexture", (H, W), np.uint32), ("AverageLuminance", (1,), np.float32)):
                a = np.zeros(shape, dtype=dt)
                assert host.pbrh_read(r, name.encode(), a.ctypes.data, a.nbytes) == a.nbytes
                out[name] = a
            return first, later, out
        finally:
            host.pbrh_destroy(r)

    f0, l0, staged = run(0)
    f1, l1, fused = run(1)
    assert (f0, l0) == (29, 23) and (f1, l1) == (1 + 1 + 1 + 1 + 1 + 1 + 2 + 1, 7)
    # The PER-FRAME passes (Clustered, Bloom + histogram) are bit-identical fused or not: same env chain (first frame dispatch by
    # dispatch in both runs), then three frames staged vs three frames fused — HDR, LDR and the adapted luminance bit for bit.
    # (with the prefilter folded into SetFusedPasses no test pinned this any more.)
    f2, l2, mixed = run(0, fused_later=1)
    assert (f2, l2) == (29, 7)
    assert np.array_equal(staged["DeferredShadingRT"].view(np.uint16), mixed["DeferredShadingRT"].view(np.uint16))
    assert np.array_equal(staged["ToneMappedTexture"], mixed["ToneMappedTexture"])
    assert staged["AverageLuminance"].view(np.uint32)[0] == mixed["AverageLuminance"].view(np.uint32)[0]
    # since round 5 PreFilterEnvMapPass hands its five dispatches over as one pbr_prefilter_env as well, and THAT chain is within
    # 1 fp16 ULP of the dispatch-by-dispatch one (next test), not identical: the all-fused frames agree with the staged ones to the
    # ULP that env texel moves an HDR value by
    d = common.half_ulp_diff(staged["DeferredShadingRT"][..., :3], fused["DeferredShadingRT"][..., :3])
    assert d.max() <= 2 and (d == 0).mean() >= 0.98, f"fused frame: {int(d.max())} fp16 ULP, {(d == 0).mean():.4f} identical"
    lb = lambda a: ((a[..., None] >> np.array([0, 8, 16], dtype=np.uint32)) & 255).astype(np.int32)   # noqa: E731
    assert np.abs(lb(staged["ToneMappedTexture"]) - lb(fused["ToneMappedTexture"])).max() <= 1
    assert abs(float(staged["AverageLuminance"][0]) - float(fused["AverageLuminance"][0])) <= 1e-3 * float(staged["AverageLuminance"][0])
    assert staged["AverageLuminance"][0] != np.float32(0.18)


@pytest.mark.gpu
@pytest.mark.parametrize("env", [64, 512])
def test_host_fused_prefilter_pass_matches_the_five_dispatches(host, golden2, env):
    """PreFilterEnvMapPass::Execute (DeferredPipeline.cpp:77-115) under SetFusedPasses: env_map_gen.hlsl's five dispatches as ONE
    pbr_prefilter_env call.  Both chains read back through the pass API's PrefilterEnvMap resource: 64^2 — every texel of the five
    mips within 1 fp16 ULP (or 1e-3) of the dispatch-by-dispatch chain; 512^2 (the reference's size, the bench sky) — the 4 096
    texels golden_v2.npz pins, both chains against the fixture and against each other."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_golden_v2 as mk
    W, H, LUT = 64, 36, 32
    sky = mk.bench_sky() if env == 512 else synth.env_cube(env)
    gb = synth.gbuffer_tile(0, 0, W, H, W, H)

    def chain(fused):
        err = C.create_string_buffer(256)
        r = host.pbrh_create(0, W, H, env, LUT, err, 256)
        assert r, err.value
        try:
            assert host.pbrh_set_fused(r, fused) == 0
            assert host.pbrh_set_skybox(r, sky[:4 * 6 * env * env].ctypes.data, env) == 0
            assert host.pbrh_set_gbuffer(r, *[np.ascontiguousarray(gb[k]).ctypes.data for k in ("A", "B", "C", "depth", "stencil")]) == 0
            assert host.pbrh_render(r, 1.0 / 60.0) == 0, host.pbrh_last_error(r)
            n = host.pbrh_dispatch_count(r)
            a = np.zeros((cube_texels(env, 5), 4), dtype=np.float16)
            assert host.pbrh_read(r, b"PrefilterEnvMap", a.ctypes.data, a.nbytes) == a.nbytes
            return n, a
        finally:
            host.pbrh_destroy(r)

    n0, staged = chain(0)
    n1, fused = chain(1)
    assert n0 - n1 >= 4          # five env_map_gen dispatches became one call (the other fused passes account for the rest)
    f32 = lambda a: a.astype(np.float32)   # noqa: E731

    def close(a, b):
        return (common.half_ulp_diff(a, b) <= 1) | (np.abs(f32(a) - f32(b)) <= 1e-3 * np.abs(f32(b)))

    if env == 64:
        ok = close(fused, staged)
        assert ok.all(), f"fused prefilter chain: {(~ok).sum()} of {ok.size} values outside 1 fp16 ULP / 1e-3 of the five dispatches"
        assert np.all(fused[:, 3] == 1.0)
    else:
        from direct12pbrrenderer_amd.structs import cube_mip_offset
        for m in range(mk.ENV_MIPS):
            idx = golden2[f"env512_m{m}_idx"].astype(np.int64) + cube_mip_offset(env, m)
            want = golden2[f"env512_m{m}_texels"]
            for name, got in (("five dispatches", staged[idx]), ("fused", fused[idx])):
                ok = close(got, want)
                assert ok.all(), f"{name}, mip {m}: {(~ok).sum()} of {len(idx)} texels outside 1 ULP / 1e-3 of the fixture"
            assert close(fused[idx], staged[idx]).all()
    print(f"prefilter through the pass API, {env}^2 x 5: fused vs five dispatches worst {int(common.half_ulp_diff(fused, staged).max())} fp16 ULP, "
          f"{(common.half_ulp_diff(fused, staged) == 0).mean() * 100:.2f} % identical")


@pytest.mark.gpu
def test_host_graph_two_tiles_with_apron_reproduce_the_single_frame(host):
    """SURVEY 8e through the C++ pass graph: the frame cut into two tiles, each rendered by its own
    DeferredRenderPipeline on an apron-extended target (pbrh_set_tile: global-pixel addressing, interior-only histogram
    and tone-map, full-frame PixelCount).  The 1 KiB histogram exchange goes through the host here (one GPU: the tiles
    are rendered one after the other; with one process per GPU pbrh_comm_init makes the same pass all-reduce over RCCL).
    Interiors must equal the single-frame render: HDR <= 2 fp16 ulp, identical exposure, LDR <= 1 LSB."""
    from direct12pbrrenderer_amd import scene
    from direct12pbrrenderer_amd.pipeline import tile_for_rank
    TW, H, ENV, LUT, NL = 512, 288, 32, 64, 64
    W = 2 * TW
    sky_np = synth.env_cube(ENV)
    cam = scene.Camera.reference_default(W, H)
    lights = synth.lights_in_view_box(NL, cam)
    packed = np.ascontiguousarray(np.concatenate([lights["Position"], lights["Color"], np.full((NL, 1), 2.0, np.float32),
                                                  lights["Intensity"][:, None]], axis=1).astype(np.float32))
    err = C.create_string_buffer(256)

    def make(spec):
        r = host.pbrh_create(0, spec.ew, spec.eh, ENV, LUT, err, 256)
        assert r, err.value
        assert host.pbrh_set_skybox(r, sky_np[:4 * 6 * ENV * ENV].ctypes.data, ENV) == 0, host.pbrh_last_error(r)
        assert host.pbrh_set_lights(r, packed.ctypes.data, NL) == 0
        gb = synth.gbuffer_tile(spec.ex0, spec.ey0, spec.ew, spec.eh, spec.full_w, spec.full_h, coverage_mask=False)
        assert host.pbrh_set_gbuffer(r, *[np.ascontiguousarray(gb[k]).ctypes.data for k in ("A", "B", "C", "depth", "stencil")]) == 0
        assert host.pbrh_set_initial_luminance(r, 0.18) == 0
        if spec.apron:
            assert host.pbrh_set_tile(r, spec.ex0, spec.ey0, spec.full_w, spec.full_h, spec.ix, spec.iy, spec.w, spec.h) == 0, host.pbrh_last_error(r)
        return r

    def read(r, name, shape, dtype):
        a = np.zeros(shape, dtype=dtype)
        assert host.pbrh_read(r, name.encode(), a.ctypes.data, a.nbytes) == a.nbytes, host.pbrh_last_error(r)
        return a

    from direct12pbrrenderer_amd.pipeline import TileSpec
    full_spec = TileSpec(0, 0, W, H, W, H, 0)
    rf = make(full_spec)
    try:
        assert host.pbrh_render(rf, 1.0 / 60.0) == 0, host.pbrh_last_error(rf)
        hdr_f = read(rf, "DeferredShadingRT", (H, W, 4), np.float16)
        ldr_f = read(rf, "ToneMappedTexture", (H, W), np.uint32)
        avg_f = read(rf, "AverageLuminance", (1,), np.float32)[0]
    finally:
        host.pbrh_destroy(rf)
    specs = [tile_for_rank(k, 2, TW, H) for k in range(2)]
    assert specs[0].apron == 256 and (specs[1].ex0, specs[1].ix) == (TW - 256, 256)
    # pass 1: every tile's own histogram (captured before the average); pass 2: the other tile's counts added
    hists = []
    for spec in specs:
        r = make(spec)
        try:
            assert host.pbrh_capture_histogram(r, 1) == 0
            assert host.pbrh_render(r, 1.0 / 60.0) == 0, host.pbrh_last_error(r)
            h = np.zeros(256, np.uint32)
            assert host.pbrh_captured_histogram(r, h.ctypes.data) == 0
            assert h.sum() == spec.w * spec.h            # interior pixels only
            hists.append(h)
        finally:
            host.pbrh_destroy(r)
    for k, spec in enumerate(specs):
        r = make(spec)
        try:
            other = np.ascontiguousarray(hists[1 - k])
            assert host.pbrh_set_external_histogram(r, other.ctypes.data) == 0
            assert host.pbrh_render(r, 1.0 / 60.0) == 0, host.pbrh_last_error(r)
            hdr = read(r, "DeferredShadingRT", (spec.eh, spec.ew, 4), np.float16)[spec.iy:spec.iy + spec.h, spec.ix:spec.ix + spec.w]
            ldr = read(r, "ToneMappedTexture", (spec.eh, spec.ew), np.uint32)[spec.iy:spec.iy + spec.h, spec.ix:spec.ix + spec.w]
            avg = read(r, "AverageLuminance", (1,), np.float32)[0]
        finally:
            host.pbrh_destroy(r)
        assert avg == pytest.approx(float(avg_f), rel=1e-6)
        d = common.half_ulp_diff(hdr[..., :3], hdr_f[spec.y0:spec.y0 + spec.h, spec.x0:spec.x0 + spec.w, :3])
        assert d.max() <= 2 and (d > 0).mean() < 2e-3, (k, d.max(), (d > 0).mean())
        b = ldr_f[spec.y0:spec.y0 + spec.h, spec.x0:spec.x0 + spec.w]
        for c in range(3):
            assert np.abs(((ldr >> (8 * c)) & 255).astype(np.int32) - ((b >> (8 * c)) & 255).astype(np.int32)).max() <= 1


@pytest.mark.gpu
def test_host_graph_comm_contract(host):
    """pbrh_comm_init forwards to the context's RCCL communicator: world 1 without an id is a no-op, a missing id at
    world 2 is reported (the pass graph then refuses to average with world > 1 and no communicator)."""
    err = C.create_string_buffer(256)
    r = host.pbrh_create(0, 64, 64, 16, 32, err, 256)
    assert r, err.value
    try:
        assert host.pbrh_comm_init(r, 1, 0, None) == 0
        assert host.pbrh_comm_init(r, 2, 0, None) == -1 and b"unique id" in host.pbrh_last_error(r)
        assert host.pbrh_set_tile(r, 0, 0, 32, 64, 0, 0, 32, 64) == -1      # target larger than the frame
    finally:
        host.pbrh_destroy(r)


@pytest.mark.gpu
@pytest.mark.parametrize("cols,rows,tw,th,fused", [(2, 1, 512, 288, 0), (2, 2, 384, 288, 1), (3, 1, 320, 320, 1)])
def test_host_graph_halo_tiles_reproduce_the_single_frame(host, cols, rows, tw, th, fused):
    """SURVEY 8e option 2 through the C++ pass graph: every tile is a DeferredRenderPipeline created by pbrh_create_tile in
    HALO mode — targets cover interior + 4 px, BloomPass::Execute issues pbr_bloom_prefilter_rect -> halo exchange ->
    pbr_bloom_tiled on the extended rectangle.  One GPU here, so the tiles are rendered one after the other with the
    loopback transport and the level-1 strips are copied between their staging areas device to device
    (pbrh_halo_copy_from) — with one process per GPU the same pass calls pbr_halo_exchange over RCCL.  Interiors must
    equal the single frame: HDR <= 2 fp16 ulp, the tile histograms sum to the frame's, identical exposure, LDR <= 1 LSB.
    fused = 1: the command list merges the pyramid with the histogram dispatch that follows it (same results)."""
    from direct12pbrrenderer_amd import scene
    from direct12pbrrenderer_amd.pipeline import tile_of_frame
    ENV, LUT, NL = 32, 64, 64
    W, H, world = cols * tw, rows * th, cols * rows
    sky_np = synth.env_cube(ENV)
    cam = scene.Camera.reference_default(W, H)
    lights = synth.lights_in_view_box(NL, cam)
    packed = np.ascontiguousarray(np.concatenate([lights["Position"], lights["Color"], np.full((NL, 1), 2.0, np.float32),
                                                  lights["Intensity"][:, None]], axis=1).astype(np.float32))
    err = C.create_string_buffer(256)

    def fill(r, x0, y0, w, h):
        assert host.pbrh_set_skybox(r, sky_np[:4 * 6 * ENV * ENV].ctypes.data, ENV) == 0, host.pbrh_last_error(r)
        assert host.pbrh_set_lights(r, packed.ctypes.data, NL) == 0
        gb = synth.gbuffer_tile(x0, y0, w, h, W, H, coverage_mask=False)
        assert host.pbrh_set_gbuffer(r, *[np.ascontiguousarray(gb[k]).ctypes.data for k in ("A", "B", "C", "depth", "stencil")]) == 0
        assert host.pbrh_set_initial_luminance(r, 0.18) == 0
        assert host.pbrh_set_fused(r, fused) == 0

    def read(r, name, shape, dtype):
        a = np.zeros(shape, dtype=dtype)
        assert host.pbrh_read(r, name.encode(), a.ctypes.data, a.nbytes) == a.nbytes, host.pbrh_last_error(r)
        return a

    rf = host.pbrh_create(0, W, H, ENV, LUT, err, 256)
    assert rf, err.value
    try:
        fill(rf, 0, 0, W, H)
        assert host.pbrh_capture_histogram(rf, 1) == 0
        assert host.pbrh_render(rf, 1.0 / 60.0) == 0, host.pbrh_last_error(rf)
        hdr_f = read(rf, "DeferredShadingRT", (H, W, 4), np.float16)
        ldr_f = read(rf, "ToneMappedTexture", (H, W), np.uint32)
        avg_f = read(rf, "AverageLuminance", (1,), np.float32)[0]
        hist_f = np.zeros(256, np.uint32)
        assert host.pbrh_captured_histogram(rf, hist_f.ctypes.data) == 0
    finally:
        host.pbrh_destroy(rf)

    specs = [tile_of_frame(k, world, W, H, layout=(cols, rows), halo=True) for k in range(world)]
    tiles = []
    try:
        for k, s in enumerate(specs):
            r = host.pbrh_create_tile(0, W, H, cols, rows, k, 1, ENV, LUT, err, 256)
            assert r, err.value
            tiles.append(r)
            fill(r, s.sx0, s.sy0, s.sw, s.sh)
            assert host.pbrh_set_halo_loopback(r, 1) == 0
            assert host.pbrh_capture_histogram(r, 1) == 0
        # frame 1: every tile computes and packs the level-1 strips of its interior (what it receives is still empty)
        for r in tiles:
            assert host.pbrh_render(r, 1.0 / 60.0) == 0, host.pbrh_last_error(r)
        for a in tiles:
            for b in tiles:
                if a is not b:
                    assert host.pbrh_halo_copy_from(a, b) == 0, host.pbrh_last_error(a)
        # frame 2: the strips are in place -> every tile's own histogram
        hists = []
        for r, s in zip(tiles, specs):
            assert host.pbrh_set_initial_luminance(r, 0.18) == 0
            assert host.pbrh_render(r, 1.0 / 60.0) == 0, host.pbrh_last_error(r)
            h = np.zeros(256, np.uint32)
            assert host.pbrh_captured_histogram(r, h.ctypes.data) == 0
            assert h.sum() == s.w * s.h            # interior pixels only
            hists.append(h)
        total = np.sum(hists, axis=0, dtype=np.uint64)
        # the same pixels land in the same bins except where a <= 2-ulp HDR difference crosses a bin edge
        assert total.sum() == W * H and np.abs(total.astype(np.int64) - hist_f.astype(np.int64)).sum() <= max(4, W * H // 20000)
        # frame 3: the other tiles' counts through the host (what pbr_allreduce_hist does over RCCL)
        for k, (r, s) in enumerate(zip(tiles, specs)):
            other = np.ascontiguousarray((total - hists[k]).astype(np.uint32))
            assert host.pbrh_set_external_histogram(r, other.ctypes.data) == 0
            assert host.pbrh_set_initial_luminance(r, 0.18) == 0
            assert host.pbrh_render(r, 1.0 / 60.0) == 0, host.pbrh_last_error(r)
            assert host.pbrh_dispatch_count(r) == (1 + 1 + 1 + 1 + 2 + 1 if fused else 2 + 1 + 1 + 1 + 2 + 1)   # clustered, sky, shade, bloom (halo), histogram + average, tone-map
            hdr = read(r, "DeferredShadingRT", (s.sh, s.sw, 4), np.float16)[s.siy:s.siy + s.h, s.six:s.six + s.w]
            ldr = read(r, "ToneMappedTexture", (s.sh, s.sw), np.uint32)[s.siy:s.siy + s.h, s.six:s.six + s.w]
            avg = read(r, "AverageLuminance", (1,), np.float32)[0]
            assert avg == pytest.approx(float(avg_f), rel=1e-6)
            d = common.half_ulp_diff(hdr[..., :3], hdr_f[s.y0:s.y0 + s.h, s.x0:s.x0 + s.w, :3])
            assert d.max() <= 2 and (d > 0).mean() < 2e-3, (k, d.max(), (d > 0).mean())
            b = ldr_f[s.y0:s.y0 + s.h, s.x0:s.x0 + s.w]
            for c in range(3):
                assert np.abs(((ldr >> (8 * c)) & 255).astype(np.int32) - ((b >> (8 * c)) & 255).astype(np.int32)).max() <= 1
    finally:
        for r in tiles:
            host.pbrh_destroy(r)


@pytest.mark.gpu
def test_host_graph_throughput_mode_renders_the_same_frames(host):
    """pbrh_set_frames_in_flight(3): a frame's end waits for frame i - 2 only (the host records ahead of the GPU); the
    frames themselves — adapted luminance after five of them, LDR image — are those of the reference's fence-per-frame loop.
    pbrh_set_tail_overlap(1) on top: average + tone-map (+ the all-reduce with a communicator) of frame i on the context's side
    stream beside frame i + 1's shade, HDR target and histogram double-buffered by the frame graph — still the same frames;
    pbrh_set_tail_overlap(2): the side stream takes over at the bloom pass (fused and dispatch by dispatch)."""
    from direct12pbrrenderer_amd import scene
    W, H, ENV, LUT, NL = 512, 288, 32, 64, 256
    sky_np = synth.env_cube(ENV)
    cam = scene.Camera.reference_default(W, H)
    lights = synth.lights_in_view_box(NL, cam)
    packed = np.ascontiguousarray(np.concatenate([lights["Position"], lights["Color"], np.full((NL, 1), 2.0, np.float32),
                                                  lights["Intensity"][:, None]], axis=1).astype(np.float32))
    gb = synth.gbuffer_tile(0, 0, W, H, W, H, coverage_mask=True)
    err = C.create_string_buffer(256)
    out = []
    configs = ((1, 0, 0), (1, 1, 0), (3, 1, 0), (3, 0, 0), (3, 1, 1), (3, 0, 1), (3, 1, 2), (3, 0, 2))
    for in_flight, fused, tail in configs:
        r = host.pbrh_create(0, W, H, ENV, LUT, err, 256)
        assert r, err.value
        try:
            assert host.pbrh_set_skybox(r, sky_np[:4 * 6 * ENV * ENV].ctypes.data, ENV) == 0
            assert host.pbrh_set_lights(r, packed.ctypes.data, NL) == 0
            assert host.pbrh_set_gbuffer(r, *[np.ascontiguousarray(gb[k]).ctypes.data for k in ("A", "B", "C", "depth", "stencil")]) == 0
            assert host.pbrh_set_initial_luminance(r, 0.18) == 0
            assert host.pbrh_set_fused(r, fused) == 0
            assert host.pbrh_set_frames_in_flight(r, in_flight) == 0
            # a refused call changes nothing (it used to drop the renderer back to the per-frame fence first)
            assert host.pbrh_set_frames_in_flight(r, 9) == -1 and b"at most" in host.pbrh_last_error(r)
            if in_flight > 1:
                assert host.pbrh_set_tail_overlap(r, 1) == 0, host.pbrh_last_error(r)      # still in throughput mode after the refusal
                assert host.pbrh_set_tail_overlap(r, 3) == -1 and b"mode 0" in host.pbrh_last_error(r)
                assert host.pbrh_set_frames_in_flight(r, 1) == 0                             # back to the fence: the tail mode goes with it
                assert host.pbrh_set_frames_in_flight(r, in_flight) == 0
            if in_flight == 1:
                assert host.pbrh_set_tail_overlap(r, 1) == -1 and b"throughput mode only" in host.pbrh_last_error(r)
            assert host.pbrh_set_tail_overlap(r, tail) == 0, host.pbrh_last_error(r)
            ms = C.c_double(0.0)
            assert host.pbrh_render_n(r, 5, 1.0 / 60.0, C.byref(ms)) == 0, host.pbrh_last_error(r)
            ldr = np.zeros((H, W), np.uint32)
            avg = np.zeros(1, np.float32)
            assert host.pbrh_read(r, b"ToneMappedTexture", ldr.ctypes.data, ldr.nbytes) == ldr.nbytes
            assert host.pbrh_read(r, b"AverageLuminance", avg.ctypes.data, 4) == 4
            out.append((ldr, float(avg[0])))
        finally:
            host.pbrh_destroy(r)
    # every throughput-mode configuration against the fence-per-frame loop with the same pass fusion (configs[0] / configs[1]): fused
    # passes prefilter the env chain in one call since round 5, within 1 fp16 ULP of the five dispatches but not identical, so a fused
    # frame is compared with a fused frame; the two baselines agree to 1 LSB
    base = {f: out[k] for k, (nf, f, t) in enumerate(configs) if nf == 1 for f in [f]}
    bad = [(k, avg == base[configs[k][1]][1], int((ldr != base[configs[k][1]][0]).sum())) for k, (ldr, avg) in enumerate(out)
           if avg != base[configs[k][1]][1] or not np.array_equal(ldr, base[configs[k][1]][0])]
    assert not bad, f"(config index, same adapted luminance, differing LDR pixels): {bad}"
    lb = lambda a: ((a[..., None] >> np.array([0, 8, 16], dtype=np.uint32)) & 255).astype(np.int32)   # noqa: E731
    assert np.abs(lb(base[0][0]) - lb(base[1][0])).max() <= 1 and abs(base[0][1] - base[1][1]) <= 1e-3 * base[0][1]
