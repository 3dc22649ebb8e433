"""Host-side behaviour that needs no GPU: the C-ABI library exports what include/pbr_hip.h declares, refuses to
run on a mixed ROCm runtime with a clear message, and bench.py's rank spawner."""
import os
import re
import subprocess
import sys

import common

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from direct12pbrrenderer_amd import _lib
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "pbr_hip.h")).read()
    # the header's last section (#ifdef PBR_DEBUG_KNOBS) declares the measurement entry points of the knobs build: not product exports
    knobs_section = re.search(r"#ifdef PBR_DEBUG_KNOBS\n(.*?)#endif", header, re.S)
    assert knobs_section
    knobs_only = set(re.findall(r"\b(pbr_[a-z0-9_]+)\s*\(", knobs_section.group(1)))
    declared = set(re.findall(r"\b(pbr_[a-z0-9_]+)\s*\(", header.replace(knobs_section.group(0), ""))) - knobs_only
    assert len(declared) >= 45 and knobs_only == set(_lib.KNOBS_ONLY) == {"pbr_ctx_set_cu_masks"}
    for name in declared:
        assert hasattr(lib, name), f"{name} is declared in include/pbr_hip.h but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature in _lib.SIGNATURES"
    import ctypes
    knobs = ctypes.CDLL(os.path.join(ROOT, "direct12pbrrenderer_amd", "libpbr_hip_knobs.so"))
    for name in knobs_only:     # ... exported by the knobs build, absent from the product library
        assert hasattr(knobs, name) and not hasattr(lib, name), name
    for name in declared:
        assert hasattr(knobs, name), f"{name} missing from the knobs build"
    assert lib.pbr_runtime_error() is None          # torch first, then the library: one ROCm installation


def test_library_loaded_before_torch_is_refused_with_a_clear_message():
    """libpbr_hip.so first maps /opt/rocm's libamdhip64; torch imported afterwards ends up on that copy instead of the
    one it ships.  pbr_ctx_create must say so instead of failing later inside the HIP runtime (round-1 failure)."""
    code = f"""
import ctypes, os
lib = ctypes.CDLL(os.path.join({ROOT!r}, "direct12pbrrenderer_amd", "libpbr_hip.so"))
lib.pbr_runtime_error.restype = ctypes.c_char_p
assert lib.pbr_runtime_error() is None
import torch
msg = lib.pbr_runtime_error()
h = ctypes.c_void_p()
st = lib.pbr_ctx_create(0, ctypes.byref(h))
print("STATUS", st, (msg or b"").decode())
"""
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("STATUS")][0]
    assert line.split()[1] == "-4" and "load torch BEFORE" in line, line      # PBR_ERR_UNSUPPORTED
    assert "two ROCm installations" in r.stderr


def test_bench_spawner_refuses_when_no_device_is_visible():
    """`python bench.py --gpus 2` from a bare shell starts its own ranks; with no GPU at all it must say so and exit
    non-zero before touching anything (this container has no GPU)."""
    import torch
    if torch.cuda.device_count() > 0:
        return
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 2 and "device(s) visible" in r.stderr, (r.returncode, r.stderr[-1000:])


def test_host_library_exports_every_declared_symbol():
    """libpbr_host.so (the C++ pass graph) exports what direct12pbrrenderer_amd/host/pbr_host.h declares."""
    import ctypes
    import torch  # noqa: F401  (its ROCm runtime first: see _lib.load)
    lib = ctypes.CDLL(common.host_lib_path())
    header = open(os.path.join(ROOT, "direct12pbrrenderer_amd", "host", "pbr_host.h")).read()
    declared = set(re.findall(r"\b(pbrh_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), f"{name} is declared in pbr_host.h but not exported"


def test_runtime_mismatch_rule_ignores_a_torch_without_a_bundled_runtime(tmp_path):
    """pbr_runtime_mismatch_dirs: a mismatch only when PyTorch's directory really holds its own libamdhip64 and the
    runtime in use comes from elsewhere; a PyTorch built against the system ROCm (nothing bundled) is one runtime."""
    from direct12pbrrenderer_amd import _lib
    lib = _lib.load()
    bundled, bare = tmp_path / "torch_bundled", tmp_path / "torch_bare"
    bundled.mkdir()
    bare.mkdir()
    (bundled / "libamdhip64.so").write_bytes(b"")
    rocm = b"/opt/rocm/lib"
    assert lib.pbr_runtime_mismatch_dirs(rocm, str(bundled).encode()) == 1      # two runtimes: refuse
    assert lib.pbr_runtime_mismatch_dirs(rocm, str(bare).encode()) == 0         # torch uses the system runtime too
    assert lib.pbr_runtime_mismatch_dirs(str(bundled).encode(), str(bundled).encode()) == 0
    assert lib.pbr_runtime_mismatch_dirs(rocm, b"") == 0 and lib.pbr_runtime_mismatch_dirs(b"", str(bundled).encode()) == 0


def test_host_tile_layout_matches_the_python_tiling_and_halo_plan():
    """TileLayout (C++ pass graph) against pipeline.py's TileSpec / halo_plan, which the gloo tests pin against the single
    frame: same interior / shaded / bloom rectangles and the same level-1 strips, for cfg5's 2 rows x 4 cols and other grids."""
    import ctypes
    import numpy as np
    import torch  # noqa: F401
    from direct12pbrrenderer_amd.pipeline import halo_plan, tile_of_frame
    lib = ctypes.CDLL(common.host_lib_path())
    lib.pbrh_tile_layout.argtypes = [ctypes.c_uint32] * 5 + [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    for fw, fh, cols, rows in ((7680, 4320, 4, 2), (7680, 4320, 2, 1), (7680, 4320, 2, 2), (3072, 576, 3, 1), (1024, 2048, 1, 4), (3840, 2160, 1, 1)):
        world = cols * rows
        for halo in (False, True):
            specs = [tile_of_frame(r, world, fw, fh, layout=(cols, rows), halo=halo) for r in range(world)]
            for rank, s in enumerate(specs):
                rects = np.zeros(12, np.uint32)
                peers = np.zeros((16, 9), np.int32)
                n = lib.pbrh_tile_layout(fw, fh, cols, rows, rank, int(halo), rects.ctypes.data, peers.ctypes.data, 16)
                assert n >= 0
                assert rects.tolist() == [s.x0, s.y0, s.w, s.h, s.sx0, s.sy0, s.sw, s.sh, s.ex0, s.ey0, s.ew, s.eh], (fw, fh, cols, rows, rank, halo)
                hx, hy = s.ex0 // 2, s.ey0 // 2
                want = []
                if halo and world > 1:
                    for peer, snd, rcv in halo_plan(rank, world, specs):
                        loc = lambda q: [0, 0, 0, 0] if q is None else [q[0] - hx, q[1] - hy, q[2] - q[0], q[3] - q[1]]
                        want.append([peer] + loc(snd) + loc(rcv))
                assert n == len(want) and peers[:n].tolist() == want, (fw, fh, cols, rows, rank, halo)
    assert lib.pbrh_tile_layout(7680, 4320, 4, 2, 8, 1, None, None, 0) == -1       # rank outside the grid
    assert lib.pbrh_tile_layout(7680, 4320, 7, 2, 0, 1, None, None, 0) == -1       # 7680 / 7 is not integral
    assert lib.pbrh_tile_layout(1000, 1000, 2, 1, 0, 1, None, None, 0) == -1       # 500-px tiles are not multiples of 16


def test_product_library_never_reads_the_environment():
    """The tuning / A-B switches of the launch code (PBR_* variables) exist in the knobs build only
    (libpbr_hip_knobs.so, -DPBR_DEBUG_KNOBS): the product library does not even import getenv."""
    lib = os.path.join(ROOT, "direct12pbrrenderer_amd", "libpbr_hip.so")
    knobs = os.path.join(ROOT, "direct12pbrrenderer_amd", "libpbr_hip_knobs.so")
    und = lambda path: subprocess.run(["nm", "-D", "--undefined-only", path], capture_output=True, text=True, check=True).stdout
    assert "getenv" not in und(lib)
    assert "getenv" in und(knobs)


def test_ctx_builds_without_the_rccl_development_header(tmp_path):
    """ctx.hip restates the few RCCL declarations it needs for build boxes without <rccl/rccl.h>; -DPBR_NO_RCCL_HEADER forces that
    branch (it had never been compiled and did not compile).  Host-only syntax pass: no GPU, no link."""
    src = os.path.join(ROOT, "direct12pbrrenderer_amd", "csrc", "ctx.hip")
    for extra in (["-DPBR_NO_RCCL_HEADER"], []):
        r = subprocess.run(["/opt/rocm/bin/hipcc", "-std=c++17", "--offload-arch=gfx950", "--cuda-host-only", "-fsyntax-only", "-Wall",
                            "-Wno-unused-function", "-I" + os.path.join(ROOT, "include")] + extra + [src],
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
