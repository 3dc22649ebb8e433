"""Host-side behaviour that needs no GPU: the C-ABI library exports what include/pbr_hip.h declares, refuses to
run on a mixed ROCm runtime with a clear message, and bench.py's rank spawner."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from direct12pbrrenderer_amd import _lib
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "pbr_hip.h")).read()
    declared = set(re.findall(r"\b(pbr_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 45
    for name in declared:
        assert hasattr(lib, name), f"{name} is declared in include/pbr_hip.h but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature in _lib.SIGNATURES"
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
    lib = ctypes.CDLL(os.path.join(ROOT, "direct12pbrrenderer_amd", "libpbr_host.so"))
    header = open(os.path.join(ROOT, "direct12pbrrenderer_amd", "host", "pbr_host.h")).read()
    declared = set(re.findall(r"\b(pbrh_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), f"{name} is declared in pbr_host.h but not exported"
