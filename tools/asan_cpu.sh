#!/bin/bash
# CPU-side sanitizer run (build container or any box; no GPU used): builds the C++ host pass graph and the oracle with
# -fsanitize=address,undefined and runs the CPU test files that exercise them against those builds.
#   tools/asan_cpu.sh [out_file]        (default profiles/r05_asan_cpu.txt)
set -u
cd "$(dirname "$0")/.."
OUT=${1:-profiles/r05_asan_cpu.txt}
make -C direct12pbrrenderer_amd/host asan -j4 >/dev/null || exit 1
make -C oracle asan >/dev/null || exit 1
ASAN_LIB=$(g++ -print-file-name=libasan.so)
UBSAN_LIB=$(g++ -print-file-name=libubsan.so)
{
  echo "# ASan + UBSan, CPU side: libpbr_host.so (C++ pass graph, file parsers, light cull, tile layout) and libpbr_oracle.so (the checker)"
  echo "# g++ $(g++ -dumpversion), flags: -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined; $(date -u +%F)"
  echo "# command: LD_PRELOAD=libasan.so:libubsan.so ASAN_OPTIONS=detect_leaks=0 PBR_TEST_HOST_LIB=direct12pbrrenderer_amd/asan/libpbr_host.so"
  echo "#          PBR_TEST_ORACLE_LIB=oracle/asan/libpbr_oracle.so python -m pytest tests/test_host.py tests/test_runtime_cpu.py tests/test_oracle_kat.py tests/test_golden_cpu.py -q -m 'not gpu'"
  echo "# (detect_leaks=0: the interpreter itself is not leak-clean; every other check — heap/stack/global overflow, use-after-free, UB — is on and fatal)"
} > "$OUT"
LD_PRELOAD="$ASAN_LIB:$UBSAN_LIB" ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
  PBR_TEST_HOST_LIB="$PWD/direct12pbrrenderer_amd/asan/libpbr_host.so" PBR_TEST_ORACLE_LIB="$PWD/oracle/asan/libpbr_oracle.so" OMP_NUM_THREADS=4 \
  python -m pytest tests/test_host.py tests/test_runtime_cpu.py tests/test_oracle_kat.py tests/test_golden_cpu.py -q -m "not gpu" -p no:cacheprovider 2>&1 | tail -40 >> "$OUT"
rc=${PIPESTATUS[0]}
# which builds the test process really mapped (a silent fall-back to the plain libraries would make the run meaningless)
LD_PRELOAD="$ASAN_LIB:$UBSAN_LIB" ASAN_OPTIONS=detect_leaks=0 PBR_TEST_HOST_LIB="$PWD/direct12pbrrenderer_amd/asan/libpbr_host.so" \
  PBR_TEST_ORACLE_LIB="$PWD/oracle/asan/libpbr_oracle.so" python - >> "$OUT" 2>/dev/null <<'PY'
import ctypes, sys
sys.path[:0] = ["tests", "."]
import common
from oracle import binding
binding.lib()
import torch  # noqa: F401
ctypes.CDLL(common.host_lib_path())
print("# mapped:", sorted({l.split()[-1] for l in open("/proc/self/maps") if ("libpbr_" in l or "libasan" in l or "libubsan" in l) and "r-xp" in l}))
PY
echo "# exit code $rc" >> "$OUT"
cat "$OUT"
exit $rc
