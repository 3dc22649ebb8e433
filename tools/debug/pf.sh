set -o pipefail
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -s -k "prefilter or sh9 or cube" 2>&1 | tail -4 || exit 1
python tools/cfg3_ms.py texel-per-lane
