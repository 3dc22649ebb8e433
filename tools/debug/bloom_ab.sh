python -m pytest tests/test_gpu_parity.py tests/test_gpu_tiling.py -m gpu -x -q 2>&1 | tail -3
python tools/shade_ms.py new 2>&1 | grep "bloom\|shade"
