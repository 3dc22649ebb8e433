python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
for i in 1 2; do
python tools/cfg2_ms.py "footprint layout" 2>&1 | grep Gpx
PBR_HIP_LIB=$PWD/tools/debug/old_csrc/libpbr_hip_old.so python tools/cfg2_ms.py "border layout (HEAD)" 2>&1 | grep Gpx
done
