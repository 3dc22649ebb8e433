set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "bloom" -rs > gpurun_out/r04_b_pytest_bloom.log 2>&1; rc=$?
tail -n 12 gpurun_out/r04_b_pytest_bloom.log
[ $rc -eq 0 ] || exit $rc
bash tools/ab_knob.sh PBR_BLOOM_POLY 0 1 2 2>&1 | tee gpurun_out/r04_b_ab_bloom_poly.txt
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r04_b_stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-host-graph --no-shade-paths > $GRAFT_REPO_ROOT/gpurun_out/r04_b_stats.log 2>&1
cd $GRAFT_REPO_ROOT; f=$(find gpurun_out/r04_b_stats -name "*kernel_stats.csv" | head -1); head -20 $f
python tools/cfg3_ms.py r04_b
