# round-4 batch k: bloom prefilter with one 16-byte load per quad row: bloom tests + kernel stats of the bench frame
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "bloom or frame" > gpurun_out/r04_k_pytest.log 2>&1; rc=$?
tail -n 4 gpurun_out/r04_k_pytest.log
[ $rc -eq 0 ] || exit $rc
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r04_k_stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-host-graph --no-shade-paths --no-tail-overlap > $GRAFT_REPO_ROOT/gpurun_out/r04_k_stats.log 2>&1
cd $GRAFT_REPO_ROOT; f=$(find gpurun_out/r04_k_stats -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/r04_k_kernel_stats_4k.csv; cut -d'"' -f2,3 $f | awk -F'"' '{print substr($1,1,46), $2}' | head -9
