# round-4 batch g: fused average + tone-map (A/B in one process), bloom prefilter with pipelined quad loads (kernel stats), tests of both
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_comm.py tests/test_gpu_bench.py -m gpu -x -q -k "average or bloom or tonemap or frame or tail or bench" -rs > gpurun_out/r04_g_pytest.log 2>&1; rc=$?
tail -n 6 gpurun_out/r04_g_pytest.log
[ $rc -eq 0 ] || exit $rc
python tools/frame_ab.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_g_frame_ab.txt
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r04_g_stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-host-graph --no-shade-paths --no-tail-overlap > $GRAFT_REPO_ROOT/gpurun_out/r04_g_stats.log 2>&1
cd $GRAFT_REPO_ROOT; f=$(find gpurun_out/r04_g_stats -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/r04_g_kernel_stats_4k.csv; head -12 $f | cut -c1-160
