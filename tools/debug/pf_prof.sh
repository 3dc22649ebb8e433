cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pf_prof -- python3 $GRAFT_REPO_ROOT/tools/cfg3_ms.py fast > $GRAFT_REPO_ROOT/gpurun_out/pf_prof.log 2>&1
cd $GRAFT_REPO_ROOT && f=$(find gpurun_out/pf_prof -name "*kernel_stats.csv" | head -1) && head -12 $f
