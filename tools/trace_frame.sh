#!/bin/bash
# per-launch timeline of one 4K frame (rocprofv3 kernel trace of tools/profile_stage.py all): bash tools/trace_frame.sh <tag>
tag=${1:-trace}
root=$(pwd)
export TMPDIR=/tmp
cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $root/gpurun_out/${tag} -- python3 $root/tools/profile_stage.py all > $root/gpurun_out/${tag}.log 2>&1 && cd $root && python3 tools/print_timeline.py $(find gpurun_out/${tag} -name "*kernel_trace.csv" | head -1)
