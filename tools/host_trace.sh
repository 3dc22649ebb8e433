#!/bin/bash
# The drop-in path under the profiler: rocprofv3 kernel trace + stats of tools/host_frame_ms.py (the C++ pass graph rendering
# the 4K / 256-light frame dispatch by dispatch and fused, fence per frame and 3 frames in flight).
#   bash tools/host_trace.sh <tag>  ->  gpurun_out/<tag>_host_frame_ms.txt (plain run), gpurun_out/<tag>_host_stats/ (trace)
tag=${1:-host}; root=$(pwd); out=$root/gpurun_out; export TMPDIR=/tmp
python3 tools/host_frame_ms.py 100 2>/dev/null | tee $out/${tag}_host_frame_ms.txt &&
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_host_stats -- python3 $root/tools/host_frame_ms.py 30 > $out/${tag}_host_stats.log 2>&1 &&
cd $root && f=$(find $out/${tag}_host_stats -name "*kernel_stats.csv" | head -1) && cp $f $out/${tag}_host_kernel_stats.csv && head -n 30 $out/${tag}_host_kernel_stats.csv | cut -c1-150
