#!/bin/bash
# per-launch durations of a cfg5 rank's frame on one GPU: bash tools/cfg5_tile_trace.sh <tag>  -> gpurun_out/<tag>_cfg5_tile_timeline.txt
tag=${1:-cfg5}; root=$(pwd); export TMPDIR=/tmp
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/${tag}_cfg5trace -- python3 $root/tools/cfg5_tile_trace.py > $root/gpurun_out/${tag}_cfg5trace.log 2>&1 || { tail -5 $root/gpurun_out/${tag}_cfg5trace.log; exit 1; }
cd $root && python3 tools/print_timeline.py $(find gpurun_out/${tag}_cfg5trace -name "*kernel_trace.csv" | head -1) > gpurun_out/${tag}_cfg5_tile_timeline.txt && cat gpurun_out/${tag}_cfg5_tile_timeline.txt
