#!/bin/bash
# round 6, batch g: rows-per-item sweep of the STATIC grid (hardware dispatcher as the queue) on one box
set -o pipefail
out=gpurun_out; mkdir -p $out
K=direct12pbrrenderer_amd/libpbr_hip_knobs.so
for rows in 8 6 4 3 2 1; do
    PBR_HIP_LIB=$K PBR_SHADE_SCHED=grid PBR_SHADE_ROWS_BIG=$rows timeout -k 10 300 python tools/shade_tile_ms.py grid_rows$rows 1440x960 1920x1080 1928x2168@7680x4320 3840x2160 >> $out/r06_g_rows.jsonl 2>> $out/r06_g_rows.err || exit 1
done
grep -v fit $out/r06_g_rows.jsonl
