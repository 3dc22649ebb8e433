#!/bin/bash
# round 6, batch b: rows-per-item sweep of both schedules (knobs build) on one box
set -o pipefail
out=gpurun_out; mkdir -p $out
K=direct12pbrrenderer_amd/libpbr_hip_knobs.so
for rows in 2 3 4 6 8; do
  for sched in grid queue; do
    PBR_HIP_LIB=$K PBR_SHADE_SCHED=$sched PBR_SHADE_ROWS_BIG=$rows timeout -k 10 300 python tools/shade_tile_ms.py ${sched}_rows$rows 1920x1080 1928x2168@7680x4320 3840x2160 >> $out/r06_b_rows.jsonl 2>> $out/r06_b_rows.err || exit 1
  done
done
PBR_HIP_LIB=$K PBR_SHADE_SCHED=queue timeout -k 10 300 python tools/shade_tile_ms.py queue_auto 1920x1080 1928x2168@7680x4320 3840x2160 >> $out/r06_b_rows.jsonl 2>> $out/r06_b_rows.err || exit 1
grep -v fit $out/r06_b_rows.jsonl
