#!/bin/bash
# round 6, batch f: DMA probe, GPU shade tests on the queue kernel v2 (LDS-DMA prefetch of the next item's lists), size table vs grid, timeline
set -o pipefail
out=gpurun_out; mkdir -p $out

K0=direct12pbrrenderer_amd/libpbr_hip_knobs.so
{ PBR_HIP_LIB=$K0 timeout -k 10 200 python tools/debug/f32_tile_repro.py /tmp/q | grep differs; } && timeout -k 10 900 python -m pytest tests -m gpu -x -q -rs > $out/r06_f_pytest.log 2>&1; rc=$?
tail -n 8 $out/r06_f_pytest.log
[ $rc -eq 0 ] || exit $rc
K=direct12pbrrenderer_amd/libpbr_hip_knobs.so
for r in 1 2; do
  PBR_HIP_LIB=$K PBR_SHADE_SCHED=grid timeout -k 10 300 python tools/shade_tile_ms.py grid$r >> $out/r06_f_sizes.jsonl 2>> $out/r06_f_sizes.err || exit 1
  PBR_HIP_LIB=$K timeout -k 10 300 python tools/shade_tile_ms.py queue$r >> $out/r06_f_sizes.jsonl 2>> $out/r06_f_sizes.err || exit 1
done
cat $out/r06_f_sizes.jsonl
T=tools/ab/libpbr_timing.so
PBR_HIP_LIB=$T timeout -k 10 300 python tools/shade_timeline.py queue > $out/r06_f_timeline.jsonl 2> $out/r06_f_timeline.err || { tail -5 $out/r06_f_timeline.err; exit 1; }
cat $out/r06_f_timeline.jsonl
for rows in 1 2 4 8; do
    PBR_HIP_LIB=$K PBR_SHADE_ROWS_BIG=$rows timeout -k 10 300 python tools/shade_tile_ms.py queue_rows$rows 1920x1080 1928x2168@7680x4320 3840x2160 >> $out/r06_f_rows.jsonl 2>> $out/r06_f_rows.err || exit 1
done
grep -v fit $out/r06_f_rows.jsonl
