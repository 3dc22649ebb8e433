#!/bin/bash
# round 6, batch a: GPU tests on the queue-scheduled shade, then before/after (static grid vs queue) size tables and timelines on ONE box
set -o pipefail
out=gpurun_out; mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -x -q -rs > $out/r06_a_pytest.log 2>&1; rc=$?
tail -n 8 $out/r06_a_pytest.log
[ $rc -eq 0 ] || exit $rc
K=direct12pbrrenderer_amd/libpbr_hip_knobs.so
for r in 1 2; do
  PBR_HIP_LIB=$K PBR_SHADE_SCHED=grid timeout -k 10 300 python tools/shade_tile_ms.py grid$r >> $out/r06_a_sizes.jsonl 2>> $out/r06_a_sizes.err || exit 1
  PBR_HIP_LIB=$K timeout -k 10 300 python tools/shade_tile_ms.py queue$r >> $out/r06_a_sizes.jsonl 2>> $out/r06_a_sizes.err || exit 1
done
cat $out/r06_a_sizes.jsonl
T=tools/ab/libpbr_timing.so
PBR_HIP_LIB=$T PBR_SHADE_SCHED=grid timeout -k 10 300 python tools/shade_timeline.py grid > $out/r06_a_timeline.jsonl 2> $out/r06_a_timeline.err || { tail -5 $out/r06_a_timeline.err; exit 1; }
PBR_HIP_LIB=$T timeout -k 10 300 python tools/shade_timeline.py queue >> $out/r06_a_timeline.jsonl 2>> $out/r06_a_timeline.err || { tail -5 $out/r06_a_timeline.err; exit 1; }
cat $out/r06_a_timeline.jsonl
