#!/bin/bash
# round 6, batch h: static grid with a long block's rows in `parts` pieces from different bands of the frame: tests, size table for parts 1 / 2 / 4, timeline
set -o pipefail
out=gpurun_out; mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -x -q -rs > $out/r06_h_pytest.log 2>&1; rc=$?
tail -n 5 $out/r06_h_pytest.log
[ $rc -eq 0 ] || exit $rc
K=direct12pbrrenderer_amd/libpbr_hip_knobs.so
for r in 1 2; do for parts in 1 2 4; do
  PBR_HIP_LIB=$K PBR_SHADE_PARTS=$parts timeout -k 10 300 python tools/shade_tile_ms.py parts${parts}_$r >> $out/r06_h_sizes.jsonl 2>> $out/r06_h_sizes.err || exit 1
done; done
cat $out/r06_h_sizes.jsonl
T=tools/ab/libpbr_timing.so
for parts in 1 2 4; do
PBR_HIP_LIB=$T PBR_SHADE_PARTS=$parts timeout -k 10 300 python tools/shade_timeline.py parts$parts >> $out/r06_h_timeline.jsonl 2>> $out/r06_h_timeline.err || { tail -5 $out/r06_h_timeline.err; exit 1; }
done
cat $out/r06_h_timeline.jsonl
