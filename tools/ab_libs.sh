#!/bin/bash
# A/B of library builds on one box: bash tools/ab_libs.sh v0 v1 v2 ...  (tools/ab/libpbr_<tag>.so), two interleaved rounds
for round in 1 2; do for t in "$@"; do
  PBR_HIP_LIB=$PWD/tools/ab/libpbr_$t.so python3 bench.py --steps 200 --warmup 50 --no-cpu-baseline --no-host-graph --no-shade-paths --no-tail-overlap > gpurun_out/ab_$t.json 2>/dev/null && python3 -c "
import json;d=json.load(open('gpurun_out/ab_$t.json'));s=d['roofline']['stage_ms'];print('$t', d['ms_per_step'], 'shade', s['shade'], 'in-frame', s['shade(in frame)'])" || exit 1
done; done
