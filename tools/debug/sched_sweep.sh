#!/bin/bash
# the shade's two-zone block schedule, priced with the whole frame (bench.py), two interleaved rounds
for round in 1 2; do for cfg in "0.85 2" "1.0 2" "0.9 2" "0.8 2" "0.7 2" "0.85 1" "0.85 4" "0.9 1" "0.95 1"; do
  set -- $cfg
  PBR_SHADE_BIGFRAC=$1 PBR_SHADE_ROWS_SMALL=$2 python3 bench.py --steps 200 --warmup 50 --no-cpu-baseline > gpurun_out/sched.json 2>/dev/null && python3 -c "
import json;d=json.load(open('gpurun_out/sched.json'));s=d['roofline']['stage_ms'];print('bigfrac $1 rows_small $2', d['ms_per_step'], 'in-frame', s['shade(in frame)'])" || exit 1
done; done
