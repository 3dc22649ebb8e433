#!/bin/bash
# interleaved A/B of library builds on the frame's bloom stages (bench.py stage timings): bash tools/ab_bloom_libs.sh tag0 tag1 ...
for round in 1 2 3; do for t in "$@"; do
  PBR_HIP_LIB=$PWD/tools/ab/libpbr_$t.so python3 bench.py --steps 200 --warmup 50 --no-cpu-baseline --no-host-graph --no-shade-paths --no-tail-overlap > gpurun_out/ab_$t.json 2>/dev/null && python3 -c "
import json;d=json.load(open('gpurun_out/ab_$t.json'));s=d['roofline']['stage_ms'];print('$t', 'frame', d['ms_per_step'], 'bloom+histogram', s['bloom+histogram'], 'bloom', s['bloom'])" || exit 1
done; done
