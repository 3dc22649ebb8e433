#!/bin/bash
# Interleaved A/B of one switch of the knobs build on one box: bash tools/ab_knob.sh <VAR> <value A> <value B> [rounds]
# (bench.py line per run: frame ms, shade in frame, bloom+histogram stage; boxes differ by 3-5 %, so both values run on the same box)
var=$1; a=$2; b=$3; rounds=${4:-2}
export PBR_HIP_LIB=$PWD/direct12pbrrenderer_amd/libpbr_hip_knobs.so
for r in $(seq $rounds); do for v in "$a" "$b"; do
  env $var=$v python3 bench.py --steps 200 --warmup 50 --no-cpu-baseline --no-host-graph --no-shade-paths --no-tail-overlap > gpurun_out/ab_${var}_$v.json 2>/dev/null && python3 -c "
import json;d=json.load(open('gpurun_out/ab_${var}_$v.json'));s=d['roofline']['stage_ms'];print('$var=$v', 'frame', d['ms_per_step'], 'shade in-frame', s['shade(in frame)'], 'bloom+histogram', s['bloom+histogram'], 'bloom', s['bloom'])" || exit 1
done; done
