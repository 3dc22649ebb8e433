#!/bin/bash
# Round 4, batch p: tone-map and bloom prefilter walking the HDR buffer in the REVERSE of the order their producer wrote it
# (PBR_REVERSE bit 0: k_tonemap, bit 1: k_bloom_prefilter_2x), interleaved on one box.
export PBR_HIP_LIB=$PWD/direct12pbrrenderer_amd/libpbr_hip_knobs.so
for r in 1 2; do for v in 0 1 2 3; do
  env PBR_REVERSE=$v python3 bench.py --steps 200 --warmup 50 --no-cpu-baseline --no-host-graph --no-shade-paths --no-tail-overlap > gpurun_out/ab_rev_$v.json 2>/dev/null && python3 -c "
import json;d=json.load(open('gpurun_out/ab_rev_$v.json'));s=d['roofline']['stage_ms'];print('PBR_REVERSE=$v', 'frame', d['ms_per_step'], 'shade in-frame', s['shade(in frame)'], 'bloom+histogram', s['bloom+histogram'], 'tonemap', s['tonemap'])" || exit 1
done; done
