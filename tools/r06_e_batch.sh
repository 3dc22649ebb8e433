#!/bin/bash
set -o pipefail
out=gpurun_out; mkdir -p $out
K=direct12pbrrenderer_amd/libpbr_hip_knobs.so
{ echo "== grid"; PBR_HIP_LIB=$K PBR_SHADE_SCHED=grid timeout -k 10 200 python tools/debug/f32_tile_repro.py /tmp/g || exit 1
  echo "== queue"; PBR_HIP_LIB=$K timeout -k 10 200 python tools/debug/f32_tile_repro.py /tmp/q || exit 1
  for n in f32_0 f32_1080 f32_whole f16_0 f16_whole; do echo "-- grid vs queue $n"; python tools/debug/sched_cmp.py /tmp/g_$n.npy /tmp/q_$n.npy; done; } > $out/r06_e_repro.txt 2>&1
cat $out/r06_e_repro.txt
