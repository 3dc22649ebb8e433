#!/bin/bash
set -o pipefail
out=gpurun_out; mkdir -p $out
K=direct12pbrrenderer_amd/libpbr_hip_knobs.so
for cfg in "3840 1080 3840 2160" "3840 2160 3840 2160" "1920 1080 1920 1080"; do
  PBR_HIP_LIB=$K PBR_SHADE_SCHED=grid timeout -k 10 200 python tools/debug/sched_dump.py /tmp/g.npy $cfg || exit 1
  PBR_HIP_LIB=$K timeout -k 10 200 python tools/debug/sched_dump.py /tmp/q.npy $cfg || exit 1
  echo "== $cfg"; python tools/debug/sched_cmp.py /tmp/g.npy /tmp/q.npy
done > $out/r06_d_diff.txt 2>&1
cat $out/r06_d_diff.txt
