#!/bin/bash
# sweep of the shade kernel's two-zone schedule (one process per setting; the knobs are read once)
for cfg in "1.0 2" "0.9 2" "0.8 2" "0.7 2" "0.6 2" "0.5 2" "0.8 4" "0.6 4" "0.0 4" "0.0 2" "0.8 1"; do
  set -- $cfg
  PBR_HIP_LIB=$PWD/direct12pbrrenderer_amd/libpbr_hip_knobs.so PBR_SHADE_BIGFRAC=$1 PBR_SHADE_ROWS_SMALL=$2 python tools/shade_ms.py "bigfrac=$1 rows_small=$2" 2>&1 | grep -v amdgpu.ids
done
