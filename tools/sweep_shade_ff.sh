#!/bin/bash
# sweep of the two-zone block schedule for the front-facing shade (knobs build; one process per setting)
for cfg in "0.92 1" "1.0 1" "0.96 1" "0.96 2" "0.92 2" "0.92 4" "0.85 4" "0.96 4" "0.8 8"; do
  set -- $cfg
  PBR_HIP_LIB=$PWD/direct12pbrrenderer_amd/libpbr_hip_knobs.so PBR_SHADE_FF=${FF:-1} PBR_SHADE_BIGFRAC=$1 PBR_SHADE_ROWS_SMALL=$2 python tools/shade_ms.py "ff=${FF:-1} bigfrac=$1 rows_small=$2" 2>&1 | grep "shade isolated"
done
