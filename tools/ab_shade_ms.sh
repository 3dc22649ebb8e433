#!/bin/bash
# interleaved A/B of library builds on the 4K / 256-light shade alone (tools/shade_ms.py): bash tools/ab_shade_ms.sh tag0 tag1 ...
for round in 1 2; do for t in "$@"; do
  PBR_HIP_LIB=$PWD/tools/ab/libpbr_$t.so python3 tools/shade_ms.py $t 2>&1 | grep "shade isolated"
done; done
