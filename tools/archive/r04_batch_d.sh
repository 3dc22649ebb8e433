# round-4 batch d: A/B of the shade's list-entry prefetch (tools/ab/libpbr_{base,prefetch}.so), prefilter after the fp32-only grouping
set -o pipefail
mkdir -p gpurun_out
bash tools/ab_libs.sh base prefetch 2>&1 | tee gpurun_out/r04_d_ab_shade_prefetch.txt
python tools/cfg3_ms.py r04_d 2>&1 | tee gpurun_out/r04_d_cfg3.txt
