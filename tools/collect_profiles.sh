#!/bin/bash
# Collects the evidence profiles/ holds for one state of the code, on the GPU box (run through gpurun):
#   bash tools/collect_profiles.sh <tag>        e.g. r01_f
# 1. bench.py line                      -> gpurun_out/<tag>_bench_4k.json
# 2. rocprofv3 --kernel-trace --stats   -> gpurun_out/<tag>_stats/ (kernel_stats.csv)
# 3. PMC passes, one counter group per run (never combined with other trace domains)
#    -> gpurun_out/<tag>_pmc_*/ ; summarised into gpurun_out/<tag>_pmc_hbm_traffic_4k.json / <tag>_pmc_sq_4k.json
# Steps are chained with && : a failed GPU step stops the script.
set -o pipefail
tag=${1:-rXX}
root=$(pwd)
out=$root/gpurun_out
mkdir -p $out
export TMPDIR=/tmp
python3 bench.py --steps 50 --warmup 5 > $out/${tag}_bench_4k.json 2> $out/${tag}_bench.err && tail -c 600 $out/${tag}_bench_4k.json && echo &&
cd /tmp &&
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats -- python3 $root/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-host-graph --no-shade-paths --no-tail-overlap --no-configs > $out/${tag}_stats.log 2>&1 &&
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  d=$out/${tag}_pmc_$(echo $c | cut -d' ' -f1)
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -- python3 $root/tools/profile_stage.py all > $d.log 2>&1 || exit 1
done &&
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS --kernel-trace --output-format csv -d $out/${tag}_pmc_sq1 -- python3 $root/tools/profile_stage.py all > $out/${tag}_pmc_sq1.log 2>&1 &&
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU SQ_WAVES --kernel-trace --output-format csv -d $out/${tag}_pmc_sq2 -- python3 $root/tools/profile_stage.py all > $out/${tag}_pmc_sq2.log 2>&1 &&
cd $root &&
python3 tools/summarize_pmc.py $out/${tag}_pmc_hbm_traffic_4k.json $out/${tag}_pmc_FETCH_SIZE $out/${tag}_pmc_WRITE_SIZE $out/${tag}_pmc_TCC_HIT_sum &&
python3 tools/summarize_pmc.py $out/${tag}_pmc_sq_4k.json $out/${tag}_pmc_sq1 $out/${tag}_pmc_sq2 &&
find $out/${tag}_stats -name "*kernel_stats.csv" | head -3
