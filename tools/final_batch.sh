#!/bin/bash
# The round's evidence batch on the final tree, under one tag, in two halves (a gpurun call is limited to 20 minutes):
#   bash tools/final_batch.sh <tag> a    GPU tests + bench line + 2-rank rehearsal (gpu_check.sh), every BASELINE config with its CPU leg
#                                        (bench_configs.py), the fp64 parity report (f64_parity.py)
#   bash tools/final_batch.sh <tag> b    rocprofv3 kernel stats + the PMC passes of the frame (collect_profiles.sh), the shade's issue counters,
#                                        the cfg2 counter passes (pmc_cfg.sh), the C++ pass graph under the profiler (host_trace.sh), a cfg5 rank's timeline
# Steps are chained: a failed GPU step stops the batch.
set -o pipefail
tag=${1:-rXX}; half=${2:-a}; out=gpurun_out; root=$(pwd); export TMPDIR=/tmp
if [ "$half" = "a" ]; then
  bash tools/gpu_check.sh $tag || exit 1
  grep "fp32 shade vs f64\|\[parity\]" $out/${tag}_pytest_gpu.log > $out/${tag}_f64_distributions.txt
  timeout -k 10 900 python tools/bench_configs.py > $out/${tag}_bench_configs.jsonl 2> $out/${tag}_bench_configs.err || { tail -n 20 $out/${tag}_bench_configs.err; exit 1; }
  grep '"cfg3"\|"cfg2"\|"cfg1"' $out/${tag}_bench_configs.jsonl | cut -c1-260
  timeout -k 10 600 python tools/f64_parity.py $out/${tag}_f64_parity.json > $out/${tag}_f64_parity.log 2>&1 || { tail -n 20 $out/${tag}_f64_parity.log; exit 1; }
else
  bash tools/collect_profiles.sh $tag > $out/${tag}_collect.log 2>&1 || { tail -n 20 $out/${tag}_collect.log; exit 1; }
  tail -n 3 $out/${tag}_collect.log
  bash tools/pmc_shade_issue.sh ${tag}_issue > $out/${tag}_issue.txt 2>&1 || { tail -n 5 $out/${tag}_issue.txt; exit 1; }
  bash tools/pmc_cfg.sh ${tag}_cfg2 1 1920 1080 > $out/${tag}_cfg2_pmc.txt 2>&1 || { tail -n 20 $out/${tag}_cfg2_pmc.txt; exit 1; }
  bash tools/host_trace.sh $tag
  bash tools/cfg5_tile_trace.sh $tag | tail -n 14
fi
echo final batch $half done
