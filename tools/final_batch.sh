#!/bin/bash
# The round's evidence batch on the final tree, under one tag (run through gpurun): bash tools/final_batch.sh <tag>
#   GPU tests + bench line + 2-rank rehearsal (gpu_check.sh), every BASELINE config with its CPU leg (bench_configs.py),
#   rocprofv3 kernel stats + the PMC passes of the frame (collect_profiles.sh), the shade's issue counters, the C++ pass graph under
#   the profiler (host_trace.sh), the fp64 parity report (f64_parity.py).  Steps are chained: a failed GPU step stops the batch.
set -o pipefail
tag=${1:-rXX}; out=gpurun_out; root=$(pwd); export TMPDIR=/tmp
bash tools/gpu_check.sh $tag || exit 1
grep "fp32 shade vs f64\|\[parity\]" $out/${tag}_pytest_gpu.log > $out/${tag}_f64_distributions.txt
timeout -k 10 900 python tools/bench_configs.py > $out/${tag}_bench_configs.jsonl 2> $out/${tag}_bench_configs.err || { tail -n 20 $out/${tag}_bench_configs.err; exit 1; }
grep '"cfg3"\|"cfg2"\|"cfg1"' $out/${tag}_bench_configs.jsonl | cut -c1-260
bash tools/collect_profiles.sh $tag > $out/${tag}_collect.log 2>&1 || { tail -n 20 $out/${tag}_collect.log; exit 1; }
tail -n 3 $out/${tag}_collect.log
bash tools/pmc_shade_issue.sh ${tag}_issue > $out/${tag}_issue.txt 2>&1 || { tail -n 5 $out/${tag}_issue.txt; exit 1; }
bash tools/host_trace.sh $tag
timeout -k 10 600 python tools/f64_parity.py $out/${tag}_f64_parity.json > $out/${tag}_f64_parity.log 2>&1 || { tail -n 20 $out/${tag}_f64_parity.log; exit 1; }
echo final batch done
