#!/bin/bash
# round-3 final check of one state of the tree: all GPU tests, the bench line, the 2-rank rehearsal, every BASELINE config with
# its CPU leg, kernel stats of the bench under rocprofv3.  usage: bash tools/r03_final.sh <tag>
set -o pipefail
tag=${1:-r03_h}; out=gpurun_out; root=$(pwd); export TMPDIR=/tmp
bash tools/r03_gpu_check.sh $tag || exit 1
timeout -k 10 900 python tools/bench_configs.py > $out/${tag}_bench_configs.jsonl 2> $out/${tag}_bench_configs.err || { tail -n 20 $out/${tag}_bench_configs.err; exit 1; }
grep '"cfg3"\|"cfg2"' $out/${tag}_bench_configs.jsonl | cut -c1-330
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/${tag}_stats -- python3 $root/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-host-graph > $root/$out/${tag}_stats.log 2>&1 && cd $root &&
cp $(find $out/${tag}_stats -name "*kernel_stats.csv" | head -1) $out/${tag}_kernel_stats_4k.csv && cut -d, -f1-4 $out/${tag}_kernel_stats_4k.csv | cut -c1-120 | head -16
