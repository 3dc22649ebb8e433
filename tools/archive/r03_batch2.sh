#!/bin/bash
# round-3 GPU batch 2: IBL tests for the new SH9 pair, every BASELINE config with its CPU leg, the full profile set of the
# cfg4 workload (bench line, kernel stats, PMC passes with fresh source stamps), the fp64 parity report
set -o pipefail
tag=${1:-r03_d}
out=gpurun_out; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "sh9 or lut or prefilter" > $out/${tag}_pytest_ibl.log 2>&1; rc=$?
tail -n 4 $out/${tag}_pytest_ibl.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 900 python tools/bench_configs.py > $out/${tag}_bench_configs.jsonl 2> $out/${tag}_bench_configs.err || { tail -n 20 $out/${tag}_bench_configs.err; exit 1; }
cut -c1-420 $out/${tag}_bench_configs.jsonl
bash tools/collect_profiles.sh $tag || exit 1
timeout -k 10 600 python tools/f64_parity.py $out/${tag}_f64_parity.json > $out/${tag}_f64_parity.log 2>&1 || { tail -n 20 $out/${tag}_f64_parity.log; exit 1; }
echo f64 parity done
