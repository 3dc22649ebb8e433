#!/bin/bash
# round-3 GPU batch 1: full GPU tests, A/B of the shade trims, cfg2 counters, the host graph under the profiler, counter list
set -o pipefail
out=gpurun_out; mkdir -p $out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q -s > $out/r03_c_pytest_gpu.log 2>&1; rc=$?
tail -n 8 $out/r03_c_pytest_gpu.log
[ $rc -eq 0 ] || exit $rc
grep "fp32 shade vs f64" $out/r03_c_pytest_gpu.log > $out/r03_c_f64_distributions.txt
bash tools/ab_libs.sh base trim 2>&1 | tee $out/r03_c_ab_shade.txt &&
(rocprofv3 --list-avail > $out/r03_counters_avail.txt 2>&1 || true) &&
bash tools/pmc_cfg.sh r03_cfg2 1 1920 1080 > $out/r03_cfg2_pmc.txt 2>&1; tail -n 40 $out/r03_cfg2_pmc.txt
bash tools/host_trace.sh r03_c
