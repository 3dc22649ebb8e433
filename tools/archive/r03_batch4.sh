#!/bin/bash
# round-3 GPU batch 4: shade parity tests for the folded attenuation polynomial, its A/B, then the full profile set with fresh stamps
set -o pipefail
tag=${1:-r03_k}; out=gpurun_out; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "shade or band or frame" > $out/${tag}_pytest_shade.log 2>&1; rc=$?
tail -n 4 $out/${tag}_pytest_shade.log
[ $rc -eq 0 ] || exit $rc
bash tools/ab_libs.sh base dist 2>&1 | tee $out/${tag}_ab_shade_dist.txt
bash tools/collect_profiles.sh $tag > $out/${tag}_collect.log 2>&1 || { tail -n 20 $out/${tag}_collect.log; exit 1; }
tail -n 4 $out/${tag}_collect.log
bash tools/pmc_shade_issue.sh ${tag}_issue > $out/${tag}_issue.txt 2>&1 || { tail -n 5 $out/${tag}_issue.txt; exit 1; }
grep "SQ_INSTS_VALU\b\|SQ_ACTIVE_INST_VALU \|TRANS" $out/${tag}_issue.txt
