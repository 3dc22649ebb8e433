#!/bin/bash
# round-3 GPU batch 3: IBL tests with the half-footprint prefilter, its A/B against the fp32 kernel (knobs build), counters of
# the new kernel, the shade's issue-side counters (fresh stamp)
set -o pipefail
tag=${1:-r03_e}
out=gpurun_out; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "sh9 or lut or prefilter or band" > $out/${tag}_pytest_ibl.log 2>&1; rc=$?
tail -n 12 $out/${tag}_pytest_ibl.log
[ $rc -eq 0 ] || exit $rc
K=$PWD/direct12pbrrenderer_amd/libpbr_hip_knobs.so
for i in 1 2; do
  PBR_HIP_LIB=$K PBR_PREFILTER_F32=1 python tools/cfg3_ms.py "fp32 padded chain (round 2)" 2>/dev/null
  PBR_HIP_LIB=$K python tools/cfg3_ms.py "half footprint chain" 2>/dev/null
done | tee $out/${tag}_prefilter_ab.txt
bash tools/pmc_shade_issue.sh ${tag}_issue > $out/${tag}_issue.txt 2>&1 || { tail -n 5 $out/${tag}_issue.txt; exit 1; }
tail -n 12 $out/${tag}_issue.txt
bash tools/pmc_cfg.sh ${tag}_pf 1 640 360 > $out/${tag}_pf_pmc.txt 2>&1; grep "k_prefilter\|k_sh9\|k_cube_foot" $out/${tag}_pf_pmc.txt
