#!/bin/bash
# Counter passes of the GGX prefilter (separate rocprofv3 --pmc runs, kernel trace only): bash tools/pmc_prefilter.sh <tag> [half|f32]
# -> gpurun_out/<tag>_pmc.json (tools/summarize_pmc.py: mean per kernel over its last 5 dispatches)
tag=${1:-pf}; src=${2:-half}
root=$(pwd); out=$root/gpurun_out; export TMPDIR=/tmp
export PBR_PROFILE_PIXELS=$((6*512*512))
cd /tmp || exit 1
dirs=""
i=0
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TA_TA_BUSY_sum TA_BUSY_avr" "TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" \
         "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS" \
         "GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVES SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" \
         "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS"; do
  i=$((i+1)); d=$out/${tag}_p$i
  if rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -- python3 $root/tools/profile_prefilter.py $src > $d.log 2>&1; then dirs="$dirs $d"; else echo "pass '$c' failed (counter not available?)"; tail -n 3 $d.log; fi
done
cd $root && python3 tools/summarize_pmc.py $out/${tag}_pmc.json $dirs && python3 tools/print_pmc.py $out/${tag}_pmc.json
