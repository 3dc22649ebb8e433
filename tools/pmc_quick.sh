#!/bin/bash
# two SQ counter passes + LDS/traffic pass of one 4K frame; prints the per-kernel summary: bash tools/pmc_quick.sh <tag>
tag=${1:-pq}
root=$(pwd); out=$root/gpurun_out; export TMPDIR=/tmp
cd /tmp &&
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS --kernel-trace --output-format csv -d $out/${tag}_sq1 -- python3 $root/tools/profile_stage.py all > $out/${tag}_sq1.log 2>&1 &&
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU SQ_WAVES --kernel-trace --output-format csv -d $out/${tag}_sq2 -- python3 $root/tools/profile_stage.py all > $out/${tag}_sq2.log 2>&1 &&
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/${tag}_fetch -- python3 $root/tools/profile_stage.py all > $out/${tag}_fetch.log 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/${tag}_write -- python3 $root/tools/profile_stage.py all > $out/${tag}_write.log 2>&1 &&
cd $root && python3 tools/summarize_pmc.py $out/${tag}_pmc.json $out/${tag}_sq1 $out/${tag}_sq2 $out/${tag}_fetch $out/${tag}_write && python3 tools/print_pmc.py $out/${tag}_pmc.json
