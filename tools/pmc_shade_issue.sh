#!/bin/bash
# issue-side counters of the shade kernel (what keeps the VALU from issuing): bash tools/pmc_shade_issue.sh <tag>
tag=${1:-psi}
root=$(pwd); out=$root/gpurun_out; export TMPDIR=/tmp
cd /tmp &&
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $out/${tag}_a -- python3 $root/tools/profile_stage.py shade > $out/${tag}_a.log 2>&1 &&
rocprofv3 --pmc SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_LDS_LOAD SQ_LDS_ADDR_CONFLICT --kernel-trace --output-format csv -d $out/${tag}_b -- python3 $root/tools/profile_stage.py shade > $out/${tag}_b.log 2>&1 &&
rocprofv3 --pmc SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_IFETCH SQ_IFETCH_LEVEL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_SALU SQ_INST_CYCLES_SALU --kernel-trace --output-format csv -d $out/${tag}_c -- python3 $root/tools/profile_stage.py shade > $out/${tag}_c.log 2>&1 &&
cd $root && python3 tools/summarize_pmc.py $out/${tag}.json $out/${tag}_a $out/${tag}_b $out/${tag}_c && python3 - <<PY
import json
d=json.load(open("$out/${tag}.json"))
for k,v in d.items():
    if "shade" in k:
        for a,b in sorted(v.items()): print(f"{a:32s} {b:16.0f}")
PY
