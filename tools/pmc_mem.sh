#!/bin/bash
# memory-pipeline counters (TA / TCP / TCC) of one command: bash tools/pmc_mem.sh <tag> <python script> [args]
tag=$1; shift
root=$(pwd); out=$root/gpurun_out; export TMPDIR=/tmp
cd /tmp &&
rocprofv3 --pmc TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/${tag}_m1 -- python3 $root/$@ > $out/${tag}_m1.log 2>&1 &&
rocprofv3 --pmc TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_DATA_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $out/${tag}_m2 -- python3 $root/$@ > $out/${tag}_m2.log 2>&1 &&
cd $root && python3 tools/summarize_pmc.py $out/${tag}_mem.json $out/${tag}_m1 $out/${tag}_m2 && python3 - <<PY
import json
d=json.load(open("$out/${tag}_mem.json"))
for k,v in d.items():
    if not isinstance(v,dict) or v.get("GRBM_GUI_ACTIVE",0)<2e5: continue
    print(k[:44].ljust(44), {a:(round(b) if isinstance(b,float) else b) for a,b in v.items() if a!="dispatches_seen"})
PY
