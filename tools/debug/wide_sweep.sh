#!/bin/bash
# kernel durations of the frame with the wide 2x-up bloom kernel, and a sweep of the histogram instance's block count
set -o pipefail
root=$(pwd); out=$root/gpurun_out; export TMPDIR=/tmp
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out/r02_k_stats -- python3 $root/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-timing > $out/r02_k_stats.log 2>&1 && cd $root &&
python3 - <<'P'
import csv,glob
f=glob.glob('gpurun_out/r02_k_stats/**/*kernel_stats.csv',recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]: print(r['Name'][:70], r['Calls'], r['AverageNs'])
P
for hb in 256 512 1024 2048 4096; do
  PBR_BLOOM_HIST_BLOCKS=$hb python3 bench.py --steps 200 --warmup 50 --no-cpu-baseline > $out/r02_k_hb$hb.json 2>/dev/null && python3 -c "
import json;d=json.load(open('gpurun_out/r02_k_hb$hb.json'));print($hb, d['ms_per_step'], d['roofline']['stage_ms']['bloom+histogram'])" || exit 1
done
