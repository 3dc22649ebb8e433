#!/bin/bash
# round 6, batch m: the reference operating point test, the bench line with its `configs` block, cfg2 counter passes
set -o pipefail
out=gpurun_out; mkdir -p $out; root=$(pwd); export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_host_graph.py -m gpu -x -q -rs -k "scene_lights or fused" > $out/r06_m_pytest.log 2>&1; rc=$?
tail -n 4 $out/r06_m_pytest.log
[ $rc -eq 0 ] || exit $rc
bash tools/pmc_cfg.sh r06_m_cfg2 1 1920 1080 > $out/r06_m_cfg2_pmc.txt 2>&1 || { tail -20 $out/r06_m_cfg2_pmc.txt; exit 1; }
cp $out/r06_m_cfg2_pmc.json profiles/pmc_cfg2_latest.json
tail -n 12 $out/r06_m_cfg2_pmc.txt
timeout -k 10 900 python bench.py --steps 50 --warmup 5 > $out/r06_m_bench_4k.json 2> $out/r06_m_bench.err || { tail -n 30 $out/r06_m_bench.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_m_bench_4k.json').read().strip().split('\n')[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'])
print(json.dumps(d.get('configs'), indent=1)[:6000])
PY
