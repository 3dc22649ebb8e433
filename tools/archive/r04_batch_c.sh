# round-4 batch c: full GPU tests of the tree (poly bloom, N = 1 throughput mode, scene lights), prefilter XCD A/B + counters, bench line
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q -rs > gpurun_out/r04_c_pytest_gpu.log 2>&1; rc=$?
tail -n 8 gpurun_out/r04_c_pytest_gpu.log
[ $rc -eq 0 ] || exit $rc
export PBR_HIP_LIB=$PWD/direct12pbrrenderer_amd/libpbr_hip_knobs.so
for r in 1 2; do for v in 1 8; do PBR_PREFILTER_XCD=$v python tools/cfg3_ms.py "xcd_groups=$v"; done; done 2>&1 | tee gpurun_out/r04_c_prefilter_xcd_ab.txt
unset PBR_HIP_LIB
bash tools/pmc_prefilter.sh r04_c_pf_f32 f32 > gpurun_out/r04_c_pmc_prefilter_f32.txt 2>&1; tail -n 12 gpurun_out/r04_c_pmc_prefilter_f32.txt
timeout -k 10 600 python bench.py --steps 50 --warmup 5 > gpurun_out/r04_c_bench_4k.json 2> gpurun_out/r04_c_bench.err || { tail -n 30 gpurun_out/r04_c_bench.err; exit 1; }
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04_c_bench_4k.json'))
print(d['value'], d['ms_per_step'], json.dumps(d['config'].get('throughput_mode')), json.dumps(d['host_graph']))
PY
