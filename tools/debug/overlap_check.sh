set -o pipefail
python -m pytest tests -m gpu -x -q 2>&1 | tail -4 || exit 1
python tools/bench_configs.py > gpurun_out/r02_g_bench_configs.log 2>&1; grep "cfg5\|cfg4\"" gpurun_out/r02_g_bench_configs.log | cut -c1-330
python bench.py --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['stage_ms'])"
