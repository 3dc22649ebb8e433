python -m pytest tests -m gpu -x -q 2>&1 | tail -3 || exit 1
bash tools/debug/collect.sh || exit 1
python tools/bench_configs.py > gpurun_out/r02_i_bench_configs.log 2>&1; tail -n 16 gpurun_out/r02_i_bench_configs.log | cut -c1-260
