#!/bin/bash
# GPU-box check of one state of the tree (run through gpurun): GPU parity tests, the bench line (with its host_graph block),
# the 2-rank rehearsal of bench.py on the one GPU (weak-scaling headline + cfg5 sub-record).  usage: bash tools/gpu_check.sh <tag> [quick]
set -o pipefail
tag=${1:-rXX}
out=gpurun_out
mkdir -p $out
if [ "$2" = "quick" ]; then
  timeout -k 10 900 python -m pytest tests/test_host_graph.py tests/test_gpu_comm.py -m gpu -x -q -s -rs > $out/${tag}_pytest_gpu.log 2>&1; rc=$?
else
  timeout -k 10 1100 python -m pytest tests -m gpu -x -q -s -rs > $out/${tag}_pytest_gpu.log 2>&1; rc=$?
fi
tail -n 15 $out/${tag}_pytest_gpu.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python bench.py --steps 50 --warmup 5 > $out/${tag}_bench_4k.json 2> $out/${tag}_bench.err || { tail -n 30 $out/${tag}_bench.err; exit 1; }
tail -c 3500 $out/${tag}_bench_4k.json; echo
timeout -k 10 600 python bench.py --gpus 2 --steps 5 --warmup 1 --no-cpu-baseline > $out/${tag}_rehearsal2.json 2> $out/${tag}_rehearsal2.err || { tail -n 40 $out/${tag}_rehearsal2.err; exit 1; }
tail -c 2500 $out/${tag}_rehearsal2.json; echo
