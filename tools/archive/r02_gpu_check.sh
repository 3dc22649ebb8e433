#!/bin/bash
# GPU-box check of one state of the tree (run through gpurun): GPU parity tests, the bench line, the multi-rank
# rehearsals of bench.py on the one GPU, the VALU issue-rate micro-benchmark.  usage: bash tools/r02_gpu_check.sh <tag>
set -o pipefail
tag=${1:-r02}
out=gpurun_out
mkdir -p $out
python -m pytest tests -m gpu -x -q -s > $out/${tag}_pytest_gpu.log 2>&1; rc=$?
tail -n 15 $out/${tag}_pytest_gpu.log
[ $rc -eq 0 ] || exit $rc
python bench.py --steps 50 --warmup 5 > $out/${tag}_bench_4k.json 2> $out/${tag}_bench.err || { tail -n 30 $out/${tag}_bench.err; exit 1; }
tail -c 2500 $out/${tag}_bench_4k.json; echo
for mode in halo apron; do
  python bench.py --gpus 2 --steps 5 --warmup 1 --mode $mode --no-cpu-baseline > $out/${tag}_rehearsal2_$mode.json 2> $out/${tag}_rehearsal2_$mode.err || { tail -n 40 $out/${tag}_rehearsal2_$mode.err; exit 1; }
  tail -c 1500 $out/${tag}_rehearsal2_$mode.json; echo
done
python bench.py --gpus 4 --steps 3 --warmup 1 --frame 3840x2048 --layout 2x2 --no-cpu-baseline --no-kernel-timing > $out/${tag}_rehearsal4_strong.json 2> $out/${tag}_rehearsal4_strong.err || { tail -n 40 $out/${tag}_rehearsal4_strong.err; exit 1; }
tail -c 1200 $out/${tag}_rehearsal4_strong.json; echo
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/valu_rate3.hip -o /tmp/valu_rate3 && timeout -k 10 300 /tmp/valu_rate3 > $out/${tag}_valu_rate3.txt && cat $out/${tag}_valu_rate3.txt
