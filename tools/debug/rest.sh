set -o pipefail
out=gpurun_out; tag=r02_a
python bench.py --gpus 4 --steps 3 --warmup 1 --frame 3840x2048 --layout 2x2 --no-cpu-baseline --no-kernel-timing > $out/${tag}_rehearsal4_strong.json 2> $out/${tag}_rehearsal4_strong.err || { tail -n 40 $out/${tag}_rehearsal4_strong.err; exit 1; }
tail -c 1200 $out/${tag}_rehearsal4_strong.json; echo
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/valu_rate3.hip -o /tmp/valu_rate3 && timeout -k 10 300 /tmp/valu_rate3 > $out/${tag}_valu_rate3.txt && cat $out/${tag}_valu_rate3.txt
