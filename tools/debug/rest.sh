set -o pipefail
out=gpurun_out; tag=r02_b
python -m pytest tests -m gpu -x -q -s > $out/${tag}_pytest_gpu.log 2>&1; rc=$?
tail -n 5 $out/${tag}_pytest_gpu.log
[ $rc -eq 0 ] || exit $rc
python bench.py --steps 50 --warmup 5 --no-cpu-baseline > $out/${tag}_bench_4k.json 2> $out/${tag}_bench.err || { tail -n 30 $out/${tag}_bench.err; exit 1; }
python -c "
import json;d=json.load(open('$out/${tag}_bench_4k.json'));print(d['value'],d['ms_per_step'],d['roofline']['stage_ms'])"
python tools/shade_coherence.py > $out/${tag}_shade_coherence.txt 2>&1; cat $out/${tag}_shade_coherence.txt
