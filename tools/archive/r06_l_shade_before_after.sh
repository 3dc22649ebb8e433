#!/bin/bash
# round 6, batch l: the shade's launch before (round 5: 8-row blocks at every size, a thread per list entry) and after (rows by target size,
# four list entries per thread) on ONE box, interleaved, + GPU tests on the product library
set -o pipefail
out=gpurun_out; mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -x -q -rs > $out/r06_l_pytest.log 2>&1; rc=$?
tail -n 4 $out/r06_l_pytest.log
[ $rc -eq 0 ] || exit $rc
for r in 1 2 3; do
  PBR_HIP_LIB=tools/ab/libpbr_r5stage.so PBR_SHADE_ROWS_BIG=8 timeout -k 10 300 python tools/shade_tile_ms.py before_$r 1280x720 1440x960 1920x1080 1928x2168@7680x4320 2720x3056@10880x6112 3840x2160 7680x4320 >> $out/r06_l_sizes.jsonl 2>> $out/r06_l_sizes.err || exit 1
  PBR_HIP_LIB=tools/ab/libpbr_new.so timeout -k 10 300 python tools/shade_tile_ms.py after_$r 1280x720 1440x960 1920x1080 1928x2168@7680x4320 2720x3056@10880x6112 3840x2160 7680x4320 >> $out/r06_l_sizes.jsonl 2>> $out/r06_l_sizes.err || exit 1
done
python - <<'PY'
import json,collections
rows=[json.loads(l) for l in open('gpurun_out/r06_l_sizes.jsonl')]
t=collections.defaultdict(dict)
for r in rows:
    if 'size' in r: t[(r['size'],r['lights'])].setdefault(r['label'].split('_')[0],[]).append(r['shade_ms'])
print("size lights | before: best median | after: best median | change of the median")
for k,v in t.items():
    b,a=sorted(v['before']),sorted(v['after'])
    print(k[0],k[1],'|',b[0],b[len(b)//2],'|',a[0],a[len(a)//2],'|',f"{(a[len(a)//2]/b[len(b)//2]-1)*100:+.1f} %")
for r in rows:
    if 'fit_intercept_us' in r: print(r)
PY
