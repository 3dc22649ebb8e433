#!/bin/bash
# round 6, batch r: the shade at 5 / 6 / 7 / 8 waves per SIMD (96 / 80 / 72 / 64 registers, scratch 12 / 68 / 112 / 196 B) at 1 and 256 lights:
# is the single-light frame, which is bound by the latency of its IBL gathers, better off with more waves and spills?
set -o pipefail
out=gpurun_out; mkdir -p $out
for r in 1 2; do for w in 5 6 7 8; do
  PBR_HIP_LIB=$PWD/tools/ab/libpbr_w$w.so timeout -k 10 300 python tools/shade_tile_ms.py w${w}_$r 1440x960 1920x1080 3840x2160 >> $out/r06_r_occ.jsonl 2>> $out/r06_r_occ.err || exit 1
done; done
python - <<'PY'
import json,collections
t=collections.defaultdict(dict)
for l in open('gpurun_out/r06_r_occ.jsonl'):
    r=json.loads(l)
    if 'size' in r: t[(r['size'],r['lights'])].setdefault(r['label'].split('_')[0],[]).append(r['shade_ms'])
for k,v in t.items(): print(k[0],k[1], {a:min(b) for a,b in sorted(v.items())})
PY
