#!/bin/bash
# round 6, batch q: the two-zone split (long blocks first, short blocks for the tail) re-swept at the small sizes (it was tuned at 4K in round 5)
set -o pipefail
out=gpurun_out; mkdir -p $out
export PBR_HIP_LIB=$PWD/direct12pbrrenderer_amd/libpbr_hip_knobs.so
for bf in 0.92 0.85 0.75 0.6 0.5; do for rs in 1 2; do
  PBR_SHADE_BIGFRAC=$bf PBR_SHADE_ROWS_SMALL=$rs timeout -k 10 300 python tools/shade_tile_ms.py bf${bf}_rs$rs 1280x720 1440x960 1920x1080 1928x2168@7680x4320 3840x2160 >> $out/r06_q_bigfrac.jsonl 2>> $out/r06_q_bigfrac.err || exit 1
done; done
python - <<'PY'
import json,collections
t=collections.defaultdict(dict)
for l in open('gpurun_out/r06_q_bigfrac.jsonl'):
    r=json.loads(l)
    if 'size' in r: t[(r['size'],r['lights'])][r['label']]=r['shade_ms']
labels=sorted({k for v in t.values() for k in v})
print('size lights', *labels)
for k,v in t.items(): print(k[0],k[1], *[v.get(l) for l in labels])
PY
