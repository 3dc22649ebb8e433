#!/bin/bash
# round 6, batch p: throughput mode with room made inside the compute unit for the previous frame's tail:
# shade capped at 4 blocks per CU (LDS pad) + the bloom chain on 256-thread tiles that fit the hole, against the product configuration
set -o pipefail
out=gpurun_out; mkdir -p $out
export PBR_HIP_LIB=$PWD/direct12pbrrenderer_amd/libpbr_hip_knobs.so
for r in 1 2; do
  timeout -k 10 600 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-host-graph --no-shade-paths --no-configs > $out/r06_p_base_$r.json 2>$out/r06_p_base_$r.err || exit 1
  PBR_SHADE_LDS_PAD=18 PBR_BLOOM_TILE=16 PBR_BLOOM_WIDE=0 timeout -k 10 600 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-host-graph --no-shade-paths --no-configs > $out/r06_p_hole_$r.json 2>$out/r06_p_hole_$r.err || exit 1
  PBR_SHADE_LDS_PAD=18 timeout -k 10 600 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-host-graph --no-shade-paths --no-configs > $out/r06_p_padonly_$r.json 2>$out/r06_p_padonly_$r.err || exit 1
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06_p_*.json')):
    d=json.loads(open(f).read().strip().split('\n')[-1])
    tm=d['config'].get('throughput_mode',{})
    k=d.get('kernels',{})
    print(f.split('/')[-1], 'in-order ms', d['ms_per_step'], 'tail', tm.get('tail',{}).get('ms_per_step'), tm.get('tail',{}).get('reproduces_in_order_frames'), 'post_shade', tm.get('post_shade',{}).get('ms_per_step'), tm.get('post_shade',{}).get('reproduces_in_order_frames'),
          'shade', k.get('shade(in frame)',{}).get('ms'), 'bloom+hist', k.get('bloom+histogram',{}).get('ms'))
PY
