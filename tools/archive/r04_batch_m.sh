# round-4 batch m: polyphase kernel forced for ALL 2x-up levels (PBR_BLOOM_WIDE=1) against the default (levels of >= 400 tiles only)
mkdir -p gpurun_out
export PBR_HIP_LIB=$PWD/direct12pbrrenderer_amd/libpbr_hip_knobs.so
run() { tag=$1; shift
  env "$@" python3 bench.py --steps 200 --warmup 50 --no-cpu-baseline --no-host-graph --no-shade-paths --no-tail-overlap > gpurun_out/ab_$tag.json 2>/dev/null && python3 -c "
import json;d=json.load(open('gpurun_out/ab_$tag.json'));s=d['roofline']['stage_ms'];print('$tag', 'frame', d['ms_per_step'], 'bloom+histogram', s['bloom+histogram'], 'bloom', s['bloom'])" || exit 1
}
for r in 1 2; do run default X=1; run wide_all PBR_BLOOM_WIDE=1; done 2>&1 | tee gpurun_out/r04_m_ab_bloom_wide_all.txt
