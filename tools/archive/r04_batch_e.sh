# round-4 batch e: the polyphase tail at 6 waves per SIMD (80 VGPRs, 16 B scratch, 4 private histograms, HDR loads behind the H pass): A/B on one box
mkdir -p gpurun_out
run() { # tag lib [env...]
  tag=$1; lib=$2; shift 2
  env "$@" PBR_HIP_LIB=$PWD/tools/ab/libpbr_$lib.so python3 bench.py --steps 200 --warmup 50 --no-cpu-baseline --no-host-graph --no-shade-paths --no-tail-overlap > gpurun_out/ab_$tag.json 2>/dev/null && python3 -c "
import json;d=json.load(open('gpurun_out/ab_$tag.json'));s=d['roofline']['stage_ms'];print('$tag', 'frame', d['ms_per_step'], 'shade in-frame', s['shade(in frame)'], 'bloom+histogram', s['bloom+histogram'], 'bloom', s['bloom'])" || exit 1
}
for r in 1 2; do
  run base base X=1
  run tail6 tail6 X=1
  run tail6_768 tail6b PBR_BLOOM_HIST_BLOCKS=768
done 2>&1 | tee gpurun_out/r04_e_ab_tail6.txt
export PBR_HIP_LIB=$PWD/direct12pbrrenderer_amd/libpbr_hip_knobs.so
for r in 1 2; do for v in 1 8; do PBR_PREFILTER_XCD=$v python tools/cfg3_ms.py "knobs xcd_groups=$v"; done; done 2>&1 | grep prefilter | tee gpurun_out/r04_e_prefilter_xcd_ab.txt
unset PBR_HIP_LIB
python tools/cfg3_ms.py "product" 2>&1 | grep prefilter | tee -a gpurun_out/r04_e_prefilter_xcd_ab.txt
