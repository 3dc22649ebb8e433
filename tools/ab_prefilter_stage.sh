export PBR_HIP_LIB=$PWD/direct12pbrrenderer_amd/libpbr_hip_knobs.so
for r in 1 2 3; do for v in 0 1; do PBR_PREFILTER_STAGE=$v python3 tools/cfg3_ms.py stage=$v 2>&1 | grep prefilter; done; done
