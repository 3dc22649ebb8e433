#!/bin/bash
# builds the stamped variant of the library (here, before gpurun) : bash tools/debug/tail_timing.sh build [tile]
# and runs the measurement on the GPU box                         : bash tools/debug/tail_timing.sh run
set -e
cd "$(dirname "$0")/../.."
if [ "$1" = build ]; then
  cd direct12pbrrenderer_amd/csrc && make -s && mkdir -p ../../tools/debug/bin
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -ffp-contract=off -DPBR_BLOOM_TIMING=${2:-1} -c bloom.hip -o /tmp/bloom_timing.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../tools/debug/bin/libpbr_hip_timing.so ctx.o ibl.o cluster.o shade.o raster.o /tmp/bloom_timing.o exposure.o -ldl
else
  PBR_HIP_LIB=$PWD/tools/debug/bin/libpbr_hip_timing.so python3 tools/debug/tail_timing.py
fi
