#!/bin/bash
# one A/B build of the product library with extra compile flags for shade.hip: bash tools/build_variant.sh <tag> [-DPBR_EXP_...]
# -> tools/ab/libpbr_<tag>.so (git-ignored; travels to the GPU box), used through PBR_HIP_LIB / tools/ab_libs.sh
tag=$1; shift
cd "$(dirname "$0")/../direct12pbrrenderer_amd/csrc" && mkdir -p ../../tools/ab &&
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -I../../include "$@" -c shade.hip -o ../../tools/ab/shade_$tag.o &&
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../tools/ab/libpbr_$tag.so ctx.o ibl.o cluster.o ../../tools/ab/shade_$tag.o raster.o bloom.o exposure.o -ldl && echo built $tag
