#!/bin/bash
# one A/B build of the product library with extra compile flags for ONE translation unit:
#   bash tools/build_variant.sh <tag> <shade|bloom|ibl|...> [-DPBR_EXP_...]
# -> tools/ab/libpbr_<tag>.so (git-ignored; travels to the GPU box), used through PBR_HIP_LIB / tools/ab_libs.sh / tools/ab_shade_ms.sh
tag=$1; tu=$2; shift 2
cd "$(dirname "$0")/../direct12pbrrenderer_amd/csrc" && mkdir -p ../../tools/ab || exit 1
extra=""; case $tu in bloom|ibl|raster) extra="-ffp-contract=off";; esac
[ $tu = ibl ] && extra="$extra -fno-slp-vectorize"
objs=""; for o in ctx ibl cluster shade raster bloom exposure; do [ $o = $tu ] && objs="$objs ../../tools/ab/${tu}_$tag.o" || objs="$objs $o.o"; done
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -I../../include $extra "$@" -c ${SRC:-$tu.hip} -o ../../tools/ab/${tu}_$tag.o &&
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../tools/ab/libpbr_$tag.so $objs -ldl && echo built $tag
