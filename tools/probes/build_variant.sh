#!/bin/bash
# Developer A/B: build _ab/lib_<tag>.so (git-ignored, travels with gpurun) with extra -D flags for ONE translation unit (the others are
# linked from csrc/_build as the in-tree library has them).  usage: build_variant.sh <tag> <unit.hip> <flags...>
set -e
R=$(cd $(dirname $0)/../.. && pwd)
tag=$1; unit=$2; shift 2
mkdir -p $R/_ab /tmp/ab_$tag
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -fno-slp-vectorize "$@" -I $R/include -c -o /tmp/ab_$tag/unit.o $R/cspn_monodepth_amd/csrc/$unit
objs=""
for o in $R/cspn_monodepth_amd/csrc/_build/*.o; do
  [ "$(basename $o .o)" = "$(basename $unit .hip)" ] && objs="$objs /tmp/ab_$tag/unit.o" || objs="$objs $o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/_ab/lib_$tag.so $objs
echo built $R/_ab/lib_$tag.so
