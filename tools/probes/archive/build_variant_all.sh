#!/bin/bash
# Developer A/B: a whole library with extra -D flags for EVERY translation unit: build_variant_all.sh <tag> <flags...>
set -e
R=$(cd $(dirname $0)/../.. && pwd); tag=$1; shift
mkdir -p $R/cspn_monodepth_amd/ab /tmp/aball_$tag
objs=""
for src in $R/cspn_monodepth_amd/csrc/*.hip; do
  o=/tmp/aball_$tag/$(basename $src .hip).o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -fno-slp-vectorize "$@" -I $R/include -c -o $o $src &
  objs="$objs $o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/cspn_monodepth_amd/ab/libcspn_hip_$tag.so $objs
echo built $R/cspn_monodepth_amd/ab/libcspn_hip_$tag.so
