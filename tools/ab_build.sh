#!/bin/bash
# tools/ab_build.sh <tag> <extra hipcc flags...>: build an A/B variant of the library as siu3r_amd/libsiu3r_hip_<tag>.so
tag=$1; shift
cd "$(dirname "$0")/.."
mkdir -p /tmp/ab_$tag
for f in siu3r_amd/csrc/*.hip; do
  b=$(basename $f .hip)
  if [ "$b" = "gemm_dma" ] || [ ! -f siu3r_amd/csrc/_obj/$b.o ]; then
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Iinclude "$@" -c $f -o /tmp/ab_$tag/$b.o &
  else
    cp siu3r_amd/csrc/_obj/$b.o /tmp/ab_$tag/$b.o
  fi
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o siu3r_amd/libsiu3r_hip_$tag.so /tmp/ab_$tag/*.o && echo built $tag
