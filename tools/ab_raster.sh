#!/bin/bash
# tools/ab_raster.sh <tag> <extra hipcc flags...>: A/B build of raster.hip as siu3r_amd/libsiu3r_hip_<tag>.so (other objects from the last full build)
tag=$1; shift
cd "$(dirname "$0")/.."
mkdir -p /tmp/ab_$tag
cp siu3r_amd/csrc/_obj/*.o /tmp/ab_$tag/
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Iinclude "$@" -c siu3r_amd/csrc/raster.hip -o /tmp/ab_$tag/raster.o 2>&1 | grep -E "error"
hipcc --offload-arch=gfx950 -shared -fPIC -o siu3r_amd/libsiu3r_hip_$tag.so /tmp/ab_$tag/*.o && echo built $tag
