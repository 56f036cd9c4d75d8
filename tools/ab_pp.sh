#!/bin/bash
# tools/ab_pp.sh <tag> "<X3,MI,NJ,MODE>" <extra hipcc flags...>: single-variant A/B build of the ping-pong GEMM as
# siu3r_amd/libsiu3r_hip_<tag>.so (select with SIU3R_LIB_OVERRIDE); the other objects come from the last full build
tag=$1; mini=$2; shift; shift
cd "$(dirname "$0")/.."
mkdir -p /tmp/ab_$tag
cp siu3r_amd/csrc/_obj/*.o /tmp/ab_$tag/
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -mllvm -pragma-unroll-threshold=1000000 -Iinclude "-DSIU3R_PP_MINI=$mini" "$@" -c siu3r_amd/csrc/gemm_pp.hip -o /tmp/ab_$tag/gemm_pp.o 2>&1 | grep -E "error" 
hipcc --offload-arch=gfx950 -shared -fPIC -o siu3r_amd/libsiu3r_hip_$tag.so /tmp/ab_$tag/*.o && echo built $tag
