#!/bin/bash
# tools/ab_ppall.sh <tag> <extra hipcc flags...>: A/B build of ALL ping-pong GEMM units (gemm_pp.hip + gemm_pp_t{1,2,3}{x,b}.hip) as
# siu3r_amd/libsiu3r_hip_<tag>.so (select with SIU3R_LIB_OVERRIDE); the other objects come from the last full build
tag=$1; shift
cd "$(dirname "$0")/.."
mkdir -p /tmp/ab_$tag
cp siu3r_amd/csrc/_obj/*.o /tmp/ab_$tag/
for u in gemm_pp gemm_pp_t1x gemm_pp_t2x gemm_pp_t3x gemm_pp_t1b gemm_pp_t2b gemm_pp_t3b; do
  [ -n "$AB_ONLY" ] && [[ " $AB_ONLY " != *" $u "* ]] && continue
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -mllvm -pragma-unroll-threshold=1000000 -Iinclude "$@" -c siu3r_amd/csrc/$u.hip -o /tmp/ab_$tag/$u.o 2>&1 | grep -E "error" &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o siu3r_amd/libsiu3r_hip_$tag.so /tmp/ab_$tag/*.o && echo built $tag
