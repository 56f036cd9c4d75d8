#!/bin/bash
# tools/ab_split.sh: ablation build of the bf16x3 ping-pong tiles WITHOUT the in-loop hi / lo split of the fp32 A fragments (SIU3R_PP_DBG=8:
# the raw register halves go to the MFMAs -- wrong numbers, same MFMA / DMA / LDS work) as siu3r_amd/libsiu3r_hip_nosplit.so: what a
# pre-split A operand (hi | lo planes written by the producer's epilogue) could buy at most.  Select with SIU3R_LIB_OVERRIDE.
cd "$(dirname "$0")/.."
d=/tmp/ab_nosplit; mkdir -p $d; cp siu3r_amd/csrc/_obj/*.o $d/
for t in 1 2 3; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -mllvm -pragma-unroll-threshold=1000000 -Iinclude -DSIU3R_PP_DBG=8 \
    -c siu3r_amd/csrc/gemm_pp_t${t}x.hip -o $d/gemm_pp_t${t}x.o &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o siu3r_amd/libsiu3r_hip_nosplit.so $d/*.o && echo built nosplit
