#!/bin/bash
# tools/ab_attn.sh <dbg values...>: build attention_pipe.hip ablations (SIU3R_AP_DBG) as library variants and time the pair shape with each
cd "$(dirname "$0")/.."
for v in "$@"; do
  mkdir -p /tmp/ab_ap$v
  cp siu3r_amd/csrc/_obj/*.o /tmp/ab_ap$v/
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Iinclude -DSIU3R_AP_DBG=$v -c siu3r_amd/csrc/attention_pipe.hip -o /tmp/ab_ap$v/attention_pipe.o &
done
wait
for v in "$@"; do hipcc --offload-arch=gfx950 -shared -fPIC -o siu3r_amd/libsiu3r_hip_ap$v.so /tmp/ab_ap$v/*.o && echo built ap$v; done
