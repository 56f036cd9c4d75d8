# usage: bash tools/ab_run.sh <tag...>   mb_gemm with 128x128 tiles forced (SIU3R_GEMM_NARROW_MAX=0) for the base library and each A/B build
for v in base "$@"; do
  if [ $v = base ]; then unset SIU3R_LIB_OVERRIDE; else export SIU3R_LIB_OVERRIDE=$PWD/siu3r_amd/libsiu3r_hip_$v.so; fi
  echo "== $v"; SIU3R_GEMM_NARROW_MAX=0 python tools/mb_gemm.py 2>&1 | grep "M="
done
