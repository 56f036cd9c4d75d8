for v in base d2 d4 d20 d18 d6; do
  if [ $v = base ]; then unset SIU3R_LIB_OVERRIDE; else export SIU3R_LIB_OVERRIDE=$PWD/siu3r_amd/libsiu3r_hip_$v.so; fi
  echo "== $v"; python tools/mb_gemm.py 2>&1 | grep "M="
done
