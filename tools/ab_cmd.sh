cd $GRAFT_REPO_ROOT
L=$PWD/siu3r_amd/libsiu3r_hip_
for t in xw1 xw0; do for sh in "4096 4096 4096" "8192 8192 1024"; do SIU3R_LIB_OVERRIDE=${L}$t.so python tools/mb_one.py bf16x3 $sh 1 2>&1 | tail -1; done; done
for t in bw1 bw0; do for sh in "4096 4096 4096" "8192 8192 1024"; do SIU3R_LIB_OVERRIDE=${L}$t.so python tools/mb_one.py bf16 $sh 1 2>&1 | tail -1; done; done
python tools/mb_pp.py check 2>&1 | grep -v " ok$" | tail -5
