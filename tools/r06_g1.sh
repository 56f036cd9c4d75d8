cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=$PWD/siu3r_amd
{
echo "== check s1"; SIU3R_LIB_OVERRIDE=$L/libsiu3r_hip_s1.so python tools/mb_pp.py check 2>&1 | grep -v "ok$" | tail -8
for t in tr0t1 tr1t1; do echo "== trace $t"; SIU3R_LIB_OVERRIDE=$L/libsiu3r_hip_$t.so python tools/pp_trace.py bf16x3 2048 4096 1024 1 2>&1 | tail -3; SIU3R_LIB_OVERRIDE=$L/libsiu3r_hip_$t.so python tools/pp_trace.py bf16x3 4096 4096 4096 1 2>&1 | tail -3; done
for t in tr0t2 tr1t2; do echo "== trace $t"; SIU3R_LIB_OVERRIDE=$L/libsiu3r_hip_$t.so python tools/pp_trace.py bf16x3 2048 4096 1024 2 2>&1 | tail -3; done
echo "== bench base"; python tools/mb_pp.py bench big 2>&1 | grep "^bench"
echo "== bench s1"; SIU3R_LIB_OVERRIDE=$L/libsiu3r_hip_s1.so python tools/mb_pp.py bench big 2>&1 | grep "^bench"
echo "== presplit base"; python tools/mb_presplit.py 2>&1 | tail -12
echo "== presplit s1"; SIU3R_LIB_OVERRIDE=$L/libsiu3r_hip_s1.so python tools/mb_presplit.py 2>&1 | tail -12
echo "== step base"; python bench.py --steps 20 --warmup 5 2>&1 | tail -1 | cut -c1-300
echo "== step s1"; SIU3R_LIB_OVERRIDE=$L/libsiu3r_hip_s1.so python bench.py --steps 20 --warmup 5 2>&1 | tail -1 | cut -c1-300
} > gpurun_out/r06_g1.txt 2>&1
tail -5 gpurun_out/r06_g1.txt
