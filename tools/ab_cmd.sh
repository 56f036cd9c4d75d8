cd $GRAFT_REPO_ROOT
for sh in "1025 3072 768" "1024 3072 768" "1025 2304 768" "2050 3072 768" "1281 3072 768"; do for c in -1 2 3; do python tools/mb_one.py bf16x3 $sh $c 2>&1 | tail -1; done; done
