# timing of bf16 GEMM loop variants (diagnostic: answers are wrong with parts left out);
#   bash tools/abl_gemm.sh "<sched>:<abl> ..."   e.g. "0:0 1:0 1:9"
R=${GRAFT_REPO_ROOT:-$(pwd)}
for sv in $1; do
  VDB_GEMM_SCHED=${sv%%:*} VDB_GEMM_ABL=${sv##*:} EXP_VARIANTS=0 EXP_ROUNDS=2 EXP_STEPS=3 timeout -k 10 120 python3 -u $R/tools/exp_gemm.py 2>&1 | grep "^0 \|recall" | sed "s/^/sched:abl $sv: /" || exit 1
done
