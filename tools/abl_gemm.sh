# ablation timing of the bf16 GEMM loop (diagnostic: answers are wrong with parts left out); bash tools/abl_gemm.sh "0 1 2 ..."
R=${GRAFT_REPO_ROOT:-$(pwd)}
for v in $1; do
  VDB_GEMM_ABL=$v EXP_VARIANTS=0 EXP_ROUNDS=2 EXP_STEPS=3 timeout -k 10 120 python3 -u $R/tools/exp_gemm.py 2>&1 | grep "^0 " | sed "s/^0 /abl $v: /" || exit 1
done
