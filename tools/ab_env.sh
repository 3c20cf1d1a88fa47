# A/B of one environment switch of the library on the C3 bf16 step, alternating processes on one box:
#   bash tools/ab_env.sh VAR "v1 v2 ..." [repeats]
R=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in $(seq ${3:-2}); do
for v in $2; do
  env $1=$v EXP_VARIANTS=0 EXP_ROUNDS=2 EXP_STEPS=3 timeout -k 10 120 python3 -u $R/tools/exp_gemm.py 2>&1 | grep "^0 " | sed "s/^0 /$1=$v: /" || exit 1
done
done
