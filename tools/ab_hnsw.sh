# A/B of one environment switch of the library on the C2 traversal (tools/exp_hnsw.py), alternating processes on one box:
#   bash tools/ab_hnsw.sh VAR "v1 v2 ..." [repeats]      (value "-" = unset)
R=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in $(seq ${3:-2}); do
for v in $2; do
  if [ "$v" = "-" ]; then unset $1; else export $1=$v; fi
  timeout -k 10 200 python3 -u $R/tools/exp_hnsw.py 2>&1 | grep "^nq=" | sed "s/^/$1=$v: /" || exit 1
done
done
