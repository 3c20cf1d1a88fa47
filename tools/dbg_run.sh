set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for d in 0 7 6 8 12 14; do
  VDB_GEMM_DBG=$d timeout -k 10 150 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/dbg$d --output-format csv -- python3 $R/bench.py --workload bruteforce --precision bf16_rerank --base-rows 400000 --steps 2 --warmup 1 > $R/gpurun_out/dbg$d.log 2>&1
  echo "dbg $d done"
done
