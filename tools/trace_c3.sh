# per-dispatch kernel trace of one C3 bf16 step (run on the GPU box from the repo root)
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 250 rocprofv3 --kernel-trace -d $R/gpurun_out/c3trace --output-format csv -- python3 $R/bench.py --workload bruteforce --precision bf16_rerank --steps 1 --warmup 1 > $R/gpurun_out/c3trace.log 2>&1
