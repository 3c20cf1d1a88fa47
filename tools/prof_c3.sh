# rocprofv3 kernel trace of the C3 brute-force bench (run on the GPU box from the repo root)
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 250 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/c3prof --output-format csv -- python3 $R/bench.py --workload bruteforce --precision bf16_rerank --steps 3 --warmup 1 > $R/gpurun_out/c3prof.log 2>&1
