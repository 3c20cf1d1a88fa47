# rocprofv3 passes over the default bench (C2 HNSW): kernel trace + stats, then FETCH_SIZE counters in a run of their own
# (run on the GPU box from the repo root)
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 250 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/c2prof --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-ef-block --no-bruteforce --no-c4-block --no-c5-block --steps 20 --warmup 3 > $R/gpurun_out/c2prof.log 2>&1
timeout -s KILL 250 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/c2pmc --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-ef-block --no-bruteforce --no-c4-block --no-c5-block --steps 3 --warmup 1 > $R/gpurun_out/c2pmc.log 2>&1
