# PMC pass over the C3 brute-force GEMM (run on the GPU box from the repo root; counters only, no other trace domain)
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d $R/gpurun_out/c3pmc --output-format csv -- python3 $R/bench.py --workload bruteforce --precision bf16_rerank --base-rows 400000 --steps 1 --warmup 0 > $R/gpurun_out/c3pmc.log 2>&1
