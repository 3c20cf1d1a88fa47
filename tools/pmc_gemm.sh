# PMC passes over the bf16 GEMM of the C3 brute-force step (run on the GPU box from the repo root; counters only,
# one rocprofv3 run per pass).  Usage: bash tools/pmc_gemm.sh <tag>   -> gpurun_out/pmc_gemm_<tag>_{lds,sq}
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-x}
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES -d $R/gpurun_out/pmc_gemm_${TAG}_lds --output-format csv -- python3 $R/bench.py --workload bruteforce --precision bf16_rerank --base-rows 400000 --steps 1 --warmup 0 > $R/gpurun_out/pmc_gemm_${TAG}_lds.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc_gemm_${TAG}_sq --output-format csv -- python3 $R/bench.py --workload bruteforce --precision bf16_rerank --base-rows 400000 --steps 1 --warmup 0 > $R/gpurun_out/pmc_gemm_${TAG}_sq.log 2>&1
