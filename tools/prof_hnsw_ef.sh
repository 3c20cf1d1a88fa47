# rocprofv3 counter passes of hnsw_search_kernel on the C2 graph at several ef (run on the GPU box from the repo
# root): SQ counters and FETCH_SIZE in runs of their own, one pair per ef.  Output: gpurun_out/efpmc_<ef>_{sq,fetch}
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
export EXP_BUILD=host EXP_NQ=10000 EXP_REPS=4
for ef in ${EFS:-32 100 256}; do
  EXP_EF=$ef timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS -d $R/gpurun_out/efpmc_${ef}_sq --output-format csv -- python3 $R/tools/exp_hnsw.py > $R/gpurun_out/efpmc_${ef}_sq.log 2>&1
  EXP_EF=$ef timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/efpmc_${ef}_fetch --output-format csv -- python3 $R/tools/exp_hnsw.py > $R/gpurun_out/efpmc_${ef}_fetch.log 2>&1
  grep "nq=" $R/gpurun_out/efpmc_${ef}_fetch.log
done
