set -e
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/gputests.log 2>&1 || { tail -30 gpurun_out/gputests.log; exit 1; }
tail -3 gpurun_out/gputests.log
export EXP_N=20000000 EXP_EFS=100,128,160,200,252,300,400 EXP_SKIP_TRUTH=1
VDB_TRACE_SEARCH=1 timeout -k 10 400 python -u tools/exp_c5.py > gpurun_out/c5_20m_ladder.log 2>&1; grep -E "^ef=|built" gpurun_out/c5_20m_ladder.log
