set -e
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/gputests.log 2>&1 || { tail -30 gpurun_out/gputests.log; exit 1; }
tail -2 gpurun_out/gputests.log
export EXP_N=20000000 EXP_EFS=100,128,160,200,252,300 EXP_SKIP_TRUTH=1
timeout -k 10 400 python -u tools/exp_c5.py > gpurun_out/c5_20m_ladder.log 2>&1; grep -E "^ef=|built" gpurun_out/c5_20m_ladder.log
timeout -k 10 600 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_default.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')}, d['roofline'])
for k in ('traversal_by_ef','filtered_search','c4_one_gpu','c5_one_rank_shard','bruteforce'):
    if k in d: print(k, json.dumps(d[k])[:900])
PY
