set -e
export EXP_N=20000000 EXP_EFS=160,200,252,300 EXP_SKIP_TRUTH=1
for kb in 20 24 32; do
echo "== spill-launch LDS cap $kb KB"; VDB_SPILL_LDS_KB=$kb VDB_TRACE_SEARCH=1 timeout -k 10 400 python -u tools/exp_c5.py > gpurun_out/cap$kb.log 2>&1; grep -E "^ef=" gpurun_out/cap$kb.log
done
