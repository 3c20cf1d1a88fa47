set -e
export EXP_N=20000000 EXP_EFS=128,160,200,252,300 EXP_SKIP_TRUTH=1
echo "== GH (half rows in flight)"; VDB_FULL_GPB=0 timeout -k 10 400 python -u tools/exp_c5.py 2>&1 | grep -E "^ef=|built"
echo "== full GPB with spill tables"; VDB_FULL_GPB=1 timeout -k 10 400 python -u tools/exp_c5.py 2>&1 | grep -E "^ef=|built"
