set -e
for d in 16 32 64; do
echo "== div $d small"; VDB_BUILD_BATCH_DIV=$d timeout -k 10 400 python -u tools/exp_build_small_m.py 16
echo "== div $d big"; EXP_BIG=1 VDB_BUILD_BATCH_DIV=$d timeout -k 10 400 python -u tools/exp_build_small_m.py 6
done
