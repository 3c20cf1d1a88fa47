set -e
timeout -k 10 300 python -u -m pytest tests/test_build_gpu.py -x -q --timeout 300 --timeout-method thread 2>&1 | tail -3
timeout -k 10 400 python -u tests/fuzz_build.py 240 33 > gpurun_out/fuzz_build.log 2>&1 || { tail -5 gpurun_out/fuzz_build.log; exit 1; }
tail -2 gpurun_out/fuzz_build.log | cut -c1-600
bash tools/prof_c5_20m.sh > gpurun_out/c5_20m_pmc.log 2>&1 || { tail -30 gpurun_out/c5_20m_pmc.log; exit 1; }
tail -60 gpurun_out/c5_20m_pmc.log
