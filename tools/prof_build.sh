# rocprofv3 kernel trace of the device graph build (run on the GPU box from the repo root)
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
EXP_MODES=device timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/buildprof --output-format csv -- python3 $R/tools/exp_build.py > $R/gpurun_out/buildprof.log 2>&1
