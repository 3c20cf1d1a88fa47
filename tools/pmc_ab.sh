# Counter passes of the traversal kernel for an environment switch (tools/exp_spec_ab.py), run on the GPU box from the
# repo root:  bash tools/pmc_ab.sh VAR "v1 v2" EF   ->  gpurun_out/pmcab_<v>_{a,b,c}/  + a printed summary
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
export EXP_NQ=10000 EXP_REPS=2 EXP_EF=$3
for v in $2; do
  export $1=$v
  timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM -d $R/gpurun_out/pmcab_${v}_a --output-format csv -- python3 $R/tools/exp_spec_ab.py pmc > $R/gpurun_out/pmcab_${v}_a.log 2>&1 || exit 1
  timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS -d $R/gpurun_out/pmcab_${v}_b --output-format csv -- python3 $R/tools/exp_spec_ab.py pmc > $R/gpurun_out/pmcab_${v}_b.log 2>&1 || exit 1
  timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_IFETCH SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU -d $R/gpurun_out/pmcab_${v}_c --output-format csv -- python3 $R/tools/exp_spec_ab.py pmc > $R/gpurun_out/pmcab_${v}_c.log 2>&1 || exit 1
done
cd $R && python3 tools/pmc_ab_summary.py $2
