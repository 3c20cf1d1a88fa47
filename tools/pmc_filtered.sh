# Counter passes of the filtered traversal kernel (tools/exp_filtered_one.py), run on the GPU box from the repo root:
#   EXP_SEL=0.1 EXP_EF=128 bash tools/pmc_filtered.sh  ->  printed means per launch of the last EXP_REPS first-level launches
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM --kernel-include-regex hnsw_search_kernel -d $R/gpurun_out/pmcf_a --output-format csv -- python3 $R/tools/exp_filtered_one.py > $R/gpurun_out/pmcf_a.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS --kernel-include-regex hnsw_search_kernel -d $R/gpurun_out/pmcf_b --output-format csv -- python3 $R/tools/exp_filtered_one.py > $R/gpurun_out/pmcf_b.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --kernel-include-regex hnsw_search -d $R/gpurun_out/pmcf_t --output-format csv -- python3 $R/tools/exp_filtered_one.py > $R/gpurun_out/pmcf_t.log 2>&1 || exit 1
cd $R && python3 - <<'PY'
import collections, csv, glob, os
reps = int(os.environ.get("EXP_REPS", 6))
for part in "ab":
    rows = collections.defaultdict(list)
    for path in glob.glob(f"gpurun_out/pmcf_{part}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(path)):
            if "true>" in r["Kernel_Name"].split("hnsw_search_kernel")[-1][:40]:
                rows[(r["Kernel_Name"].split("hnsw_search_kernel")[1][:30], r["Counter_Name"], r["Grid_Size"], r["VGPR_Count"], r["LDS_Block_Size"])].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    for key in sorted(rows):
        v = [x for _, x in sorted(rows[key])][-reps:]
        print(key, f"mean of last {len(v)}: {sum(v) / len(v):.4g}")
for path in glob.glob("gpurun_out/pmcf_t/*/*kernel_stats.csv"):
    print(open(path).read()[:1500])
PY
grep "ms per batch" gpurun_out/pmcf_*.log
rm -rf $R/gpurun_out/pmcf_a $R/gpurun_out/pmcf_b $R/gpurun_out/pmcf_t
