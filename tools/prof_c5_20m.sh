# Counter passes of the traversal kernels on a 20 M-row graph (tools/exp_c5.py, EXP_N = 20 000 000, k = 100), run on the
# GPU box from the repo root: SQ counters and FETCH_SIZE in runs of their own.  Output: gpurun_out/c5pmc_{sq,fetch}
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
export EXP_N=20000000 EXP_EFS=${EFS:-100,200} EXP_SKIP_TRUTH=1
timeout -s KILL 500 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --kernel-include-regex hnsw_search_kernel -d $R/gpurun_out/c5pmc_sq --output-format csv -- python3 $R/tools/exp_c5.py > $R/gpurun_out/c5pmc_sq.log 2>&1
timeout -s KILL 500 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex hnsw_search_kernel -d $R/gpurun_out/c5pmc_fetch --output-format csv -- python3 $R/tools/exp_c5.py > $R/gpurun_out/c5pmc_fetch.log 2>&1
grep "ef=" $R/gpurun_out/c5pmc_fetch.log
cd $R && python3 - <<'PY'
import collections, csv, glob, json, os
out = {}
for part in ("sq", "fetch"):
    for path in glob.glob(f"gpurun_out/c5pmc_{part}/*/*counter_collection.csv"):
        agg, cnt = collections.defaultdict(lambda: collections.defaultdict(float)), collections.defaultdict(collections.Counter)
        for r in csv.DictReader(open(path)):
            if "hnsw_search_kernel" in r["Kernel_Name"] and int(r["Grid_Size"]) == 10000 * 64:   # the query launches, not the build's
                key = r["Kernel_Name"].split("hnsw_search_kernel")[1][:28]
                agg[key][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[key][r["Counter_Name"]] += 1
        for key in agg:
            out.setdefault(key, {}).update({n: agg[key][n] / cnt[key][n] for n in agg[key]})
            out[key]["launches_" + part] = max(cnt[key].values())
for key, rec in out.items():
    if "SQ_WAVE_CYCLES" in rec:
        rec["issuing_fraction_of_wave_cycles"] = rec["SQ_ACTIVE_INST_ANY"] / rec["SQ_WAVE_CYCLES"]
        rec["wait_any_fraction_of_wave_cycles"] = rec["SQ_WAIT_ANY"] / rec["SQ_WAVE_CYCLES"]
    if "FETCH_SIZE" in rec:
        rec["traffic_bytes_per_launch"] = rec["FETCH_SIZE"] * 1024 * 2
json.dump(out, open("gpurun_out/c5_20m_pmc.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
PY
rm -rf $R/gpurun_out/c5pmc_sq $R/gpurun_out/c5pmc_fetch
