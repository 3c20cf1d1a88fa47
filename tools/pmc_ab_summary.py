#!/usr/bin/env python3
"""Means per first-level launch of the counters collected by tools/pmc_ab.sh, one column per variant."""
import collections, csv, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cols = {}
for v in sys.argv[1:]:
    agg, cnt = collections.defaultdict(float), collections.Counter()
    for part in "abc":
        for path in glob.glob(os.path.join(ROOT, "gpurun_out", f"pmcab_{v}_{part}", "*", "*counter_collection.csv")):
            for r in csv.DictReader(open(path)):
                if "hnsw_search_kernel" in r["Kernel_Name"] and int(r["Grid_Size"]) == 10000 * 64:
                    agg[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
                    agg["_vgpr"] = float(r["VGPR_Count"]); cnt["_vgpr"] = 1
    cols[v] = {k: agg[k] / cnt[k] for k in agg}
names = sorted(set().union(*[c.keys() for c in cols.values()]))
print(f"{'counter':28s}" + "".join(f"{v:>16s}" for v in cols))
for n in names:
    print(f"{n:28s}" + "".join(f"{cols[v].get(n, float('nan')):16.4g}" for v in cols))
