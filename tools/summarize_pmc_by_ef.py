#!/usr/bin/env python3
"""Per-ef counter means of the traversal kernel from the passes of tools/prof_hnsw_ef.sh (gpurun_out/efpmc_<ef>_{sq,fetch}):
python tools/summarize_pmc_by_ef.py r2 32 100 256  ->  profiles/r2/hnsw_c2_pmc_by_ef.json"""
import collections, csv, glob, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd, efs = sys.argv[1], sys.argv[2:]
out = {}
for ef in efs:
    rec = {}
    for part in ("sq", "fetch"):
        path = max(glob.glob(os.path.join(ROOT, "gpurun_out", f"efpmc_{ef}_{part}", "*", "*counter_collection.csv")), key=os.path.getmtime)
        agg, cnt = collections.defaultdict(float), collections.Counter()
        for r in csv.DictReader(open(path)):
            # the first-level launches of the 10 000-query batch (one block per query), not the large configuration's
            if "hnsw_search_kernel" in r["Kernel_Name"] and int(r["Grid_Size"]) == 10000 * 64:
                agg[r["Counter_Name"]] += float(r["Counter_Value"])
                cnt[r["Counter_Name"]] += 1
                rec["kernel"] = r["Kernel_Name"][:96]
                for f in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size"):
                    rec[f] = r[f]
        for name, v in agg.items():
            rec[name] = v / cnt[name]
    rec["issuing_fraction_of_wave_cycles"] = rec["SQ_ACTIVE_INST_ANY"] / rec["SQ_WAVE_CYCLES"]
    rec["valu_instructions_per_query"] = rec["SQ_INSTS_VALU"] / rec["SQ_WAVES"]
    rec["traffic_bytes_per_launch"] = rec["FETCH_SIZE"] * 1024 * 2  # gfx950: half the bytes of 16-B-per-lane loads are counted
    out[ef] = rec
dst = os.path.join(ROOT, "profiles", rnd, "hnsw_c2_pmc_by_ef.json")
json.dump(out, open(dst, "w"), indent=1)
for ef, rec in out.items():
    print(ef, rec["kernel"][-40:], f"issuing {rec['issuing_fraction_of_wave_cycles']:.3f}", f"VALU/query {rec['valu_instructions_per_query']:.0f}",
          f"traffic {rec['traffic_bytes_per_launch'] / 1e9:.2f} GB")
