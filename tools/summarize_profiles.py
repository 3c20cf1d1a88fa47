#!/usr/bin/env python3
"""Copies what the judge reads out of gpurun_out/ (scratch) into profiles/<round>/ (tracked): the rocprofv3
--kernel-trace --stats summaries and the PMC passes collected by tools/prof_c2.sh, tools/prof_c3.sh and
tools/pmc_gemm.sh <tag>.   python tools/summarize_profiles.py r2 <pmc_gemm tag>"""
import collections, csv, glob, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd, tag = sys.argv[1], sys.argv[2]
out = os.path.join(ROOT, "profiles", rnd)
os.makedirs(out, exist_ok=True)
newest = lambda pat: max(glob.glob(os.path.join(ROOT, "gpurun_out", pat)), key=os.path.getmtime)

shutil.copy(newest("c2prof/*/*kernel_stats.csv"), os.path.join(out, "hnsw_c2_kernel_stats.csv"))
have = lambda pat: bool(glob.glob(os.path.join(ROOT, "gpurun_out", pat)))
if have("c3prof/*/*kernel_stats.csv"):
    shutil.copy(newest("c3prof/*/*kernel_stats.csv"), os.path.join(out, "bf_c3_kernel_stats.csv"))

# FETCH_SIZE of the traversal at the headline configuration (ef = 32 -> the R = 1 instantiation)
rows = [r for r in csv.DictReader(open(newest("c2pmc/*/*counter_collection.csv")))
        if "hnsw_search_kernel<2, 8, 1, 0, true, false" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"
        and int(r["Grid_Size"]) == 10000 * 64]
kb = sum(float(r["Counter_Value"]) for r in rows) / len(rows)
json.dump({"kernel": rows[0]["Kernel_Name"].split("(vdb")[0].split("::")[-1] + " (NCH=2, GPB=8, R=1, VIS=0 = LDS hash, L2, bare, rows of exactly 128 floats)",
           "FETCH_SIZE_KB_mean": kb, "dispatches": len(rows), "traffic_bytes_per_launch": int(kb * 1024 * 2),
           "VGPR_Count": rows[0]["VGPR_Count"], "SGPR_Count": rows[0]["SGPR_Count"], "LDS_Block_Size": rows[0]["LDS_Block_Size"],
           "note": "traffic = FETCH_SIZE(KB)*1024*2: gfx950 reports half the bytes of 16-B-per-lane loads (MI355X_MICROARCH.md, HBM); "
                   "dispatches at ef=32 (10000 queries x 64 lanes)",
           "command": "tools/prof_c2.sh: rocprofv3 --pmc FETCH_SIZE --output-format csv -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 1"},
          open(os.path.join(out, "hnsw_c2_pmc_fetch.json"), "w"), indent=1)

print(open(os.path.join(out, "hnsw_c2_pmc_fetch.json")).read())
if not have(f"pmc_gemm_{tag}_sq/*/*counter_collection.csv"):
    sys.exit(0)  # (only the C2 passes were re-run)

# SQ / LDS counters of the bf16 GEMM, summed over the launches of one 400 000-row step
res = {}
for part in ("sq", "lds"):
    agg = collections.defaultdict(float)
    n = 0
    for r in csv.DictReader(open(newest(f"pmc_gemm_{tag}_{part}/*/*counter_collection.csv"))):
        if "bf_gemm8_filter" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
            n += r["Counter_Name"] == "SQ_WAVE_CYCLES"
    res[part] = dict(agg)
    res[part]["launches"] = n
sq, lds = res["sq"], res["lds"]
simd_cycles = sq["SQ_WAVE_CYCLES"] * 4 / 2  # quad-cycles summed over waves; two waves share a SIMD
json.dump({"kernel": "bf_gemm8_filter<false, 0> (bf16 ping-pong loop), all slab launches of one 400 000 x 960 step, 10 000 queries",
           "counters": res,
           "mfma_busy_fraction_of_simd_cycles": sq["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles,
           "wait_any_fraction_of_wave_cycles": sq["SQ_WAIT_ANY"] / sq["SQ_WAVE_CYCLES"],
           "wait_inst_any_fraction_of_wave_cycles": sq["SQ_WAIT_INST_ANY"] / sq["SQ_WAVE_CYCLES"],
           "lds_bank_conflict_fraction_of_lds_cycles": lds["SQ_LDS_BANK_CONFLICT"] / lds["SQ_LDS_IDX_ACTIVE"],
           "note": "SQ_WAVE_CYCLES etc. count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES cycles (MI355X_MICROARCH.md); half of every wave's "
                   "time is spent parked at a barrier by design (its SIMD partner has the matrix pipe)",
           "command": f"tools/pmc_gemm.sh {tag}"},
          open(os.path.join(out, "bf_c3_pmc_gemm.json"), "w"), indent=1)
print(open(os.path.join(out, "bf_c3_pmc_gemm.json")).read()[:1500])
