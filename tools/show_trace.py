"""Per-dispatch durations of the last brute-force step in a rocprofv3 --kernel-trace CSV (tools/trace_c3.sh)."""
import csv, glob, sys
import os
f = max(glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"), key=os.path.getmtime)
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'bf_prep_queries' in r['Kernel_Name']]
start = idx[-1]
t0 = int(rows[start]['Start_Timestamp'])
for r in rows[start:start + 60]:
    n = r['Kernel_Name'].replace('vdb::(anonymous namespace)::', '').replace('vdb::', '').split('(')[0][:40]
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} us  +{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:8.1f} us  {n}")
