#!/usr/bin/env python3
"""Print the dfl kernels of a rocprofv3 --stats kernel_stats.csv with short names (the torch
kernel names in those files run to kilobytes).  usage: kstats.py <dir-or-csv> [min_calls]"""
import csv, glob, os, re, sys
p = sys.argv[1]
if os.path.isdir(p):
    p = max(glob.glob(os.path.join(p, "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)
rows = list(csv.DictReader(open(p)))
for r in rows:
    n = r["Name"]
    if "k_" not in n or ("anonymous" not in n and "dfl_k_" not in n):
        continue
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"\(.*", "", n).replace("void ", "")
    print(f"{n:32s} calls={int(r['Calls']):6d} avg={float(r['AverageNs'])/1e3:9.1f} us  total={float(r['TotalDurationNs'])/1e6:9.2f} ms  min={float(r['MinNs'])/1e3:8.1f} max={float(r['MaxNs'])/1e3:8.1f}")
