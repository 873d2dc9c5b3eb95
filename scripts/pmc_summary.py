#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs, scripts/profile_gpu.sh pmc
or pmcb) into one CSV: per dfl kernel the mean KB per dispatch and the corrected HBM bytes.
On gfx950 FETCH_SIZE counts half of a wide coalesced read stream (MI355X_MICROARCH.md §HBM), so
hbm_bytes = 2 * FETCH_SIZE_KB * 1024 + WRITE_SIZE_KB * 1024.
usage: pmc_summary.py <dir with pmc_FETCH_SIZE*/ and pmc_WRITE_SIZE*/> [suffix] > out.csv"""
import csv, glob, os, re, sys
from collections import defaultdict

root = sys.argv[1]
suffix = sys.argv[2] if len(sys.argv) > 2 else ""


def load(counter):
    d = os.path.join(root, f"pmc_{counter}{suffix}")
    f = max(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    acc = defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        n = r["Kernel_Name"]
        if "k_" not in n or ("anonymous" not in n and "dfl_k_" not in n):
            continue
        n = re.sub(r"\(anonymous namespace\)::", "", n)
        n = re.sub(r"\(.*", "", n).replace("void ", "")
        acc[(n, int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    return acc


fs, ws = load("FETCH_SIZE"), load("WRITE_SIZE")
print("kernel,grid_threads,dispatches,FETCH_SIZE_KB_mean,WRITE_SIZE_KB_mean,hbm_bytes_corrected")
for key in sorted(fs):
    f = sum(fs[key]) / len(fs[key])
    w = sum(ws.get(key, [0.0])) / max(1, len(ws.get(key, [0.0])))
    print(f'"{key[0]}",{key[1]},{len(fs[key])},{f:.1f},{w:.1f},{int(2 * f * 1024 + w * 1024)}')
