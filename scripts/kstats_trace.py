#!/usr/bin/env python3
"""Per (kernel, grid) averages from a rocprofv3 --kernel-trace CSV of bench.py: duration, the gap to the NEXT dfl
kernel on the stream, calls — the in-situ view (o_proj and qkv share a template instantiation but not a grid).
usage: kstats_trace.py <dir-or-csv> [skip_first_n_dispatches]"""
import collections, csv, glob, os, re, sys
p = sys.argv[1]
if os.path.isdir(p):
    p = max(glob.glob(os.path.join(p, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rows = sorted(csv.DictReader(open(p)), key=lambda r: int(r["Start_Timestamp"]))[skip:]


def nm(r):
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    return re.sub(r"\(.*", "", n).replace("void ", "")


agg = collections.OrderedDict()
for i, r in enumerate(rows):
    if ("anonymous namespace" not in r["Kernel_Name"] and "dfl_k_" not in r["Kernel_Name"]) or "k_" not in r["Kernel_Name"]:
        continue
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (int(rows[i + 1]["Start_Timestamp"]) - e) / 1e3 if i + 1 < len(rows) else 0.0
    a = agg.setdefault((nm(r), int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Grid_Size_Y"])), [0, 0.0, 0.0])
    a[0] += 1
    a[1] += (e - s) / 1e3
    if gap < 50:
        a[2] += gap
tot = sum(a[1] for a in agg.values())
for (n, gx, gy), a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{n:34s} grid {gx:4d}x{gy:<2d} calls={a[0]:6d} avg={a[1] / a[0]:8.2f} us  gap after={a[2] / a[0]:5.2f} us  share={100 * a[1] / tot:5.1f} %")
