"""One late steady-state cycle from a rocprofv3 --kernel-trace CSV of bench.py, EVERY kernel on the stream (torch's too):
prints the launches whose gap to the previous kernel exceeds a threshold, the small kernels, and the totals.
usage: trace_gaps.py <dir-or-csv> [gap_us=0.6]"""
import csv, glob, os, re, sys

p = sys.argv[1]
if os.path.isdir(p):
    p = max(glob.glob(os.path.join(p, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 0.6
rows = sorted(csv.DictReader(open(p)), key=lambda r: int(r["Start_Timestamp"]))


def nm(r):
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).replace("void ", "")
    return re.sub(r"\(.*", "", n)[:60]


acc = [i for i, r in enumerate(rows) if "k_accept_commit" in r["Kernel_Name"]]
a, b = acc[-3], acc[-2]          # one whole cycle between two accept launches, late in the run
seg = rows[a:b + 1]
span = (int(seg[-1]["End_Timestamp"]) - int(seg[0]["End_Timestamp"])) / 1e3
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg[1:]) / 1e3
print(f"cycle: {len(seg) - 1} launches, span {span:.1f} us, kernels {busy:.1f} us, gaps {span - busy:.1f} us")
for i in range(1, len(seg)):
    r, q = seg[i], seg[i - 1]
    gap = (int(r["Start_Timestamp"]) - int(q["End_Timestamp"])) / 1e3
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if gap > thr or dur < 6.0:
        print(f"  gap {gap:6.2f} us -> {nm(r):60s} {dur:7.2f} us   (after {nm(q)[:30]})")
