"""Cycle anatomy from a rocprofv3 kernel trace of bench.py: splits our kernels into the
draft part and the target-verify part of one late cycle (boundaries = the two
lm_head argmax launches) and prints per-kernel totals.  usage: trace_cycle.py <dir>"""
import collections
import csv
import glob
import sys

d = sys.argv[1]
import os
rows = list(csv.DictReader(open(max(glob.glob(d + '/*kernel_trace.csv'), key=os.path.getmtime))))
mine = [r for r in rows if '(anonymous namespace)::k_' in r['Kernel_Name'] and 'at::native' not in r['Kernel_Name']]
rs = sorted(mine, key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rs) if 'k_gemm<1, false, 2' in r['Kernel_Name']]


def nm(r):
    return r['Kernel_Name'].replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', '').split('(')[0]


def report(title, seg):
    tot = gaps = 0.0
    prev = None
    agg = collections.OrderedDict()
    for r in seg:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        dd = (e - s) / 1e3
        tot += dd
        key = (nm(r), r['Grid_Size_X'], r['Grid_Size_Y'])
        agg.setdefault(key, [0, 0.0])
        agg[key][0] += 1
        agg[key][1] += dd
        if prev is not None:
            gaps += max(0, (s - prev) / 1e3)
        prev = e
    span = (int(seg[-1]['End_Timestamp']) - int(seg[0]['Start_Timestamp'])) / 1e3
    print(f"== {title}: {len(seg)} launches, sum of durations {tot:.1f} us, span {span:.1f} us, gaps {gaps:.1f} us")
    for (k, gx, gy), (n, dsum) in agg.items():
        print(f"   {k:24s} grid=({gx},{gy}) n={n:3d} total={dsum:8.1f} us avg={dsum / n:7.1f} us")


two_per_cycle = len(idx) >= 4 and all('k_argmax_finish' in nm(rs[i + 1]) for i in idx[-4:])
if len(idx) >= 6:
    # pattern per cycle with a native target: [draft ... argmax_d finish] [verify ... argmax_t finish] accept
    a, b, c = idx[-4], idx[-3], idx[-2]
    gap1 = int(rs[b]['Start_Timestamp']) - int(rs[a]['Start_Timestamp'])
    gap2 = int(rs[c]['Start_Timestamp']) - int(rs[b]['Start_Timestamp'])
    if abs(gap1 - gap2) > 0.3 * max(gap1, gap2):   # unequal halves: draft vs verify
        if gap1 > gap2:   # a->b is the verify part (longer)
            report("target verify (native)", rs[a + 2:b + 2])
            report("accept + draft", rs[b + 2:c + 2])
        else:
            report("accept + draft", rs[a + 2:b + 2])
            report("target verify (native)", rs[b + 2:c + 2])
    else:
        report("cycle (HF verify not shown)", rs[a + 2:b + 2])
else:
    report("cycle", rs[idx[-3] + 2:idx[-2] + 2])
