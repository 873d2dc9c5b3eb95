import csv, collections, sys, glob
d = sys.argv[1]
f = glob.glob(d + '/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
mine = [r for r in rows if '(anonymous namespace)::k_' in r['Kernel_Name'] and 'at::native' not in r['Kernel_Name']]
rs = sorted(mine, key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rs) if 'k_gemm<1, 8, 2>' in r['Kernel_Name']]
a, b = idx[-3], idx[-2]
seg = rs[a + 2:b + 2]
t0 = int(seg[0]['Start_Timestamp'])
tot = 0; gaps = 0; prev = None
agg = collections.OrderedDict()
for r in seg:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name'].replace('void (anonymous namespace)::', '').split('(')[0]
    dd = (e - s) / 1e3; tot += dd
    agg.setdefault(name, [0, 0.0]); agg[name][0] += 1; agg[name][1] += dd
    if prev is not None: gaps += max(0, (s - prev) / 1e3)
    prev = e
span = (int(seg[-1]['End_Timestamp']) - t0) / 1e3
print(f"kernels in cycle: {len(seg)}  sum_dur={tot:.1f}us span={span:.1f}us gaps={gaps:.1f}us")
for k, (n, dsum) in agg.items(): print(f"  {k:36s} n={n:3d} tot={dsum:7.1f}us avg={dsum/n:6.1f}")
print(list(rows[0].keys()))
for r in seg[:14]:
    print(r['Kernel_Name'].replace('void (anonymous namespace)::','')[:44].ljust(44), f"{(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:7.1f}us", 'grid', r.get('Grid_Size_X'), r.get('Grid_Size_Y'), 'wg', r.get('Workgroup_Size_X'), 'vgpr', r.get('VGPR_Count'), 'lds', r.get('LDS_Block_Size'))
