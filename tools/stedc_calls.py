#!/usr/bin/env python3
"""The kernels of the divide & conquer stage in a rocprofv3 kernel trace (/tmp/ktx, as tools/nt_gemm_calls.py): from the leaf
kernel to the last merge product, with workgroups and durations; summary per kernel name."""
import csv, glob, collections
f = glob.glob('/tmp/ktx/**/*kernel_trace.csv', recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name'].replace('ek::(anonymous namespace)::', '').replace('void ', '')
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), n.split('(')[0][:44],
                 int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) // int(r['Workgroup_Size_X'])))
rows.sort()
i0 = max(i for i, r in enumerate(rows) if 'dc_leaf' in r[2])
i1 = max(i for i, r in enumerate(rows) if r[2].startswith('dc_'))
# the product of the last merge follows the last dc_ kernel
while i1 + 1 < len(rows) and 'gemm' in rows[i1 + 1][2]: i1 += 1
win = rows[i0:i1 + 1]
print("stage span %.2f ms, %d kernels" % ((win[-1][1] - win[0][0]) / 1e6, len(win)))
acc = collections.defaultdict(lambda: [0, 0.0])
for s, e, n, g in win:
    acc[n][0] += 1; acc[n][1] += (e - s) / 1e3
for n, (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]): print("%-46s %5d %10.1f us" % (n, c, t))
print("gemm calls in order (us, workgroups):")
print(" ".join("%.0f/%d" % ((e - s) / 1e3, g) for s, e, n, g in win if 'gemm' in n))
busy = 0; ce = win[0][0]
for s, e, _, _ in win:
    if e > ce: busy += e - max(s, ce); ce = e
print("idle inside the stage %.2f ms" % ((win[-1][1] - win[0][0] - busy) / 1e6))
