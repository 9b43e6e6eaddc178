#!/usr/bin/env python3
"""Kernels of a rocprofv3 kernel trace (/tmp/ktx) between the LAST launch matching <from> and the FIRST later launch matching
<to>: per kernel name count and time, idle time, and the calls in order.  e.g. the reduction to standard form:
    python3 tools/stage_calls.py potrf_diag_kernel "panel_kernel<0>" """
import csv, glob, collections, sys
f = glob.glob('/tmp/ktx/**/*kernel_trace.csv', recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name'].replace('ek::(anonymous namespace)::', '').replace('void ', '').replace('ek::', '')
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), n.split('(')[0][:48],
                 int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) // int(r['Workgroup_Size_X'])))
rows.sort()
a, b = sys.argv[1], sys.argv[2]
i1 = min(i for i, r in enumerate(rows) if b in r[2])
i0 = max(i for i, r in enumerate(rows[:i1]) if a in r[2]) + 1
win = rows[i0:i1]
print("span %.2f ms, %d kernels" % ((win[-1][1] - win[0][0]) / 1e6, len(win)))
acc = collections.defaultdict(lambda: [0, 0.0])
for s, e, n, g in win:
    acc[n][0] += 1; acc[n][1] += (e - s) / 1e3
for n, (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]): print("%-50s %5d %10.1f us" % (n, c, t))
busy = 0; ce = win[0][0]
for s, e, _, _ in win:
    if e > ce: busy += e - max(s, ce); ce = e
print("idle %.2f ms" % ((win[-1][1] - win[0][0] - busy) / 1e6))
if len(sys.argv) > 3:
    for s, e, n, g in win: print("%9.1f us %6d wg  %s" % ((e - s) / 1e3, g, n))
