#!/usr/bin/env python3
"""Idle time of the GPU inside ONE solve of a rocprofv3 kernel trace (csv): the union of the busy intervals of all streams
against the span from the solve's first kernel to its last, and the gaps by the pair (kernel that ended, kernel that
started).   python tools/idle_report.py trace.csv"""
import csv
import sys
from collections import defaultdict

def short(n):
    return n.replace("ek::(anonymous namespace)::", "").replace("void ", "").split("(")[0][:40]

def wgs_of(r):
    g = int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1)
    w = int(r["Workgroup_Size_X"]) * int(r.get("Workgroup_Size_Y", 1) or 1) * int(r.get("Workgroup_Size_Z", 1) or 1)
    return g // max(w, 1)

raw = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), wgs_of(r)) for r in csv.DictReader(open(sys.argv[1])))
rows = [(s, e, n) for s, e, n, _ in raw]
# the last solve: from the last launch of maxabs (the solve's first kernel) on, if present; else the whole trace
starts = [i for i, r in enumerate(rows) if "maxabs" in r[2] or "synth" in r[2]]
first = 0
for i in starts:
    if any("potrf_diag" in r[2] or "symm_lower" in r[2] for r in rows[i:i + 400]): first = i
rows = rows[first:]
# cut at the solve's last kernel: the last trsm / gemm before the verifier (take everything; the bench traced has no verifier)
span0, cur_end, busy = rows[0][0], rows[0][1], 0
gaps = defaultdict(lambda: [0, 0.0])
last_name = rows[0][2]
seg_start = rows[0][0]
for s, e, n in rows[1:]:
    if s > cur_end:
        busy += cur_end - seg_start
        g = (s - cur_end) / 1e3
        if g < 20000:                      # (a gap of 20 ms is the boundary to another solve or to the host)
            k = gaps[(last_name, n)]; k[0] += 1; k[1] += g
        seg_start = s
    if e > cur_end:
        cur_end = e; last_name = n
busy += cur_end - seg_start
span = cur_end - span0
tot_gap = sum(v[1] for v in gaps.values())
print("span %.1f ms, busy %.1f ms, idle inside (gaps < 20 ms) %.2f ms in %d gaps" % (span / 1e6, busy / 1e6, tot_gap / 1e3, sum(v[0] for v in gaps.values())))
print("%-42s %-42s %6s %9s %8s" % ("after", "before", "gaps", "total us", "mean us"))
for (a, b), (c, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:30]:
    print("%-42s %-42s %6d %9.0f %8.1f" % (a, b, c, t, t / c))

# time without a chip-filling kernel: the span minus the union of the launches of >= 256 workgroups; what runs then
big = [(s, e) for s, e, n, w in raw[first:] if w >= 256]
big.sort()
cover, ce, cs = 0, None, None
merged = []
for s, e in big:
    if ce is None or s > ce:
        if ce is not None: merged.append((cs, ce))
        cs, ce = s, e
    elif e > ce: ce = e
if ce is not None: merged.append((cs, ce))
cover = sum(e - s for s, e in merged)
print("\nwithout a launch of >= 256 workgroups in flight: %.1f ms of the span; small launches by their time outside the big ones:" % ((span - cover) / 1e6))
import bisect
ms = [m[0] for m in merged]
out = defaultdict(lambda: [0, 0.0])
for s, e, n, w in raw[first:]:
    if w >= 256: continue
    # part of [s, e) not covered by merged
    t, i = s, max(bisect.bisect_right(ms, s) - 1, 0)
    un = 0
    while t < e and i < len(merged):
        a, b = merged[i]
        if b <= t: i += 1; continue
        if a >= e: break
        if a > t: un += a - t
        t = max(t, b); i += 1
    if t < e: un += e - t
    if un > 0: o = out[n]; o[0] += 1; o[1] += un / 1e3
for n, (c, t) in sorted(out.items(), key=lambda kv: -kv[1][1])[:16]:
    print("  %-42s %6d launches %9.0f us" % (n, c, t))
