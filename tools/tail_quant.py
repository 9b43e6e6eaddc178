#!/usr/bin/env python3
"""Tail-quantisation estimate from a rocprofv3 kernel trace (csv): for every launch of the library's GEMM kernels the
share of its duration that a launch whose workgroups came in whole rounds of the chip's slots would not have spent:
    loss = duration * (1 - wgs / (slots * ceil(wgs / slots)))      slots = 256 CUs x 2 workgroups
(an upper bound: workgroups do not run in lock-step rounds).   python tools/tail_quant.py trace.csv"""
import csv
import math
import sys
from collections import defaultdict

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append(r)
diag = [i for i, r in enumerate(rows) if "potrf_diag_kernel" in r["Kernel_Name"]]
# the LAST solve of the trace: from the last run of potrf_diag launches on
start = 0
if diag:
    # solves are separated by long gaps in the diag indices
    groups = [[diag[0]]]
    for i in diag[1:]:
        if i - groups[-1][-1] > 2000: groups.append([i])
        else: groups[-1].append(i)
    start = groups[-1][0]
acc = defaultdict(lambda: [0, 0.0, 0.0])
top = []
for r in rows[start:]:
    name = r["Kernel_Name"]
    if "gemm_kernel" not in name and "gemm_small" not in name and "symm_lower" not in name: continue
    gx = int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1)
    wx = int(r["Workgroup_Size_X"]) * int(r.get("Workgroup_Size_Y", 1) or 1) * int(r.get("Workgroup_Size_Z", 1) or 1)
    wgs = gx // max(wx, 1)
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    # slots: workgroups the chip holds at once (hipcc -Rpass-analysis=kernel-resource-usage: two workgroups per CU for the
    # 128 x 128 tilings and the SYMM, three for the 64 x 64 tiling; the trace's register columns are in other units)
    slots = 768 if "gemm_small" in name else 512
    loss = dur * (1.0 - wgs / (slots * math.ceil(wgs / slots))) if wgs > 0 else 0.0
    if wgs > slots: top.append((loss, dur, wgs, slots, name))
    key = name.replace("ek::(anonymous namespace)::", "").replace("void ", "").split("(")[0][:44]
    a = acc[key]; a[0] += 1; a[1] += dur; a[2] += loss
print("%-46s %6s %10s %10s" % ("kernel", "calls", "ms", "tail ms"))
tot = 0.0
for k, (c, d, l) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print("%-46s %6d %10.2f %10.2f" % (k, c, d, l)); tot += l
print("total tail estimate: %.2f ms" % tot)
print("largest single tails among launches of more than one round (ms lost, ms, workgroups, slots):")
for loss, dur, wgs, slots, name in sorted(top, reverse=True)[:25]:
    print("  %6.3f %8.3f %7d %5d  %s" % (loss, dur, wgs, slots, name.replace("ek::(anonymous namespace)::", "").replace("void ", "").split("(")[0][:50]))
