#!/usr/bin/env python3
"""rocprofv3 kernel-trace CSV -> a compact gzip'd CSV (stream, short kernel name, start ns, end ns, grid x, grid y) that fits
gpurun_out/:   python tools/trace_compact.py <kernel_trace.csv> <out.csv.gz>"""
import csv
import gzip
import sys

with open(sys.argv[1]) as f, gzip.open(sys.argv[2], "wt") as g:
    w = csv.writer(g)
    w.writerow(["stream", "kernel", "start", "end", "gx", "gy"])
    for r in csv.DictReader(f):
        nm = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        nm = nm.split("(")[0]
        w.writerow([r.get("Stream_Id", "0"), nm[:80], r["Start_Timestamp"], r["End_Timestamp"], r.get("Grid_Size_X", 0), r.get("Grid_Size_Y", 0)])
