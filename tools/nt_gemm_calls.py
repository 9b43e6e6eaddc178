#!/usr/bin/env python3
"""The calls of the 4-wave GEMM with a transposed second operand (the SYRK / SYR2K updates of the Cholesky factorisation
and of the reduction to standard form) in a rocprofv3 kernel trace: duration, workgroups, share of a full wave of 512.
    rocprofv3 --kernel-trace --output-format csv -d /tmp/ktx -o t -- python3 bench.py --steps 1 ... ; python3 tools/nt_gemm_calls.py"""
import csv, glob
f = glob.glob('/tmp/ktx/**/*kernel_trace.csv', recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name']
    if 'gemm_kernel<false, true' in n:
        rows.append((int(r['Start_Timestamp']), (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3,
                     int(r['Grid_Size_X']) // int(r['Workgroup_Size_X'])))
rows.sort()
tot = 0
for t, d, g in rows:
    print("%9.1f us  %5d workgroups" % (d, g)); tot += d
print('total us', tot)
