#!/usr/bin/env python3
"""From a rocprofv3 kernel trace of one solve: wall time the tridiagonalisation spends on its
last K columns (per-column fixed costs), by K."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select start, start + duration, duration from kernels where name like '%symv_kernel%' order by start").fetchall()
n = len(rows) + 1
col = db.execute("select start, start + duration, duration from kernels where name like '%colupd_kernel%' order by start").fetchall()
end = col[-1][1]
print("symv launches", len(rows), "first-to-last wall %.1f ms" % ((end - rows[0][0]) / 1e6))
for K in (512, 1024, 2048, 2816, 4096, 8192):
    s = rows[len(rows) - K]
    wall = (end - s[0]) / 1e6
    ssum = sum(r[2] for r in rows[len(rows) - K:]) / 1e6
    csum = sum(r[2] for r in col if r[0] >= s[0]) / 1e6
    print("last %5d columns: wall %7.2f ms (%.1f us/col), symv %.2f ms, colupd %.2f ms, other+gaps %.2f ms"
          % (K, wall, 1e3 * wall / K, ssum, csum, wall - ssum - csum))
