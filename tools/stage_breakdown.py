#!/usr/bin/env python3
"""Per-stage kernel breakdown of ONE solve from a rocprofv3 kernel trace: the Cholesky stage is
[first, last] potrf_diag launch, the reduction runs from there to the first symv launch."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, duration, grid_x, workgroup_x from kernels order by start").fetchall() \
    if "grid_x" in [r[1] for r in db.execute("pragma table_info(kernels)")] else \
    [(r[0], r[1], r[2], 0, 1) for r in db.execute("select name, start, duration from kernels order by start")]
def short(n):
    n = n.replace("ek::(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:48]
idx_pd = [i for i, r in enumerate(rows) if "potrf_diag" in r[0]]
i_sy = next(i for i, r in enumerate(rows) if "symv_kernel" in r[0])
stages = {"potrf": (idx_pd[0], idx_pd[-1] + 1), "sygst": (idx_pd[-1] + 1, i_sy - 1)}
for name, (a, b) in stages.items():
    seg = rows[a:b]
    wall = (seg[-1][1] + seg[-1][2] - seg[0][1]) / 1e6
    busy = sum(r[2] for r in seg) / 1e6
    print("%s: %d kernels, wall %.2f ms, busy %.2f ms, gaps %.2f ms" % (name, len(seg), wall, busy, wall - busy))
    agg = {}
    for r in seg:
        k = (short(r[0]), r[3] // max(r[4], 1))
        c = agg.setdefault(k, [0, 0.0]); c[0] += 1; c[1] += r[2] / 1e6
    for k, c in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
        print("   %-48s wgs %6d  x%4d  %8.3f ms" % (k[0], k[1], c[0], c[1]))
