#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace into a text table: per-kernel calls,
total / average / min / max duration and share -- the `--stats` view, kept as a small text
file under profiles/ (the .db itself is scratch)."""
import sqlite3
import sys


def main(db_path, out=sys.stdout):
    db = sqlite3.connect(db_path)
    rows = db.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
        "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    out.write("%-86s %8s %14s %12s %10s %12s %7s\n" % ("kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "share"))
    for name, calls, tot, avg, mn, mx in rows:
        short = name if len(name) <= 86 else name[:83] + "..."
        out.write("%-86s %8d %14.3f %12.2f %10.2f %12.2f %6.2f%%\n"
                  % (short, calls, tot / 1e6, avg / 1e3, mn / 1e3, mx / 1e3, 100.0 * tot / total))
    out.write("TOTAL kernel time: %.3f ms over %d dispatches\n" % (total / 1e6, sum(r[1] for r in rows)))


def list_calls(db_path, pattern, out=sys.stdout):
    """Every dispatch whose kernel name contains `pattern`, in launch order, with its grid."""
    db = sqlite3.connect(db_path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    gx = "grid_x" if "grid_x" in cols else ("grid_size_x" if "grid_size_x" in cols else None)
    gy = gx.replace("x", "y") if gx else None
    wx = "workgroup_x" if "workgroup_x" in cols else ("workgroup_size_x" if "workgroup_size_x" in cols else None)
    sel = "name, start, duration" + (", %s, %s" % (gx, gy) if gx else "") + (", %s" % wx if wx else "")
    t0 = None
    for row in db.execute("select %s from kernels where name like ? order by start" % sel, ("%" + pattern + "%",)):
        t0 = row[1] if t0 is None else t0
        out.write("%10.3f ms  %9.1f us  %s  %s\n" % ((row[1] - t0) / 1e6, row[2] / 1e3, row[3:], row[0][:60]))


if __name__ == "__main__":
    if len(sys.argv) > 3 and sys.argv[2] == "--list":
        list_calls(sys.argv[1], sys.argv[3])
    else:
        main(sys.argv[1])
