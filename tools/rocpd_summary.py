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


if __name__ == "__main__":
    main(sys.argv[1])
