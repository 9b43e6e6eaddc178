#!/usr/bin/env python3
"""Summarise a rocprofv3 counter-collection CSV (`--pmc ... --output-format csv`): per kernel name
and counter, number of dispatches and the sum / mean of the counter value.  The CSV itself is scratch."""
import csv, glob, sys
from collections import defaultdict


def short(n):
    n = n.replace("ek::(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:70]


def main(pattern, out=sys.stdout):
    acc = defaultdict(lambda: [0, 0.0])
    for path in glob.glob(pattern, recursive=True):
        with open(path, newline="") as f:
            rd = csv.DictReader(f)
            for row in rd:
                name = row.get("Kernel_Name") or row.get("kernel_name") or "?"
                ctr = row.get("Counter_Name") or row.get("counter_name") or "?"
                val = float(row.get("Counter_Value") or row.get("counter_value") or 0.0)
                a = acc[(short(name), ctr)]
                a[0] += 1; a[1] += val
    out.write("%-72s %-34s %9s %18s %16s\n" % ("kernel", "counter", "dispatches", "sum", "mean"))
    for (name, ctr), (cnt, tot) in sorted(acc.items(), key=lambda kv: (-kv[1][1] if kv[0][1].startswith("SQ_INSTS_VALU_MFMA") else 0, kv[0])):
        out.write("%-72s %-34s %9d %18.6g %16.6g\n" % (name, ctr, cnt, tot, tot / max(cnt, 1)))


if __name__ == "__main__":
    main(sys.argv[1])
