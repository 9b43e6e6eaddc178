"""Reads a rocprofv3 --kernel-trace CSV and reports, per kernel name, count / total duration, and for the
whole trace the busy time against the span of a window (by default the last third of the calls, i.e. after
warm-up): how much of a launch-chain-bound stage is spent between kernels rather than in them."""
import csv, sys, collections

path = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60], int(r.get("Stream_Id", 0) or 0)))
rows.sort()
# window: from the last launch of a kernel matching `first` to the end of the stage
names = [r[2] for r in rows]
def last_index(sub):
    idx = [i for i, nm in enumerate(names) if sub in nm]
    return idx
first = sys.argv[3] if len(sys.argv) > 3 else "symm_lower"
idx = last_index(first)
# split the calls of the stage: a gap of more than 64 kernels without `first` separates solves
groups = []; cur = [idx[0]]
for a, b in zip(idx, idx[1:]):
    if rows[b][0] - rows[a][1] > 3_000_000: groups.append(cur); cur = []
    cur.append(b)
groups.append(cur)
g = groups[-1]
lo = g[0]
# the stage ends at the last kernel whose name matches pat after the group's last `first`
hi = g[-1]
while hi + 1 < len(rows) and (pat in rows[hi + 1][2]) and rows[hi + 1][0] - rows[hi][1] < 1_000_000: hi += 1
win = rows[lo:hi + 1]
t0, t1 = win[0][0], max(r[1] for r in win)
# union of busy intervals
busy = 0; ce = t0
for s, e, _, _ in win:
    if e > ce: busy += e - max(s, ce); ce = e
print(f"window: {len(win)} kernels, span {(t1 - t0) / 1e6:.3f} ms, busy (union) {busy / 1e6:.3f} ms, idle {(t1 - t0 - busy) / 1e6:.3f} ms")
acc = collections.defaultdict(lambda: [0, 0])
for s, e, nm, _ in win:
    acc[nm][0] += 1; acc[nm][1] += e - s
for nm, (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(f"  {nm:60s} {c:6d} {t / 1e6:9.3f} ms  {t / c / 1e3:8.1f} us")
# per-call durations of the kernels whose name contains argv[4] (every 6th call), in launch order
if len(sys.argv) > 4:
    for sub in sys.argv[4].split(","):
        calls = [(s, e) for s, e, nm, _ in win if sub in nm]
        print(sub, "per call (us):", " ".join("%.0f" % ((e - s) / 1e3) for s, e in calls[::6]))
