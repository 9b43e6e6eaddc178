#!/usr/bin/env python3
"""Reads a rocprofv3 --kernel-trace CSV of a team rehearsal of the dense -> band stage (tools/team_trace_run.py) and shows
the look-ahead at work: for every panel chain (its hr_kernel is the marker; the chain runs on the second stream) which
kernels of the OTHER stream -- the rest of the previous panel's trailing update on the members' strips -- ran while
the chain did.
    python tools/team_overlap_report.py <kernel_trace.csv> [panels to print]"""
import collections
import csv
import sys

path = sys.argv[1]
nshow = int(sys.argv[2]) if len(sys.argv) > 2 else 6
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        nm = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), nm, r.get("Stream_Id", r.get("Queue_Id", "0"))))
rows.sort()
t00 = rows[0][0]
by_stream = collections.Counter(r[3] for r in rows)
print("streams (id: dispatches):", dict(by_stream))
chain_names = ("panel_kernel", "reduce_parts_kernel", "chol_kernel", "hr_kernel", "house_tall_kernel", "tall_finish_kernel",
               "t_from_gram_kernel")
hr = [r for r in rows if r[2].startswith("ek::hr_kernel") or r[2].endswith("hr_kernel")]
gem = [r for r in rows if "gemm_kernel" in r[2]]
print("%d hr_kernel launches (one per CholeskyQR2 panel chain), %d GEMM launches" % (len(hr), len(gem)))
# for each chain marker: GEMM time on another stream that overlaps [start of the chain's first kernel, end of its last]
chains = []
i = 0
names = [r[2] for r in rows]
for h in hr:
    st = h[3]
    # the chain = contiguous run of chain kernels on the same stream around the marker
    k = rows.index(h)
    lo = k
    while lo > 0 and rows[lo - 1][3] == st and any(c in rows[lo - 1][2] for c in chain_names) and rows[lo][0] - rows[lo - 1][1] < 200_000: lo -= 1
    hi = k
    while hi + 1 < len(rows) and rows[hi + 1][3] == st and any(c in rows[hi + 1][2] for c in chain_names) and rows[hi + 1][0] - rows[hi][1] < 200_000: hi += 1
    # (kernels of other streams interleave in the sorted list: walk by stream)
    same = [r for r in rows[max(0, k - 40):k + 40] if r[3] == st and any(c in r[2] for c in chain_names)]
    c0 = min(r[0] for r in same if abs(r[0] - h[0]) < 400_000)
    c1 = max(r[1] for r in same if abs(r[0] - h[0]) < 400_000)
    other = [(max(g[0], c0), min(g[1], c1), g[2], g[3]) for g in rows if g[3] != st and g[1] > c0 and g[0] < c1]
    ov = sum(e - s for s, e, _, _ in other)
    chains.append((c0, c1, st, ov, other))
with_ov = [c for c in chains if c[3] > 0]
tot_chain = sum(c[1] - c[0] for c in chains)
tot_ov = sum(min(c[3], c[1] - c[0]) for c in chains)
print("chains whose span overlaps kernels of another stream: %d of %d; chain time %.3f ms, of which %.3f ms (%.0f %%) beside "
      "another stream's kernels" % (len(with_ov), len(chains), tot_chain / 1e6, tot_ov / 1e6, 100.0 * tot_ov / max(tot_chain, 1)))
print("first chains with an overlap (times in us from the first kernel of the trace):")
for c0, c1, st, ov, other in with_ov[:nshow]:
    print("  chain on stream %s: %.1f .. %.1f us (%.1f us); meanwhile on other streams:" % (st, (c0 - t00) / 1e3, (c1 - t00) / 1e3, (c1 - c0) / 1e3))
    acc = collections.OrderedDict()
    for s, e, nm, so in other:
        key = (so, nm[:70])
        a = acc.setdefault(key, [0, 0, s, e]); a[0] += 1; a[1] += e - s; a[2] = min(a[2], s); a[3] = max(a[3], e)
    for (so, nm), (cnt, dur, s, e) in acc.items():
        print("      stream %s  %-70s x%d  %.1f us inside the chain's span (%.1f .. %.1f)" % (so, nm, cnt, dur / 1e3, (s - t00) / 1e3, (e - t00) / 1e3))
