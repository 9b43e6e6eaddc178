# usage (on the GPU box): bash tools/potrf_trace.sh -- kernel trace of one headline solve; durations of the
# Cholesky chain kernels in launch order (every 6th call)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/pt
timeout -k 10 300 rocprofv3 --kernel-trace -d /tmp/pt -o t --output-format csv -- python3 $R/bench.py --config c3 --steps 1 --warmup 1 --no-cpu-baseline --no-host-path --no-parity-check --no-symv-events > /tmp/pt.log 2>&1
F=$(find /tmp/pt -name '*kernel_trace.csv' | head -1)
python3 - $F <<'PY'
import csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
diag = [(s, e) for s, e, n in rows if "potrf_diag_kernel" in n]
half = diag[len(diag) // 2:]          # the timed solve
print("potrf_diag per call (us):", " ".join("%.0f" % ((e - s) / 1e3) for s, e in half[::4]))
print("potrf span (ms): %.2f" % ((half[-1][1] - half[0][0]) / 1e6))
print("calls 32..55 (us):", " ".join("%.0f" % ((e - s) / 1e3) for s, e in half[32:56]))
# what runs between two consecutive diag kernels of the timed solve, in the middle of the factorisation
a, b = half[40][0], half[44][0]
for s, e, n in rows:
    if a <= s < b: print("  %8.1f us +%7.1f  %s" % ((s - a) / 1e3, (e - s) / 1e3, n.replace("(anonymous namespace)::", "").split("(")[0][:70]))
PY
