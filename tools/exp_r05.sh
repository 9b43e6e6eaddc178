cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/gp
timeout -k 10 300 rocprofv3 --kernel-trace -d /tmp/gp -o t --output-format csv -- python3 $R/bench.py --config c2 --steps 1 --warmup 1 --no-cpu-baseline --no-host-path --no-other-configs --no-parity-check --no-symv-events > /tmp/gp.log 2>&1
F=$(find /tmp/gp -name '*kernel_trace.csv' | head -1)
python3 - $F <<'PY'
import csv, sys
rows=[]
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ","").replace("ek::","").split("(")[0][:40]))
rows.sort()
# find the last solve: last chase_pos_kernel; take the symm launches before it
ic=[i for i,r in enumerate(rows) if "chase_pos" in r[2]][-1]
sy=[i for i,r in enumerate(rows[:ic]) if "symm_lower" in r[2]]
# group belonging to last solve: last 63
sy=sy[-63:]
for which in (5, 30, 55):
    a=sy[which]; b=sy[which+1]
    t0=rows[a][0]
    print("--- panel", which, "span us", (rows[b][0]-t0)/1e3)
    for r in rows[a:b]:
        print("  %-40s start %7.1f dur %6.1f" % (r[2], (r[0]-t0)/1e3, (r[1]-r[0])/1e3))
PY
