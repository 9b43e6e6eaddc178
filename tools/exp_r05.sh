cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/gp
timeout -k 10 300 rocprofv3 --kernel-trace -d /tmp/gp -o t --output-format csv -- python3 $R/bench.py --config c3 --steps 1 --warmup 1 --no-cpu-baseline --no-host-path --no-other-configs --no-parity-check --no-symv-events > /tmp/gp.log 2>&1
F=$(find /tmp/gp -name '*kernel_trace.csv' | head -1)
python3 - $F <<'PY'
import csv, sys
rows=[]
rd=csv.DictReader(open(sys.argv[1]))
for r in rd:
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ","").replace("ek::","").split("(")[0][:34], r.get("Stream_Id") or r.get("Queue_Id")))
rows.sort()
ic=[i for i,r in enumerate(rows) if "chase_pos" in r[2]][-1]
sy=[i for i,r in enumerate(rows[:ic]) if "symm_lower" in r[2]]
sy=sy[-255:]
for which in (61,):
    a=sy[which]; b=sy[which+1]
    t0=rows[a][0]
    print("--- panel", which, "span us", (rows[b][0]-t0)/1e3)
    for r in rows[a:b+1]:
        print("  %-34s q%-3s start %8.1f end %8.1f dur %7.1f" % (r[2], r[3], (r[0]-t0)/1e3, (r[1]-t0)/1e3, (r[1]-r[0])/1e3))
PY
