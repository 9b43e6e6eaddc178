#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02c10; mkdir -p $O
(time timeout -k 10 800 python -m pytest tests -x -q -m gpu --durations=8) > $O/pytest_gpu.log 2>&1; echo "gpu tests rc=$?"; tail -15 $O/pytest_gpu.log
python bench.py --steps 3 --warmup 1 > $O/bench_c3.json 2> $O/bench_c3.err; echo "bench rc=$?"; tail -3 $O/bench_c3.err
python bench.py --config c2 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_c2.json 2>> $O/bench_c3.err; echo "bench c2 rc=$?"
python bench.py --config c5 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_c5.json 2>> $O/bench_c3.err; echo "bench c5 rc=$?"
python - <<'PY'
import json
for c in ("c3","c2","c5"):
    try:
        d=json.load(open("gpurun_out/r02c10/bench_%s.json"%c))
        print(c, round(d["ms_per_step"],1), round(d["value"]), {k.split(":")[-1]:round(v,4) for k,v in d["stage_seconds_per_step"].items() if v>1e-4})
        print("  roofline", {k:(round(v,3) if isinstance(v,float) else v) for k,v in (d.get("roofline") or {}).items() if k in ("bound","achieved","frac","avg_launch_us","launches")})
        print("  incl copies", d.get("value_incl_copies")); print("  cpu", {k:v for k,v in (d.get("cpu_baseline") or {}).items() if k in ("value","cores","seconds","gpu_same_order")})
    except Exception as e: print(c, "ERR", e)
PY
