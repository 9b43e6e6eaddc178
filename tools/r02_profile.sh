#!/bin/bash
# round-2 profile artefacts: bench lines of the BASELINE configs, rocprofv3 kernel summary of the headline
# command, PMC passes (MFMA instruction / busy counters; HBM traffic of the dominant kernel).
# usage (on the GPU box): bash tools/r02_profile.sh <tag>
export TMPDIR=/tmp
TAG=${1:-v2}
O=gpurun_out/r02prof_$TAG; mkdir -p $O
python bench.py --steps 5 --warmup 2 > $O/bench_c3.json 2> $O/bench.err; echo "c3 rc=$?"
python bench.py --config c2 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_c2.json 2>> $O/bench.err; echo "c2 rc=$?"
python bench.py --config c5 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_c5.json 2>> $O/bench.err; echo "c5 rc=$?"
python bench.py --config c4 --steps 2 --warmup 1 --no-cpu-baseline --no-host-path > $O/bench_c4.json 2>> $O/bench.err; echo "c4 rc=$?"
rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-host-path --no-symv-events --no-parity-check > $O/kt.log 2>&1; echo "kt rc=$?"
find /tmp/kt -name "*.db" | head -1 | xargs -r -I{} python tools/rocpd_summary.py {} > $O/kernel_stats_c3.txt 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F64 -d /tmp/pmc1 -o pmc -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-host-path --no-symv-events --no-parity-check > $O/pmc1.log 2>&1; echo "pmc1 rc=$?"
python tools/pmc_summary.py "/tmp/pmc1/**/*counter_collection*.csv" > $O/pmc_mfma_c3.txt 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE --kernel-include-regex "q2_apply|chase|symm_lower" -d /tmp/pmc2 -o pmc -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-host-path --no-symv-events --no-parity-check > $O/pmc2.log 2>&1; echo "pmc2 rc=$?"
python tools/pmc_summary.py "/tmp/pmc2/**/*counter_collection*.csv" > $O/pmc_fetch_c3.txt 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE --kernel-include-regex "q2_apply|chase|symm_lower" -d /tmp/pmc3 -o pmc -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-host-path --no-symv-events --no-parity-check > $O/pmc3.log 2>&1; echo "pmc3 rc=$?"
python tools/pmc_summary.py "/tmp/pmc3/**/*counter_collection*.csv" > $O/pmc_write_c3.txt 2>&1
du -sh $O; python - <<PY
import json
for c in ("c3","c2","c5","c4"):
    try:
        d=json.load(open("$O/bench_%s.json"%c)); print(c, round(d["ms_per_step"],1), round(d["value"]), {k.split(":")[-1]:round(v,4) for k,v in d["stage_seconds_per_step"].items() if v>1e-3})
    except Exception as e: print(c,"ERR",e)
PY
