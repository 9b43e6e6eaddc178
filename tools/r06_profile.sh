#!/bin/bash
# round-6 profile artefacts: the default bench line (headline + other_configs + cpu baseline), rocprofv3 kernel
# summary of the headline command, PMC passes (MFMA instruction / busy counters; HBM traffic of q2_apply_nb_kernel).
# usage (on the GPU box): bash tools/r06_profile.sh <tag> ; then locally: python tools/make_traffic_record.py <tag>
export TMPDIR=/tmp
TAG=${1:-v1}
O=gpurun_out/r06prof_$TAG; mkdir -p $O
B="python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-host-path --no-other-configs --no-symv-events --no-parity-check"
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_c3.json 2> $O/bench.err; echo "bench rc=$?"
rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-host-path --no-other-configs --no-symv-events --no-parity-check > $O/kt.log 2>&1; echo "kt rc=$?"
find /tmp/kt -name "*.db" | head -1 | xargs -r -I{} python tools/rocpd_summary.py {} > $O/kernel_stats_c3.txt 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F64 -d /tmp/pmc1 -o pmc -- $B > $O/pmc1.log 2>&1; echo "pmc1 rc=$?"
python tools/pmc_summary.py "/tmp/pmc1/**/*counter_collection*.csv" > $O/pmc_mfma_c3.txt 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE --kernel-include-regex "q2_apply|chase|symm_lower" -d /tmp/pmc2 -o pmc -- $B > $O/pmc2.log 2>&1; echo "pmc2 rc=$?"
python tools/pmc_summary.py "/tmp/pmc2/**/*counter_collection*.csv" > $O/pmc_fetch_c3.txt 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE --kernel-include-regex "q2_apply|chase|symm_lower" -d /tmp/pmc3 -o pmc -- $B > $O/pmc3.log 2>&1; echo "pmc3 rc=$?"
python tools/pmc_summary.py "/tmp/pmc3/**/*counter_collection*.csv" > $O/pmc_write_c3.txt 2>&1
sha256sum eigenkernel_amd/csrc/ek_sb2st.hip > $O/source_sha256.txt
du -sh $O; python - <<PY
import json
try:
    d=json.load(open("$O/bench_c3.json")); print("c3", round(d["ms_per_step"],1), round(d["value"]), {k.split(":")[-1]:round(v,4) for k,v in d["stage_seconds_per_step"].items() if v>1e-3})
    for c,v in d.get("other_configs",{}).items(): print(c, v.get("ms_per_step"), v.get("parity",{}).get("ok"), v.get("error"))
    print("incl copies", d.get("value_incl_copies")); print("cpu", {k:v for k,v in d.get("cpu_baseline",{}).items() if k in ("value","seconds","cores","error")})
except Exception as e: print("ERR",e)
PY
