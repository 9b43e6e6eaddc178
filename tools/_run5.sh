cd $GRAFT_REPO_ROOT
S=$(date +%s)
timeout -k 10 600 python bench.py --gpus 2 --rehearse-on-one-gpu > gpurun_out/bench_gpus2_rehearsal.json 2> gpurun_out/bench_gpus2_rehearsal.err; rc=$?
E=$(date +%s)
echo "rc=$rc wall=$((E-S)) s"
tail -5 gpurun_out/bench_gpus2_rehearsal.err
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/bench_gpus2_rehearsal.json").read().strip().splitlines()[-1])
    print({k: d.get(k) for k in ("value", "n_gpus", "ms_per_step", "scaling")})
    gp = d.get("grid_probe", {})
    print("grid_probe:", gp.get("error"), {m: {k: v.get(k) for k in ("ms_per_step", "parity_ok_all_ranks", "eigenvalues_identical_on_all_ranks", "error")} for m, v in gp.get("modes", {}).items()})
    print(gp.get("modes", {}).get("two_stage", {}).get("stage_seconds_per_step_rank0"))
except Exception as e:
    print("ERR", e)
PY
