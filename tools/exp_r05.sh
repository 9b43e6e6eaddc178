set -o pipefail
EK_SY2SB_PAIR_MIN=1 EK_SY2SB_LOOKAHEAD_MIN=300 python -m pytest tests/test_gpu_two_stage.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -5
EK_SY2SB_PAIR_MIN=1 python -m pytest tests/test_gpu_two_stage.py -m gpu -x -q 2>&1 | tail -3
for PM in 0 5120 2048 8192; do
EK_SY2SB_PAIR_MIN=$PM python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-path --no-other-configs > /tmp/b.json 2>/tmp/b.err
python - <<PY
import json
d=json.load(open("/tmp/b.json"))
print("pair_min", $PM, round(d["ms_per_step"],1), round(d["stage_seconds_per_step"]["eigen_solver_scalapack_all:pdsytrd"],4), d["parity"]["residual_norm_max"], d["parity"]["orthogonality"])
PY
done
