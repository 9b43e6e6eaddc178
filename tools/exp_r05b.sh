python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -2
run() {
  env "$@" python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-path --no-other-configs --no-parity-check > /tmp/b.json 2>/tmp/b.err || tail -5 /tmp/b.err
  python - "$*" <<'PY'
import json,sys
d=json.load(open("/tmp/b.json"))
print(sys.argv[1], round(d["ms_per_step"],1), round(d["stage_seconds_per_step"]["eigen_solver_scalapack_all:pdsytrd"],4))
PY
}
run A=1
python tools/hard_inputs_timing.py 16384 2>&1 | grep -E "dense|band65|low_rank"
