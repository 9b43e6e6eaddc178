run() { # name env...
  env "$@" python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-path --no-other-configs --no-parity-check > /tmp/b.json 2>/tmp/b.err
  python - "$*" <<'PY'
import json,sys
d=json.load(open("/tmp/b.json"))
print(sys.argv[1], round(d["ms_per_step"],1), round(d["stage_seconds_per_step"]["eigen_solver_scalapack_all:pdsytrd"],4))
PY
}
run EK_SY2SB_STAGED_MAX=8192
run EK_SY2SB_STAGED_MAX=0
run EK_SY2SB_STAGED_MAX=4096
run EK_SY2SB_STAGED_MAX=12288
run EK_SY2SB_STAGED_MAX=20000
run EK_SY2SB_LOOKAHEAD_MIN=3072 EK_SY2SB_PAIR_MIN=3072
run EK_SY2SB_LOOKAHEAD_MIN=4096 EK_SY2SB_PAIR_MIN=4096
run EK_SY2SB_LOOKAHEAD_MIN=7168 EK_SY2SB_PAIR_MIN=7168
run EK_SY2SB_LOOKAHEAD_MIN=4096 EK_SY2SB_PAIR_MIN=6144
