#!/bin/bash
export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_gpu_blocks.py -x -q 2>&1 | tail -2
python - <<'PY'
import numpy as np, sys
sys.path.insert(0, ".")
from eigenkernel_amd import solver
solver.load_library().ek_hip_init(0)
rng = np.random.default_rng(0)
bad = 0
for (m, n, k, lower) in [(300, 300, 128, True), (257, 129, 64, False), (1000, 1000, 128, True), (640, 640, 200, True), (130, 700, 33, False), (512, 512, 256, True), (129, 129, 128, True)]:
    A = rng.standard_normal((m, k)); B = rng.standard_normal((n, k)); C = rng.standard_normal((m, n))
    got = solver.dgemm(False, True, -1.0, A, B, 1.0, C, lower_only=lower)
    ref = C - A @ B.T
    if lower:
        tm = np.add.outer(np.arange(m) // 128, -(np.arange(n) // 128)) >= 0     # tiles touching the lower triangle
        err = np.abs((got - ref)[tm]).max(); keep = np.abs((got - C)[~tm]).max() if (~tm).any() else 0.0
    else:
        err = np.abs(got - ref).max(); keep = 0.0
    ok = err <= 1e-12 * k and keep == 0.0
    bad += not ok
    print("rank-k", m, n, k, lower, "err %.2e untouched %.1e %s" % (err, keep, "ok" if ok else "BAD"))
print("BAD", bad)
PY
timeout -k 10 300 python tools/gemm_shapes.py 16384 2>&1 | head -5
for n in 8192 16384; do timeout -k 10 200 python tools/two_stage_timing.py $n 64 2>&1 | tail -1; done
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-host-path | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],1), {k.split(':')[-1]:round(v,4) for k,v in d['stage_seconds_per_step'].items() if v>1e-3})"
EK_HIP_TWO_STAGE_MIN=0 python bench.py --order 8192 --steps 2 --warmup 1 --no-cpu-baseline --no-host-path --no-symv-events | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('8192 one-stage', round(d['ms_per_step'],1), {k.split(':')[-1]:round(v,4) for k,v in d['stage_seconds_per_step'].items() if v>1e-3})"
