R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
cat > /tmp/d2b.py <<'PY'
import ctypes, os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from eigenkernel_amd import solver
lib = solver.load_library(); assert lib.ek_hip_init(0) == 0
sec = ctypes.c_double(0)
n, P = int(sys.argv[1]), int(sys.argv[2])
assert lib.ek_hip_debug_sy2sb_team_timing(n, P, 1, ctypes.byref(sec)) == 0
print("team of", P, "n", n, sec.value)
PY
rm -rf /tmp/ks
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d /tmp/ks -o ks -- python3 /tmp/d2b.py 32768 8 > /tmp/ks.log 2>&1
tail -2 /tmp/ks.log
find /tmp/ks -name "*.db" | head -1 | xargs -r -I{} python3 $R/tools/rocpd_summary.py {} > $R/gpurun_out/d2b_team8_32k.txt 2>&1
head -24 $R/gpurun_out/d2b_team8_32k.txt | cut -c1-165
