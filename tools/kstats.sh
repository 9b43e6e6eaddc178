# usage (on the GPU box): CFG=c3 PAT="dc_|gather|copy" bash tools/kstats.sh -- per-kernel totals of one configuration
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/ks
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d /tmp/ks -o ks -- python3 $R/bench.py --config ${CFG:-c3} --steps 2 --warmup 1 --no-cpu-baseline --no-host-path --no-symv-events --no-parity-check > /tmp/ks.log 2>&1
find /tmp/ks -name "*.db" | head -1 | xargs -r -I{} python3 $R/tools/rocpd_summary.py {} > /tmp/ks.txt 2>&1
head -1 /tmp/ks.txt | cut -c1-150; grep -E "${PAT:-.}" /tmp/ks.txt | cut -c1-150 | head -${TOP:-40}
