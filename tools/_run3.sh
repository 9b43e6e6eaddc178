set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out
for LV in 2 3 4; do timeout -k 10 300 python3 $R/tools/dc_team_cell.py 32768 8 $LV 2 | tail -1; done
rm -rf /tmp/ks
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d /tmp/ks -o ks -- python3 $R/tools/dc_team_cell.py 32768 8 2 1 > /tmp/ks.log 2>&1
find /tmp/ks -name "*.db" | head -1 | xargs -r -I{} python3 $R/tools/rocpd_summary.py {} > $R/gpurun_out/dc_team_cell_32k.txt 2>&1
head -1 $R/gpurun_out/dc_team_cell_32k.txt | cut -c1-160; grep -E "dc_|gather_col|ormtr|tfactor|larft|record" $R/gpurun_out/dc_team_cell_32k.txt | cut -c1-160 | head -40
