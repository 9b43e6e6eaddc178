set -e
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_dist.py -x -q -k "stedc_team or three_processes or four_processes" > gpurun_out/t4.log 2>&1 || { tail -60 gpurun_out/t4.log; exit 1; }
tail -3 gpurun_out/t4.log
for LV in 2 3; do timeout -k 10 300 python3 $R/tools/dc_team_cell.py 32768 8 $LV 2 | tail -1; done
timeout -k 10 300 python3 $R/tools/dc_team_cell.py 16384 8 2 2 | tail -1
