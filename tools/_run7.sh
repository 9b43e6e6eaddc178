set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_dist.py -x -q -k "potrf or three_processes or four_processes or fortran" > gpurun_out/t7.log 2>&1 || { tail -60 gpurun_out/t7.log; exit 1; }
tail -3 gpurun_out/t7.log
EK_TEAM_TWO_STAGE_ONLY=1 timeout -k 10 500 python tools/team_timing.py 16384 8 2>&1 | grep -E "potrf|per-rank"
EK_TEAM_TWO_STAGE_ONLY=1 timeout -k 10 600 python tools/team_timing.py 32768 8 2>&1 | grep -E "potrf|per-rank"
