# usage (on the GPU box): N=4096 LA=5120 bash tools/gap_trace.sh  -- kernel trace of the two-stage pieces and the idle share of stage 1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/gp
EK_SY2SB_LOOKAHEAD_MIN=${LA:-5120} timeout -k 10 300 rocprofv3 --kernel-trace -d /tmp/gp -o t --output-format csv -- python3 $R/tools/two_stage_timing.py ${N:-4096} 64 > /tmp/gp.log 2>&1
grep -a -v "simple_timer\|output_stream" /tmp/gp.log | tail -20; wc -l /tmp/gp/t_kernel_trace.csv
find /tmp/gp -name '*.csv'
F=$(find /tmp/gp -name '*kernel_trace.csv' | head -1)
head -2 $F
python3 $R/tools/gap_report.py $F "" symm_lower ${CALLS:-symm_lower,gemm_kernel_w8,hr_kernel,yred}
