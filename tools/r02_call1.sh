#!/bin/bash
# round-2 GPU call 1: full-size config tests, final-build kernel summary, MFMA counters
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/r02c1; mkdir -p $O
(time python -m pytest tests/test_gpu_configs.py -x -q --durations=10) > $O/configs_pytest.log 2>&1; echo "configs rc=$?" | tee -a $O/status.txt
rocprofv3 -L > /tmp/counters_list.txt 2>&1; grep -i "mfma\|^gpu" /tmp/counters_list.txt | cut -c1-160 | sort -u | head -80 > $O/counters_mfma.txt
python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_n16384.json 2> $O/bench_n16384.err; echo "bench rc=$?" | tee -a $O/status.txt
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --order 4096 --problem sep > $O/bench_n4096_sep.json 2>> $O/bench_n16384.err; echo "bench4096 rc=$?" | tee -a $O/status.txt
rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-symv-events --no-parity-check > $O/kt.log 2>&1; echo "kt rc=$?" | tee -a $O/status.txt
find /tmp/kt -name "*.db" | head -1 | xargs -r -I{} python tools/rocpd_summary.py {} > $O/kernel_stats_n16384.txt 2>&1
find /tmp/kt -name "*.db" | head -1 | xargs -r -I{} python tools/stage_breakdown.py {} > $O/stage_breakdown_n16384.txt 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F64 -d /tmp/pmc -o pmc -- python3 bench.py --steps 1 --warmup 0 --order 4096 --no-cpu-baseline --no-symv-events --no-parity-check > $O/pmc.log 2>&1; echo "pmc rc=$?" | tee -a $O/status.txt
python tools/pmc_summary.py "/tmp/pmc/**/*counter_collection*.csv" > $O/pmc_mfma_n4096.txt 2>&1
ls /tmp/pmc /tmp/pmc/* | head -20 >> $O/pmc.log
du -sh $O
