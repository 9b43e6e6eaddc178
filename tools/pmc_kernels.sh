#!/bin/bash
# SQ counters of selected kernels of the headline solve (two passes of four counters):
#   bash tools/pmc_kernels.sh "symm_lower|gemm_kernel" [tag]    -> gpurun_out/pmck_<tag>/{a,b}.txt
export TMPDIR=/tmp
RX=${1:-symm_lower}
O=$PWD/gpurun_out/pmck_${2:-x}; mkdir -p $O
B="python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-host-path --no-other-configs --no-symv-events --no-parity-check"
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-include-regex "$RX" -d /tmp/pk1 -o pmc -- $B > $O/a.log 2>&1; echo "a rc=$?"
python tools/pmc_summary.py "/tmp/pk1/**/*counter_collection*.csv" > $O/a.txt 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-include-regex "$RX" -d /tmp/pk2 -o pmc -- $B > $O/b.log 2>&1; echo "b rc=$?"
python tools/pmc_summary.py "/tmp/pk2/**/*counter_collection*.csv" > $O/b.txt 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F64 SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM --kernel-include-regex "$RX" -d /tmp/pk3 -o pmc -- $B > $O/c.log 2>&1; echo "c rc=$?"
python tools/pmc_summary.py "/tmp/pk3/**/*counter_collection*.csv" > $O/c.txt 2>&1
cat $O/a.txt $O/b.txt $O/c.txt
