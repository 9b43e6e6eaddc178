# on the GPU box: duration of every SYMM launch of one headline solve by (T, nsplit), for forced numbers of splits
#   bash tools/symm_split_table.sh  -> gpurun_out/symm_split_table.csv   (T, nsplit, tiles per split, us)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "T,nsplit,us" > $R/gpurun_out/symm_split_table.csv
for k in 1 2 3 4 5 6 7 8 9 10 12 14 16; do
  rm -rf /tmp/sst
  EK_SY2SB_NSPLIT=$k timeout -k 10 200 rocprofv3 --kernel-trace -d /tmp/sst -o t --output-format csv -- python3 $R/bench.py --config c3 --steps 1 --warmup 0 --no-cpu-baseline --no-host-path --no-parity-check --no-symv-events --no-other-configs > /tmp/sst.log 2>&1 || { echo "run $k failed"; tail -5 /tmp/sst.log; exit 1; }
  F=$(find /tmp/sst -name '*kernel_trace.csv' | head -1)
  python3 - $F >> $R/gpurun_out/symm_split_table.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "symm_lower_kernel<false>" in r["Kernel_Name"]:
        print("%d,%d,%.2f" % (int(r["Grid_Size_X"]) // 256, int(r["Grid_Size_Y"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
PY
  echo "forced $k done"
done
