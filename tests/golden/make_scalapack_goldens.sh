#!/bin/sh
# Regenerates the ScaLAPACK golden eigenvalues (oneMKL ScaLAPACK from /opt/conda, MPICH):
# the reference's own call sequence (oracle/scalapack_path.c) on the synthetic inputs of
# SURVEY.md 8(d), on the reference's 2x2 grid.  Run from the repo root after `make -C oracle`.
set -e
for spec in "256 1 gep" "256 0 sep" "1000 1 gep"; do
  set -- $spec
  /opt/conda/bin/mpiexec -np 4 oracle/scalapack_path $1 $2 tests/golden/scalapack_synth_$3_n$1_np4.txt > /dev/null
done
/opt/conda/bin/mpiexec -np 1 oracle/scalapack_path 256 1 tests/golden/scalapack_synth_gep_n256_np1.txt > /dev/null
# full-size BASELINE configurations (C2: N=4096 standard; C3/C5: N=16384 generalized), the
# reference's 2x4 grid for 8 ranks (processes.f90:56-65); 3 s and 5 min on 8 cores
/opt/conda/bin/mpiexec -np 8 oracle/scalapack_path 4096 0 tests/golden/scalapack_synth_sep_n4096_np8.txt > /dev/null
/opt/conda/bin/mpiexec -np 8 oracle/scalapack_path 16384 1 tests/golden/scalapack_synth_gep_n16384_np8.txt > /dev/null
# C4: N=32768 generalized on the same 2x4 grid: 38 min on 8 cores (2268 s of ScaLAPACK: pdsytrd 1235 s), ~45 GiB
/opt/conda/bin/mpiexec -np 8 oracle/scalapack_path 32768 1 tests/golden/scalapack_synth_gep_n32768_np8.txt > /dev/null
