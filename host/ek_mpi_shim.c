/* ek_mpi_shim.c -- the handful of MPI calls host/eigenkernel_hip_mpi_app.f90 makes, as plain C functions.
 *
 * The reference's hosts say `use mpi` (main.f90:2, processes.f90:2).  This image has MPICH's C library and
 * headers (/opt/conda) but no `mpi.mod` that AMD flang can read, so the Fortran program reaches MPI through
 * these bind(C) wrappers; with an MPI Fortran module at hand they are replaced one for one by the calls
 * INTEGRATION.md 1c / 1d show (mpi_allgatherv, mpi_bcast, ...).  ekm_allgatherv has exactly the signature of
 * ek_hip_allgatherv_fn (include/ek_hip.h): it is the exchange hook the library borrows from the host.
 */
#include <mpi.h>
#include <stdlib.h>

int ekm_init(int *rank, int *nprocs) {
  int rc = MPI_Init(NULL, NULL);
  if (rc != MPI_SUCCESS) return rc;
  MPI_Comm_rank(MPI_COMM_WORLD, rank);
  MPI_Comm_size(MPI_COMM_WORLD, nprocs);
  return 0;
}

int ekm_finalize(void) { return MPI_Finalize(); }
int ekm_barrier(void) { return MPI_Barrier(MPI_COMM_WORLD); }
void ekm_abort(int code) { MPI_Abort(MPI_COMM_WORLD, code); }
double ekm_wtime(void) { return MPI_Wtime(); }

/* MPI_Allgatherv on doubles over MPI_COMM_WORLD (= row-major BLACS order, processes.f90:23).  MPI counts
 * are ints: larger pieces travel in slabs. */
int ekm_allgatherv(const double *send, long long count, double *recv, const long long *counts,
                   const long long *displs, void *user) {
  (void)user;
  int np = 1, me = 0;
  MPI_Comm_size(MPI_COMM_WORLD, &np);
  MPI_Comm_rank(MPI_COMM_WORLD, &me);
  const long long slab = 1ll << 27;   /* doubles per rank and round */
  long long maxc = 0;
  for (int r = 0; r < np; ++r) if (counts[r] > maxc) maxc = counts[r];
  /* a rank whose own count disagrees with the table must not leave the collective alone: agree first */
  int bad = (counts[me] != count) ? 1 : 0, anybad = 0;
  MPI_Allreduce(&bad, &anybad, 1, MPI_INT, MPI_MAX, MPI_COMM_WORLD);
  if (anybad) return -1;
  int *cnt = (int *)malloc(sizeof(int) * (size_t)np), *dsp = (int *)malloc(sizeof(int) * (size_t)np);
  if (!cnt || !dsp) { free(cnt); free(dsp); return -2; }
  int rc = 0;
  if (maxc <= slab && displs[np - 1] + counts[np - 1] < (1ll << 31)) {
    for (int r = 0; r < np; ++r) { cnt[r] = (int)counts[r]; dsp[r] = (int)displs[r]; }
    rc = MPI_Allgatherv(send, (int)count, MPI_DOUBLE, recv, cnt, dsp, MPI_DOUBLE, MPI_COMM_WORLD);
  } else {
    /* round q moves doubles [q*slab, (q+1)*slab) of every rank's piece: a broadcast per rank keeps the
     * displacements out of the int range */
    for (long long off = 0; off < maxc && rc == 0; off += slab)
      for (int r = 0; r < np && rc == 0; ++r) {
        long long c = counts[r] - off;
        if (c <= 0) continue;
        if (c > slab) c = slab;
        double *dst = recv + displs[r] + off;
        if (r == me) for (long long i = 0; i < c; ++i) dst[i] = send[off + i];
        rc = MPI_Bcast(dst, (int)c, MPI_DOUBLE, r, MPI_COMM_WORLD);
      }
  }
  free(cnt); free(dsp);
  return rc == MPI_SUCCESS ? 0 : rc;
}

int ekm_bcast_bytes(void *buf, int nbytes, int root) { return MPI_Bcast(buf, nbytes, MPI_BYTE, root, MPI_COMM_WORLD); }
int ekm_bcast_doubles(double *buf, int n, int root) { return MPI_Bcast(buf, n, MPI_DOUBLE, root, MPI_COMM_WORLD); }

int ekm_max_int(int v) {
  int out = v;
  MPI_Allreduce(&v, &out, 1, MPI_INT, MPI_MAX, MPI_COMM_WORLD);
  return out;
}
int ekm_min_int(int v) {
  int out = v;
  MPI_Allreduce(&v, &out, 1, MPI_INT, MPI_MIN, MPI_COMM_WORLD);
  return out;
}
double ekm_max_double(double v) {
  double out = v;
  MPI_Allreduce(&v, &out, 1, MPI_DOUBLE, MPI_MAX, MPI_COMM_WORLD);
  return out;
}
