/*
 * scalapack_path.c -- CPU BASELINE / CROSS-CHECK (test infrastructure only; never linked
 * into the product).  Runs the reference's own solver path -- the same six ScaLAPACK calls
 * in the same order, uplo = 'L', NB = 64, 2-D block-cyclic on the near-square grid of
 * processes.f90:56-65 -- on the synthetic SPD pair of SURVEY.md 8(d), and prints one JSON line
 * with the reference's stage-event names and times:
 *
 *   generalized_to_standard.f90:24   PDPOTRF('L')          reduce_generalized:pdpotrf
 *   generalized_to_standard.f90:37   PDSYGST(1,'L')        reduce_generalized:pdsygst
 *   solver_scalapack_all.f90:59      PDSYTRD('L')          eigen_solver_scalapack_all:pdsytrd
 *   solver_scalapack_all.f90:75-78   allgather of d, e     eigen_solver_scalapack_all:gather1
 *   solver_scalapack_all.f90:96      PDSTEDC('I')          eigen_solver_scalapack_all:pdstedc
 *   solver_scalapack_all.f90:115     PDORMTR('L','L','N')  eigen_solver_scalapack_all:pdormtr
 *   generalized_to_standard.f90:103  PDTRTRS('L','T','N')  recovery_generalized
 *
 * The arithmetic is the third-party library the reference links (ScaLAPACK; here Intel
 * oneMKL from /opt/conda, MPICH), not a restatement: this is the closest thing to "the
 * reference's CPU path" that can be built without the reference's Fortran sources.
 *
 * usage: mpiexec -np P scalapack_path <n> <problem: 0 sep | 1 gep> [eigenvalue_file]
 */
#include <math.h>
#include <mpi.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

extern void Cblacs_pinfo(int *, int *);
extern void Cblacs_get(int, int, int *);
extern void Cblacs_gridinit(int *, const char *, int, int);
extern void Cblacs_gridinfo(int, int *, int *, int *, int *);
extern void Cblacs_gridexit(int);
extern int numroc_(const int *, const int *, const int *, const int *, const int *);
extern void descinit_(int *, const int *, const int *, const int *, const int *, const int *,
                      const int *, const int *, const int *, int *);
extern void pdpotrf_(const char *, const int *, double *, const int *, const int *, const int *, int *);
extern void pdsygst_(const int *, const char *, const int *, double *, const int *, const int *,
                     const int *, const double *, const int *, const int *, const int *, double *, int *);
extern void pdsytrd_(const char *, const int *, double *, const int *, const int *, const int *,
                     double *, double *, double *, double *, const int *, int *);
extern void pdlared1d_(const int *, const int *, const int *, const int *, const double *, double *,
                       double *, const int *);
extern void pdstedc_(const char *, const int *, double *, double *, double *, const int *, const int *,
                     const int *, double *, const int *, int *, const int *, int *);
extern void pdormtr_(const char *, const char *, const char *, const int *, const int *, const double *,
                     const int *, const int *, const int *, const double *, double *, const int *,
                     const int *, const int *, double *, const int *, int *);
extern void pdtrtrs_(const char *, const char *, const char *, const int *, const int *, const double *,
                     const int *, const int *, const int *, double *, const int *, const int *,
                     const int *, int *);

static uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ULL;
  uint64_t z = x;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
static double synth(int n, uint64_t seed, int i, int j, double inv) {
  const uint64_t hi = i > j ? i : j, lo = i > j ? j : i;
  const uint64_t r = splitmix64((seed << 40) + hi * (uint64_t)n + lo);
  const double u = (double)(r >> 11) * (1.0 / 4503599627370496.0) - 1.0;
  return u * inv + (i == j ? 2.0 : 0.0);
}

int main(int argc, char **argv) {
  MPI_Init(&argc, &argv);
  int rank, nprocs;
  Cblacs_pinfo(&rank, &nprocs);
  const int n = argc > 1 ? atoi(argv[1]) : 1024;
  const int problem = argc > 2 ? atoi(argv[2]) : 1;
  const char *evfile = argc > 3 ? argv[3] : NULL;
  int nb = 64;                                   /* global_variables.f90:5 */
  /* layout_procs, processes.f90:56-65 */
  int prow = (int)sqrt((double)(nprocs + 1));
  while (nprocs % prow != 0) --prow;
  const int pcol = nprocs / prow;
  const int maxnb = (n / prow < n / pcol ? n / prow : n / pcol);   /* distribute_matrix.f90:114-120 */
  if (nb > (maxnb > 1 ? maxnb : 1)) nb = maxnb > 1 ? maxnb : 1;
  int ctxt, myrow, mycol, izero = 0, ione = 1, info;
  Cblacs_get(-1, 0, &ctxt);
  Cblacs_gridinit(&ctxt, "R", prow, pcol);       /* row-major, processes.f90:23 */
  Cblacs_gridinfo(ctxt, &prow, (int *)&pcol, &myrow, &mycol);
  const int lr = numroc_(&n, &nb, &myrow, &izero, &prow), lc = numroc_(&n, &nb, &mycol, &izero, &pcol);
  const int lld = lr > 1 ? lr : 1;
  int descA[9], descB[9], descZ[9];
  descinit_(descA, &n, &n, &nb, &nb, &izero, &izero, &ctxt, &lld, &info);
  descinit_(descB, &n, &n, &nb, &nb, &izero, &izero, &ctxt, &lld, &info);
  descinit_(descZ, &n, &n, &nb, &nb, &izero, &izero, &ctxt, &lld, &info);
  const size_t loc = (size_t)lld * (lc > 1 ? lc : 1);
  double *A = calloc(loc, 8), *B = problem ? calloc(loc, 8) : NULL, *Z = calloc(loc, 8);
  const double inv = 1.0 / sqrt((double)n);
  for (int jl = 0; jl < lc; ++jl) {
    const int j = ((jl / nb) * pcol + mycol) * nb + jl % nb;
    for (int il = 0; il < lr; ++il) {
      const int i = ((il / nb) * prow + myrow) * nb + il % nb;
      A[il + (size_t)jl * lld] = synth(n, 1, i, j, inv);
      if (problem) B[il + (size_t)jl * lld] = synth(n, 2, i, j, inv);
    }
  }
  double t[8] = {0}, t0, scale = 1.0;
  {   /* untimed warm-up: the first ScaLAPACK/BLACS call pays library initialisation */
    double *S = malloc(loc * 8);
    memcpy(S, problem ? B : A, loc * 8);
    pdpotrf_("L", &n, S, &ione, &ione, descA, &info);
    free(S);
  }
  MPI_Barrier(MPI_COMM_WORLD);
  const double tstart = MPI_Wtime();
  if (problem) {
    t0 = MPI_Wtime();
    pdpotrf_("L", &n, B, &ione, &ione, descB, &info);
    if (info) { if (!rank) fprintf(stderr, "info(pdpotrf): %d\n", info); MPI_Abort(MPI_COMM_WORLD, info); }
    t[0] = MPI_Wtime() - t0; t0 = MPI_Wtime();
    pdsygst_(&ione, "L", &n, A, &ione, &ione, descA, B, &ione, &ione, descB, &scale, &info);
    if (info) { if (!rank) fprintf(stderr, "info(pdsygst): %d\n", info); MPI_Abort(MPI_COMM_WORLD, info); }
    t[1] = MPI_Wtime() - t0;
  }
  /* pdsytrd: d, e, tau distributed like the columns (solver_scalapack_all.f90:43-46) */
  const int dsz = lc > 1 ? lc : 1;
  double *dl = calloc(dsz, 8), *el = calloc(dsz, 8), *tau = calloc(dsz, 8);
  int lwork = nb * (lld + 1) > 3 * nb ? nb * (lld + 1) : 3 * nb;
  double *work = malloc((size_t)lwork * 8);
  t0 = MPI_Wtime();
  pdsytrd_("L", &n, A, &ione, &ione, descA, dl, el, tau, work, &lwork, &info);
  t[2] = MPI_Wtime() - t0; t0 = MPI_Wtime();
  free(work);
  double *d = calloc(n, 8), *e = calloc(n, 8);
  {
    lwork = n;
    work = malloc((size_t)n * 8);
    pdlared1d_(&n, &ione, &ione, descA, dl, d, work, &lwork);
    pdlared1d_(&n, &ione, &ione, descA, el, e, work, &lwork);
    free(work);
  }
  t[3] = MPI_Wtime() - t0; t0 = MPI_Wtime();
  lwork = 6 * n + 2 * lld * (lc > 1 ? lc : 1);     /* solver_scalapack_all.f90:84-86 */
  int liwork = 2 + 7 * n + 8 * pcol;
  work = malloc((size_t)lwork * 8);
  int *iwork = malloc((size_t)liwork * 4);
  pdstedc_("I", &n, d, e, Z, &ione, &ione, descZ, work, &lwork, iwork, &liwork, &info);
  if (info && !rank) fprintf(stderr, "info(pdstedc): %d\n", info);
  free(work); free(iwork);
  t[4] = MPI_Wtime() - t0; t0 = MPI_Wtime();
  {
    double wq; lwork = -1;
    pdormtr_("L", "L", "N", &n, &n, A, &ione, &ione, descA, tau, Z, &ione, &ione, descZ, &wq, &lwork, &info);
    lwork = (int)wq + 1;
    work = malloc((size_t)lwork * 8);
    pdormtr_("L", "L", "N", &n, &n, A, &ione, &ione, descA, tau, Z, &ione, &ione, descZ, work, &lwork, &info);
    free(work);
  }
  t[5] = MPI_Wtime() - t0;
  if (problem) {
    t0 = MPI_Wtime();
    pdtrtrs_("L", "T", "N", &n, &n, B, &ione, &ione, descB, Z, &ione, &ione, descZ, &info);
    if (info) { if (!rank) fprintf(stderr, "info(pdtrtrs): %d\n", info); MPI_Abort(MPI_COMM_WORLD, info); }
    t[6] = MPI_Wtime() - t0;
  }
  const double total = MPI_Wtime() - tstart;
  double tmax[8], totmax;
  MPI_Reduce(t, tmax, 8, MPI_DOUBLE, MPI_MAX, 0, MPI_COMM_WORLD);
  MPI_Reduce(&total, &totmax, 1, MPI_DOUBLE, MPI_MAX, 0, MPI_COMM_WORLD);
  if (!rank) {
    if (evfile) {
      FILE *f = fopen(evfile, "w");
      for (int i = 0; i < n; ++i) fprintf(f, "%.17e\n", d[i]);
      fclose(f);
    }
    printf("{\"n\": %d, \"problem\": %d, \"np\": %d, \"grid\": [%d, %d], \"nb\": %d, \"seconds\": %.6f, "
           "\"stages\": {\"reduce_generalized:pdpotrf\": %.6f, \"reduce_generalized:pdsygst\": %.6f, "
           "\"eigen_solver_scalapack_all:pdsytrd\": %.6f, \"eigen_solver_scalapack_all:gather1\": %.6f, "
           "\"eigen_solver_scalapack_all:pdstedc\": %.6f, \"eigen_solver_scalapack_all:pdormtr\": %.6f, "
           "\"recovery_generalized\": %.6f}, \"w_min\": %.17e, \"w_max\": %.17e}\n",
           n, problem, nprocs, prow, pcol, nb, totmax, tmax[0], tmax[1], tmax[2], tmax[3], tmax[4], tmax[5],
           tmax[6], d[0], d[n - 1]);
  }
  Cblacs_gridexit(ctxt);
  MPI_Finalize();
  return 0;
}
