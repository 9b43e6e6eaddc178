/* ek_ranks_demo.c -- a plain-C, MPI-shaped host for libek_hip.so (nothing but include/ek_hip.h).
 *
 *   ek_ranks_demo [n=1500] [nprow=2] [npcol=2] [problem=1]
 *
 * What an MPI host of the reference does around eigen_solver (main.f90:84-104), with fork() and a
 * shared-memory segment standing in for mpirun and MPI_Allgatherv, so that it runs on a box with
 * one GPU and no MPI: nprow*npcol ranks (processes) bind to GPU 0, build their block-cyclic
 * pieces of the synthetic pair of SURVEY.md 8(d) (setup_distributed_matrix +
 * distribute_global_sparse_matrix of the reference), lend the library the all-gather hook, attach
 * the host communicator, and call ek_hip_solve -- the reference's
 * own data contract: pieces of A and B in; eigenvalues on every rank, pieces of Z, of the
 * reflectors and of L out.  Rank 0 then collects the pieces of Z and checks
 * ||A z - lambda B z|| for a few eigenpairs and that every rank holds the same eigenvalues.
 * On a node with MPI and RCCL the hook is MPI_Allgatherv and the communicator comes from
 * ek_hip_comm_unique_id / ek_hip_comm_init.
 */
#define _GNU_SOURCE
#include "../include/ek_hip.h"

#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/wait.h>
#include <unistd.h>

typedef struct {
  pthread_barrier_t bar;
  int nranks;
  int failed;
  double w0[8];            /* a few eigenvalues of rank 0, for the cross-rank check */
  double data[];           /* exchange area */
} shared_t;

static shared_t *g_sh;
static int g_rank;

/* MPI_Allgatherv on doubles over the ranks, through the shared segment */
static int hook_allgatherv(const double *send, long long count, double *recv, const long long *counts,
                           const long long *displs, void *user) {
  (void)user;
  memcpy(g_sh->data + displs[g_rank], send, (size_t)count * sizeof(double));
  pthread_barrier_wait(&g_sh->bar);
  const long long tot = displs[g_sh->nranks - 1] + counts[g_sh->nranks - 1];
  memcpy(recv, g_sh->data, (size_t)tot * sizeof(double));
  pthread_barrier_wait(&g_sh->bar);
  return 0;
}

/* the synthetic generator of SURVEY.md 8(d) */
static uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  uint64_t z = x;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
static double synth(int n, uint64_t seed, int i, int j) {
  if (i < j) { int t = i; i = j; j = t; }
  const double u = (double)(splitmix64((seed << 40) + (uint64_t)i * (uint64_t)n + (uint64_t)j) >> 11) * 0x1p-52 - 1.0;
  return u / sqrt((double)n) + (i == j ? 2.0 : 0.0);
}
static int numroc(int n, int nb, int me, int np) {
  const int nblocks = n / nb;
  int num = (nblocks / np) * nb;
  const int extra = nblocks % np;
  if (me < extra) num += nb;
  else if (me == extra) num += n % nb;
  return num;
}
static int l2g(int l, int nb, int me, int np) { return ((l / nb) * np + me) * nb + l % nb; }

static int run_rank(int rank, int n, int nprow, int npcol, int problem) {
  g_rank = rank;
  const int P = nprow * npcol, myrow = rank / npcol, mycol = rank % npcol, nb = 64;
  int rc = ek_hip_init(0);
  if (rc) { fprintf(stderr, "[%d] ek_hip_init: %d\n", rank, rc); return 2; }
  rc = ek_hip_set_allgatherv(hook_allgatherv, NULL);
  if (!rc) rc = ek_hip_comm_attach_host(P, rank);
  if (rc) { fprintf(stderr, "[%d] communicator: %d\n", rank, rc); return 2; }
  const int nr = numroc(n, nb, myrow, nprow), nc = numroc(n, nb, mycol, npcol), lld = nr > 1 ? nr : 1;
  const int desc[9] = {1, 0, n, n, nb, nb, 0, 0, lld};
  double *A = calloc((size_t)lld * (nc > 0 ? nc : 1), sizeof(double));
  double *B = calloc((size_t)lld * (nc > 0 ? nc : 1), sizeof(double));
  double *Z = calloc((size_t)lld * (nc > 0 ? nc : 1), sizeof(double));
  double *w = calloc((size_t)n, sizeof(double));
  for (int lc = 0; lc < nc; ++lc)
    for (int lr = 0; lr < nr; ++lr) {
      const int gi = l2g(lr, nb, myrow, nprow), gj = l2g(lc, nb, mycol, npcol);
      A[(size_t)lr + (size_t)lc * lld] = synth(n, 1, gi, gj);
      B[(size_t)lr + (size_t)lc * lld] = synth(n, 2, gi, gj);
    }
  double stage[EK_HIP_N_STAGES] = {0};
  const int info = ek_hip_solve(problem, n, n, A, desc, problem ? B : NULL, problem ? desc : NULL, w, Z, desc, nprow,
                                npcol, myrow, mycol, stage, EK_HIP_N_STAGES);
  if (info) { fprintf(stderr, "[%d] ek_hip_solve info = %d\n", rank, info); g_sh->failed = 1; }
  /* every rank holds the same eigenvalues */
  if (rank == 0) for (int k = 0; k < 8; ++k) g_sh->w0[k] = w[(size_t)k * (n - 1) / 7];
  pthread_barrier_wait(&g_sh->bar);
  for (int k = 0; k < 8; ++k)
    if (g_sh->w0[k] != w[(size_t)k * (n - 1) / 7]) { fprintf(stderr, "[%d] eigenvalue %d differs from rank 0's\n", rank, k); g_sh->failed = 1; }
  pthread_barrier_wait(&g_sh->bar);
  /* rank 0 collects Z (one all-gather through the same hook) and checks a few eigenpairs */
  {
    long long counts[64], displs[64], tot = 0;
    for (int r = 0; r < P; ++r) {
      counts[r] = (long long)numroc(n, nb, r / npcol, nprow) * numroc(n, nb, r % npcol, npcol);
      displs[r] = tot; tot += counts[r];
    }
    double *packed = malloc((size_t)(counts[rank] > 0 ? counts[rank] : 1) * sizeof(double));
    for (int lc = 0; lc < nc; ++lc) memcpy(packed + (size_t)lc * nr, Z + (size_t)lc * lld, (size_t)nr * sizeof(double));
    double *all = malloc((size_t)tot * sizeof(double));
    hook_allgatherv(packed, counts[rank], all, counts, displs, NULL);
    if (rank == 0 && !g_sh->failed) {
      double worst = 0.0;
      double *z = malloc((size_t)n * sizeof(double)), *az = malloc((size_t)n * sizeof(double)), *bz = malloc((size_t)n * sizeof(double));
      for (int t = 0; t < 6; ++t) {
        const int col = (int)((long long)t * (n - 1) / 5);
        const int pc = (col / nb) % npcol, lc = (col / (nb * npcol)) * nb + col % nb;
        for (int pr = 0; pr < nprow; ++pr) {
          const int r = pr * npcol + pc, nrr = numroc(n, nb, pr, nprow);
          for (int lr = 0; lr < nrr; ++lr) z[l2g(lr, nb, pr, nprow)] = all[displs[r] + (size_t)lr + (size_t)lc * nrr];
        }
        double res = 0.0;
        for (int i = 0; i < n; ++i) {
          double a = 0.0, b = 0.0;
          for (int j = 0; j < n; ++j) { a += synth(n, 1, i, j) * z[j]; b += (problem ? synth(n, 2, i, j) : (i == j)) * z[j]; }
          az[i] = a; bz[i] = b;
          const double d = fabs(a - w[col] * b);
          if (d > res) res = d;
        }
        if (res > worst) worst = res;
      }
      printf("n=%d grid %dx%d (%d processes on GPU 0), %s exchange, %s problem: solve %.3f s "
             "(potrf %.3f sygst %.3f sytrd %.3f gather %.3f stedc %.3f ormtr %.3f trtrs %.3f), "
             "lambda_min %.12f lambda_max %.12f, max |A z - lambda B z| over 6 eigenpairs %.2e\n",
             n, nprow, npcol, P, "host-hook", problem ? "generalized" : "standard",
             stage[0] + stage[1] + stage[2] + stage[3] + stage[4] + stage[5] + stage[6], stage[0], stage[1], stage[2],
             stage[3], stage[4], stage[5], stage[6], w[0], w[n - 1], worst);
      fflush(stdout);
      if (!(worst <= 1e-11)) g_sh->failed = 1;
      free(z); free(az); free(bz);
    }
    free(packed); free(all);
  }
  pthread_barrier_wait(&g_sh->bar);
  ek_hip_comm_destroy();
  const int failed = g_sh->failed;
  free(A); free(B); free(Z); free(w);
  return failed ? 1 : 0;
}

int main(int argc, char **argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 1500;
  const int nprow = argc > 2 ? atoi(argv[2]) : 2, npcol = argc > 3 ? atoi(argv[3]) : 2;
  const int problem = argc > 4 ? atoi(argv[4]) : 1;
  const int P = nprow * npcol;
  if (n < 2 || P < 1 || P > 16) { fprintf(stderr, "usage: ek_ranks_demo [n] [nprow] [npcol] [problem]\n"); return 64; }
  /* the largest exchange is an all-gather of a matrix in the library's padded layout (ld <= n + 255) */
  const size_t bytes = sizeof(shared_t) + ((size_t)(n + 256) * (n + 256)) * sizeof(double);
  g_sh = mmap(NULL, bytes, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
  if (g_sh == MAP_FAILED) { perror("mmap"); return 1; }
  pthread_barrierattr_t ba;
  pthread_barrierattr_init(&ba);
  pthread_barrierattr_setpshared(&ba, PTHREAD_PROCESS_SHARED);
  pthread_barrier_init(&g_sh->bar, &ba, (unsigned)P);
  g_sh->nranks = P; g_sh->failed = 0;
  fflush(stdout);
  pid_t pids[16];
  for (int r = 0; r < P; ++r) {          /* fork BEFORE anything touches the GPU */
    pids[r] = fork();
    if (pids[r] == 0) _exit(run_rank(r, n, nprow, npcol, problem));
  }
  int rc = 0;
  for (int r = 0; r < P; ++r) {
    int st = 0;
    waitpid(pids[r], &st, 0);
    if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) rc = 1;
  }
  return rc;
}
