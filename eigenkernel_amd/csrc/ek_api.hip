// ek_api.hip -- the C-ABI of libek_hip.so (declared in include/ek_hip.h).
//
// Host side of the drop-in boundary: validates arguments the way a LAPACK-style routine
// does (info = -k), stages host arrays into padded device work arrays, runs the stage
// kernels on one HIP stream and hands results back.  No numerical work happens on the CPU.
#include "ek_api_internal.h"

namespace ek {
namespace api {

Context g_ctx;
std::mutex g_mu;

int ensure_init() {
  if (g_ctx.ready) return 0;
  int ndev = 0;
  EK_HIP_CHECK(hipGetDeviceCount(&ndev));
  if (ndev <= 0) {
    fprintf(stderr, "[ek_hip] no HIP device visible: this library has no CPU fallback\n");
    return -1000 - (int)hipErrorNoDevice;
  }
  EK_HIP_CHECK(hipSetDevice(g_ctx.device));
  EK_HIP_CHECK(hipStreamCreateWithFlags(&g_ctx.stream, hipStreamNonBlocking));
  {
    // the look-ahead stream carries short latency-bound chains beside a chip-filling GEMM: its
    // workgroups must be dispatched first
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    EK_HIP_CHECK(hipStreamCreateWithPriority(&g_ctx.stream2, hipStreamNonBlocking, hi));
  }
  EK_HIP_CHECK(hipMalloc((void **)&g_ctx.d_info, 64 * sizeof(int)));
  EK_HIP_CHECK(hipMalloc((void **)&g_ctx.d_status, 64));
  EK_HIP_CHECK(hipMalloc((void **)&g_ctx.d_stats, 64));
  EK_HIP_CHECK(hipMemset(g_ctx.d_stats, 0, 64));
  g_ctx.ready = true;
  return 0;
}

// device images of the caller's host arrays (ek_hip_solve on a 1 x 1 grid): grown, never shrunk, released in ek_hip_finalize
namespace { void *g_uimg = nullptr; size_t g_uimg_bytes = 0; }
int user_images(size_t bytes, void **p) {
  if (bytes > g_uimg_bytes) {
    if (g_uimg) EK_HIP_CHECK(hipFree(g_uimg));
    g_uimg = nullptr; g_uimg_bytes = 0;
    EK_HIP_CHECK(hipMalloc(&g_uimg, bytes));
    g_uimg_bytes = bytes;
  }
  *p = g_uimg;
  return 0;
}
void release_user_images() {
  if (g_uimg) (void)hipFree(g_uimg);
  g_uimg = nullptr; g_uimg_bytes = 0;
}

int workspace(size_t bytes, void **p) {
  if (bytes > g_ctx.ws_bytes) {
    if (g_ctx.ws_alloc) EK_HIP_CHECK(hipFree(g_ctx.ws_alloc));
    g_ctx.ws = nullptr; g_ctx.ws_alloc = nullptr; g_ctx.ws_bytes = 0;
    EK_HIP_CHECK(hipMalloc(&g_ctx.ws_alloc, bytes));
    g_ctx.ws = g_ctx.ws_alloc;
    g_ctx.ws_bytes = bytes;
  }
  *p = g_ctx.ws;
  return 0;
}


// Placement of the tridiagonalisation's scratch.  The HBM-bound symv runs 2.6 % faster (4.5 % on the
// largest trailing matrices) or slower depending on where its SCRATCH (x, the panel, the partial
// sums: 64 MB at N = 16384) lies relative to the matrix it streams: device memory comes in two
// "colours" of some physical origin (tools/placement_addr.py: every allocation has one, large blocks
// change it at multi-GiB boundaries), and the launch is fast when matrix and scratch have different
// colours -- whichever they are -- and slow when they share one.  Nothing about an address tells its
// colour, so it is measured: when the workspace has been (re)allocated, the first panel of a
// tridiagonalisation of a synthetic matrix is timed in the matrix's place with the scratch (a) where
// the arena has it and (b) in up to three small separate allocations, until one is clearly faster
// than another; that one is kept for all later solves.  ~30 ms per candidate, once per workspace
// size (it happens in the warm-up solve); EK_HIP_PLACEMENT=0 turns it off.
namespace {
struct ScratchChoice {
  void *buf = nullptr;            // separately allocated scratch in use (nullptr: the arena's own)
  size_t bytes = 0;
  const void *for_ws = nullptr;   // the workspace allocation this choice was made for
  int for_n = 0;
};
ScratchChoice g_scratch;
}  // namespace

void release_scratch_choice() {
  if (g_scratch.buf) (void)hipFree(g_scratch.buf);
  g_scratch = ScratchChoice{};
}

// returns the scratch to use for the tridiagonalisation of order n on matrix wA (arena_work if nothing better)
void *choose_sytrd_scratch(int n, int ld, double *wA, void *arena_work, double *vecs, size_t need) {
  static int enabled = -1;
  if (enabled < 0) { const char *e = getenv("EK_HIP_PLACEMENT"); enabled = e ? atoi(e) : 1; }
  if (!enabled || n < 8192) return arena_work;
  if (g_scratch.for_ws == g_ctx.ws_alloc && g_scratch.for_n == n && g_scratch.bytes >= need)
    return g_scratch.buf ? g_scratch.buf : arena_work;
  release_scratch_choice();
  hipStream_t s = g_ctx.stream;
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { (void)hipGetLastError(); return arena_work; }
  const int old_cols = sytrd_get_max_cols();
  auto probe = [&](void *work) -> float {
    sytrd_set_max_cols(64);
    for (int rep = 0; rep < 2; ++rep) {                                // first pass warms up, second is timed
      (void)hipMemsetAsync(wA, 0, (size_t)ld * ld * 8, s);
      synth_matrix(s, n, 1, wA, ld);
      (void)hipEventRecord(e0, s);
      sytrd_lower(s, n, wA, ld, vecs, vecs + ld, vecs + 2 * (size_t)ld, nullptr, 0, work);
      (void)hipEventRecord(e1, s);
    }
    sytrd_set_max_cols(old_cols);
    float ms = 1e30f;
    if (hipStreamSynchronize(s) == hipSuccess) (void)hipEventElapsedTime(&ms, e0, e1);
    else (void)hipGetLastError();
    return ms;
  };
  constexpr int kCand = 4;
  void *cand[kCand] = {arena_work, nullptr, nullptr, nullptr};
  float t[kCand];
  int ncand = 0, best = 0;
  float tmin = 1e30f, tmax = 0.f;
  char msg[256]; int mlen = 0; msg[0] = 0;
  for (int c = 0; c < kCand; ++c) {
    if (c > 0 && hipMalloc(&cand[c], need) != hipSuccess) { (void)hipGetLastError(); cand[c] = nullptr; break; }
    t[c] = probe(cand[c]);
    ++ncand;
    if (mlen < 240) mlen += snprintf(msg + mlen, sizeof(msg) - mlen, " %.3f", t[c]);
    if (t[c] < tmin) { tmin = t[c]; best = c; }
    if (t[c] > tmax) tmax = t[c];
    if (tmin < 0.98f * tmax) break;                     // both colours seen: the faster one is known
  }
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  for (int c = 1; c < ncand; ++c) if (c != best && cand[c]) (void)hipFree(cand[c]);
  g_scratch.buf = best > 0 ? cand[best] : nullptr;
  g_scratch.bytes = need; g_scratch.for_ws = g_ctx.ws_alloc; g_scratch.for_n = n;
  if (getenv("EK_HIP_PLACEMENT_VERBOSE"))
    fprintf(stderr, "[ek_hip] scratch placement: first-panel probes (ms; first = inside the arena):%s -> %s\n", msg,
            best ? "a separate allocation" : "the arena's own");
  return g_scratch.buf ? g_scratch.buf : arena_work;
}


// descriptor checks for the 1x1 grid this round implements; returns 0 or the LAPACK-style
// 100*argpos + field code ScaLAPACK uses (-(argpos*100 + field)).
int check_desc(const int *desc, int argpos, int m, int n, int lld_rows) {
  if (lld_rows < 0) lld_rows = m;
  if (!desc) return -argpos;
  if (desc[0] != 1) return -(argpos * 100 + 1);
  if (desc[2] != m) return -(argpos * 100 + 3);
  if (desc[3] != n) return -(argpos * 100 + 4);
  if (desc[4] < 1 || desc[4] != desc[5]) return -(argpos * 100 + 5);
  if (desc[6] != 0) return -(argpos * 100 + 7);
  if (desc[7] != 0) return -(argpos * 100 + 8);
  if (desc[8] < (lld_rows > 1 ? lld_rows : 1)) return -(argpos * 100 + 9);
  return 0;
}

// NUMROC with source process 0: rows/columns of an n-long dimension (blocks nb) owned by `me` of `np`
int numroc0(int n, int nb, int me, int np) {
  const int nblocks = n / nb;
  int num = (nblocks / np) * nb;
  const int extra = nblocks % np;
  if (me < extra) num += nb;
  else if (me == extra) num += n % nb;
  return num;
}


// Exchange hook for block-cyclically distributed inputs (ek_hip_set_allgatherv)
ek_hip_allgatherv_fn g_allgatherv = nullptr;
void *g_allgatherv_user = nullptr;


// test aid (EK_HIP_TEAM_POISON=1): NaN into every column of the strips a member does not own, to
// prove that the distributed tridiagonalisation never reads them
namespace {
__global__ void poison_foreign_strips_kernel(int n, double *A, int lda, int P, int rank) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)n * n) return;
  const int r = (int)(idx % n), c = (int)(idx / n);
  if ((c / 128) % P != rank) A[(size_t)r + (size_t)c * lda] = __longlong_as_double(0x7ff8000000000000ll);
}

__global__ void count_mismatch_kernel(int m, int n, const double *X, int ldx, const double *Y, int ldy,
                                      int lower, unsigned long long *count) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)m * n) return;
  const int r = (int)(idx % m), c = (int)(idx / m);
  if (lower && r < c) return;
  const unsigned long long a = __double_as_longlong(X[(size_t)r + (size_t)c * ldx]);
  const unsigned long long b = __double_as_longlong(Y[(size_t)r + (size_t)c * ldy]);
  if (a != b) atomicAdd(count, 1ull);
}

}  // namespace
void poison_foreign_strips(hipStream_t s, int n, double *A, int lda, int P, int rank) {
  hipLaunchKernelGGL(poison_foreign_strips_kernel, dim3((unsigned)(((size_t)n * n + 255) / 256)), dim3(256), 0, s, n, A, lda, P, rank);
}
void count_mismatch(hipStream_t s, int m, int n, const double *X, int ldx, const double *Y, int ldy, int lower,
                    unsigned long long *count) {
  hipLaunchKernelGGL(count_mismatch_kernel, dim3((unsigned)(((size_t)m * n + 255) / 256)), dim3(256), 0, s, m, n, X, ldx, Y, ldy,
                     lower, count);
}

int pad_ld(int n) {
  static int extra = -1;
  if (extra < 0) { const char *e = getenv("EK_HIP_LDPAD"); extra = e ? atoi(e) : 0; }
  return round_up(n > 0 ? n : 1, 128) + extra;
}

int h2d_matrix(int m, int n, const double *h, int ldh, double *d, int ldd, hipStream_t s) {
  EK_HIP_CHECK(hipMemcpy2DAsync(d, (size_t)ldd * sizeof(double), h, (size_t)ldh * sizeof(double),
                                (size_t)m * sizeof(double), n, hipMemcpyHostToDevice, s));
  return 0;
}
int d2h_matrix(int m, int n, const double *d, int ldd, double *h, int ldh, hipStream_t s) {
  EK_HIP_CHECK(hipMemcpy2DAsync(h, (size_t)ldh * sizeof(double), d, (size_t)ldd * sizeof(double),
                                (size_t)m * sizeof(double), n, hipMemcpyDeviceToHost, s));
  return 0;
}

int fetch_info(int *info) {
  EK_HIP_CHECK(hipMemcpyAsync(info, g_ctx.d_info, sizeof(int), hipMemcpyDeviceToHost, g_ctx.stream));
  EK_HIP_CHECK(hipStreamSynchronize(g_ctx.stream));
  return 0;
}


}  // namespace api
}  // namespace ek

using namespace ek;
using namespace ek::api;

extern "C" {

int ek_hip_version(void) { return 2; }

int ek_hip_init(int device) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (g_ctx.ready && g_ctx.device == device) return 0;
  if (g_ctx.ready) return -1;   // already bound to another device
  g_ctx.device = device;
  return ensure_init();
}

int ek_hip_finalize(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_ctx.ready) return 0;
  (void)hipStreamSynchronize(g_ctx.stream);
  if (g_ctx.ws_alloc) (void)hipFree(g_ctx.ws_alloc);
  g_ctx.ws = nullptr; g_ctx.ws_alloc = nullptr; g_ctx.ws_bytes = 0;
  release_scratch_choice();
  release_user_images();
  release_pipe_streams();
  // a communicator does not outlive the library's device state
  comm_teardown();
  return 0;
}

const char *ek_hip_stage_name(int stage) {
  static const char *names[EK_HIP_N_STAGES] = {
      "reduce_generalized:pdpotrf", "reduce_generalized:pdsygst",
      "eigen_solver_scalapack_all:pdsytrd", "eigen_solver_scalapack_all:gather1",
      "eigen_solver_scalapack_all:pdstedc", "eigen_solver_scalapack_all:pdormtr",
      "recovery_generalized", "ek_hip:host_device_copies"};
  return (stage >= 0 && stage < EK_HIP_N_STAGES) ? names[stage] : "";
}

int ek_hip_malloc(void **dptr, unsigned long long bytes) {
  int rc = ensure_init(); if (rc) return rc;
  EK_HIP_CHECK(hipMalloc(dptr, bytes));
  return 0;
}
int ek_hip_free(void *dptr) { EK_HIP_CHECK(hipFree(dptr)); return 0; }
int ek_hip_memcpy_h2d(void *dst, const void *src, unsigned long long bytes) {
  int rc = ensure_init(); if (rc) return rc;
  EK_HIP_CHECK(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
  return 0;
}
int ek_hip_memcpy_d2h(void *dst, const void *src, unsigned long long bytes) {
  int rc = ensure_init(); if (rc) return rc;
  EK_HIP_CHECK(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
  return 0;
}
int ek_hip_synchronize(void) {
  int rc = ensure_init(); if (rc) return rc;
  EK_HIP_CHECK(hipDeviceSynchronize());
  return 0;
}

int ek_hip_dgemm(int transa, int transb, int m, int n, int k, double alpha, const double *A,
                 int lda, const double *B, int ldb, double beta, double *C, int ldc,
                 int lower_only) {
  if (m < 0) return -3; if (n < 0) return -4; if (k < 0) return -5;
  const int ar = transa ? k : m, ac = transa ? m : k, br = transb ? n : k, bc = transb ? k : n;
  if (lda < (ar > 1 ? ar : 1)) return -8;
  if (ldb < (br > 1 ? br : 1)) return -10;
  if (ldc < (m > 1 ? m : 1)) return -13;
  int rc = ensure_init(); if (rc) return rc;
  if (m == 0 || n == 0) return 0;
  std::lock_guard<std::mutex> lk(g_mu);
  hipStream_t s = g_ctx.stream;
  const int pa = pad_ld(ar), pb = pad_ld(br), pc = pad_ld(m);
  void *ws;
  rc = workspace(al((size_t)pa * (ac > 0 ? ac : 1) * 8) + al((size_t)pb * (bc > 0 ? bc : 1) * 8) +
                 al((size_t)pc * n * 8), &ws);
  if (rc) return rc;
  Arena ar_(ws, g_ctx.ws_bytes);
  double *dA = ar_.get<double>((size_t)pa * (ac > 0 ? ac : 1));
  double *dB = ar_.get<double>((size_t)pb * (bc > 0 ? bc : 1));
  double *dC = ar_.get<double>((size_t)pc * n);
  if (k > 0) {
    rc = h2d_matrix(ar, ac, A, lda, dA, pa, s); if (rc) return rc;
    rc = h2d_matrix(br, bc, B, ldb, dB, pb, s); if (rc) return rc;
  }
  rc = h2d_matrix(m, n, C, ldc, dC, pc, s); if (rc) return rc;
  gemm(s, transa != 0, transb != 0, m, n, k, alpha, dA, pa, dB, pb, beta, dC, pc, lower_only != 0);
  EK_HIP_CHECK(hipGetLastError());
  rc = d2h_matrix(m, n, dC, pc, C, ldc, s); if (rc) return rc;
  EK_HIP_CHECK(hipStreamSynchronize(s));
  return 0;
}

int ek_hip_potrf(int n, double *B_loc, const int desc_B[9]) {
  if (n < 0) return -1;
  if (!B_loc && n > 0) return -2;
  int rc = check_desc(desc_B, 3, n, n); if (rc) return rc;
  rc = ensure_init(); if (rc) return rc;
  if (n == 0) return 0;
  std::lock_guard<std::mutex> lk(g_mu);
  hipStream_t s = g_ctx.stream;
  const int ld = pad_ld(n), nblk = ceil_div(n, kDiagNB);
  void *ws;
  const size_t rlb = al(potrf_rl_work_bytes(n, ld));
  rc = workspace(al((size_t)ld * n * 8) + al((size_t)nblk * kDiagNB * kDiagNB * 8) +
                 al((size_t)128 * ld * 8) + rlb, &ws);
  if (rc) return rc;
  Arena a(ws, g_ctx.ws_bytes);
  double *dB = a.get<double>((size_t)ld * n);
  double *dInv = a.get<double>((size_t)nblk * kDiagNB * kDiagNB);
  double *work = a.get<double>((size_t)128 * ld);
  char *rlwork = a.get<char>(rlb);
  rc = h2d_matrix(n, n, B_loc, desc_B[8], dB, ld, s); if (rc) return rc;
  EK_HIP_CHECK(hipMemsetAsync(g_ctx.d_info, 0, sizeof(int), s));
  if (n >= kPotrfRlMin) potrf_lower_rl(s, g_ctx.stream2, n, dB, ld, dInv, g_ctx.d_info, rlwork);
  else potrf_lower(s, n, dB, ld, dInv, g_ctx.d_info, work);
  EK_HIP_CHECK(hipGetLastError());
  rc = d2h_matrix(n, n, dB, ld, B_loc, desc_B[8], s); if (rc) return rc;
  int info = 0;
  rc = fetch_info(&info); if (rc) return rc;
  return info;
}

int ek_hip_sygst(int n, double *A_loc, const int desc_A[9], const double *L_loc,
                 const int desc_B[9], double *scale) {
  if (n < 0) return -1;
  if (!A_loc && n > 0) return -2;
  int rc = check_desc(desc_A, 3, n, n); if (rc) return rc;
  if (!L_loc && n > 0) return -4;
  rc = check_desc(desc_B, 5, n, n); if (rc) return rc;
  rc = ensure_init(); if (rc) return rc;
  if (scale) *scale = 1.0;
  if (n == 0) return 0;
  std::lock_guard<std::mutex> lk(g_mu);
  hipStream_t s = g_ctx.stream;
  const int ld = pad_ld(n), nblk = ceil_div(n, kDiagNB);
  void *ws;
  rc = workspace(2 * al((size_t)ld * n * 8) + al((size_t)nblk * kDiagNB * kDiagNB * 8) +
                 al((size_t)128 * ld * 8) + al(sygst_scratch_doubles(n) * 8), &ws);
  if (rc) return rc;
  Arena a(ws, g_ctx.ws_bytes);
  double *dA = a.get<double>((size_t)ld * n);
  double *dL = a.get<double>((size_t)ld * n);
  double *dInv = a.get<double>((size_t)nblk * kDiagNB * kDiagNB);
  double *work = a.get<double>((size_t)128 * ld);
  double *scr = a.get<double>(sygst_scratch_doubles(n));
  rc = h2d_matrix(n, n, A_loc, desc_A[8], dA, ld, s); if (rc) return rc;
  rc = h2d_matrix(n, n, L_loc, desc_B[8], dL, ld, s); if (rc) return rc;
  trtri_diag_blocks(s, n, dL, ld, dInv);
  sygst_lower(s, n, dA, ld, dL, ld, dInv, work, scr);
  EK_HIP_CHECK(hipGetLastError());
  rc = d2h_matrix(n, n, dA, ld, A_loc, desc_A[8], s); if (rc) return rc;
  EK_HIP_CHECK(hipStreamSynchronize(s));
  return 0;
}

int ek_hip_trtrs(int n, int nrhs, const double *L_loc, const int desc_B[9], double *Z_loc,
                 const int desc_Z[9]) {
  if (n < 0) return -1;
  if (nrhs < 0) return -2;
  if (!L_loc && n > 0) return -3;
  int rc = check_desc(desc_B, 4, n, n); if (rc) return rc;
  if (!Z_loc && n > 0 && nrhs > 0) return -5;
  if (!desc_Z) return -6;
  rc = check_desc(desc_Z, 6, n, desc_Z[3]); if (rc) return rc;
  if (desc_Z[3] < nrhs) return -604;
  rc = ensure_init(); if (rc) return rc;
  if (n == 0 || nrhs == 0) return 0;
  for (int i = 0; i < n; ++i)   // PDTRTRS singularity check: info = i if L(i,i) == 0
    if (L_loc[(size_t)i + (size_t)i * desc_B[8]] == 0.0) return i + 1;
  std::lock_guard<std::mutex> lk(g_mu);
  hipStream_t s = g_ctx.stream;
  const int ld = pad_ld(n), nblk = ceil_div(n, kDiagNB);
  void *ws;
  rc = workspace(al((size_t)ld * n * 8) + al((size_t)ld * nrhs * 8) +
                 al((size_t)nblk * kDiagNB * kDiagNB * 8) + al((size_t)128 * (ld > nrhs ? ld : nrhs) * 8), &ws);
  if (rc) return rc;
  Arena a(ws, g_ctx.ws_bytes);
  double *dL = a.get<double>((size_t)ld * n);
  double *dZ = a.get<double>((size_t)ld * nrhs);
  double *dInv = a.get<double>((size_t)nblk * kDiagNB * kDiagNB);
  double *work = a.get<double>((size_t)128 * (ld > nrhs ? ld : nrhs));
  rc = h2d_matrix(n, n, L_loc, desc_B[8], dL, ld, s); if (rc) return rc;
  rc = h2d_matrix(n, nrhs, Z_loc, desc_Z[8], dZ, ld, s); if (rc) return rc;
  trtri_diag_blocks(s, n, dL, ld, dInv);
  trsm_llt(s, n, nrhs, dL, ld, dInv, dZ, ld, work);
  EK_HIP_CHECK(hipGetLastError());
  rc = d2h_matrix(n, nrhs, dZ, ld, Z_loc, desc_Z[8], s); if (rc) return rc;
  EK_HIP_CHECK(hipStreamSynchronize(s));
  return 0;
}


int ek_hip_sytrd(int n, double *A_loc, const int desc_A[9], double *d, double *e, double *tau) {
  if (n < 0) return -1;
  if (!A_loc && n > 0) return -2;
  int rc = check_desc(desc_A, 3, n, n); if (rc) return rc;
  if (n > 0 && !d) return -4;
  if (n > 1 && !e) return -5;
  if (n > 1 && !tau) return -6;
  rc = ensure_init(); if (rc) return rc;
  if (n == 0) return 0;
  std::lock_guard<std::mutex> lk(g_mu);
  hipStream_t s = g_ctx.stream;
  const int ld = pad_ld(n);
  const size_t wb = sytrd_work_bytes(n);
  void *ws;
  rc = workspace(al((size_t)ld * ld * 8) + al(wb) + 3 * al((size_t)ld * 8), &ws);
  if (rc) return rc;
  Arena a(ws, g_ctx.ws_bytes);
  double *dA = a.get<double>((size_t)ld * ld);
  char *work = a.get<char>(wb);
  double *dd = a.get<double>(ld), *de = a.get<double>(ld), *dt = a.get<double>(ld);
  EK_HIP_CHECK(hipMemsetAsync(dA, 0, (size_t)ld * ld * 8, s));
  EK_HIP_CHECK(hipMemsetAsync(dd, 0, 3 * al((size_t)ld * 8), s));
  rc = h2d_matrix(n, n, A_loc, desc_A[8], dA, ld, s); if (rc) return rc;
  sytrd_lower(s, n, dA, ld, dd, de, dt, nullptr, 0, work);
  EK_HIP_CHECK(hipGetLastError());
  rc = d2h_matrix(n, n, dA, ld, A_loc, desc_A[8], s); if (rc) return rc;
  EK_HIP_CHECK(hipMemcpyAsync(d, dd, (size_t)n * 8, hipMemcpyDeviceToHost, s));
  if (n > 1) {
    EK_HIP_CHECK(hipMemcpyAsync(e, de, (size_t)(n - 1) * 8, hipMemcpyDeviceToHost, s));
    EK_HIP_CHECK(hipMemcpyAsync(tau, dt, (size_t)(n - 1) * 8, hipMemcpyDeviceToHost, s));
  }
  EK_HIP_CHECK(hipStreamSynchronize(s));
  return 0;
}

int ek_hip_stedc(int n, double *d, double *e, double *Z_loc, const int desc_Z[9]) {
  if (n < 0) return -1;
  if (n > 0 && !d) return -2;
  if (n > 1 && !e) return -3;
  if (n > 0 && !Z_loc) return -4;
  int rc = check_desc(desc_Z, 5, n, n); if (rc) return rc;
  rc = ensure_init(); if (rc) return rc;
  if (n == 0) return 0;
  std::lock_guard<std::mutex> lk(g_mu);
  hipStream_t s = g_ctx.stream;
  const int ld = pad_ld(n);
  const size_t wb = stedc_work_bytes(n);
  void *ws;
  rc = workspace(al((size_t)ld * n * 8) + al(wb) + 3 * al((size_t)ld * 8), &ws);
  if (rc) return rc;
  Arena a(ws, g_ctx.ws_bytes);
  double *dZ = a.get<double>((size_t)ld * n);
  char *work = a.get<char>(wb);
  double *dd = a.get<double>(ld), *de = a.get<double>(ld), *dw = a.get<double>(ld);
  EK_HIP_CHECK(hipMemcpyAsync(dd, d, (size_t)n * 8, hipMemcpyHostToDevice, s));
  EK_HIP_CHECK(hipMemsetAsync(de, 0, (size_t)ld * 8, s));
  if (n > 1) EK_HIP_CHECK(hipMemcpyAsync(de, e, (size_t)(n - 1) * 8, hipMemcpyHostToDevice, s));
  EK_HIP_CHECK(hipMemsetAsync(g_ctx.d_info, 0, 4 * sizeof(int), s));
  stedc(s, n, dd, de, dw, dZ, ld, work, g_ctx.d_info);
  EK_HIP_CHECK(hipGetLastError());
  rc = d2h_matrix(n, n, dZ, ld, Z_loc, desc_Z[8], s); if (rc) return rc;
  EK_HIP_CHECK(hipMemcpyAsync(d, dw, (size_t)n * 8, hipMemcpyDeviceToHost, s));
  int info = 0;
  rc = fetch_info(&info); if (rc) return rc;
  return info;
}

int ek_hip_ormtr(int n, int ncols, const double *A_loc, const int desc_A[9], const double *tau,
                 double *Z_loc, const int desc_Z[9]) {
  if (n < 0) return -1;
  if (ncols < 0) return -2;
  if (n > 0 && !A_loc) return -3;
  int rc = check_desc(desc_A, 4, n, n); if (rc) return rc;
  if (n > 1 && !tau) return -5;
  if (n > 0 && ncols > 0 && !Z_loc) return -6;
  if (!desc_Z) return -7;
  rc = check_desc(desc_Z, 7, n, desc_Z[3]); if (rc) return rc;
  if (desc_Z[3] < ncols) return -704;
  rc = ensure_init(); if (rc) return rc;
  if (n <= 1 || ncols == 0) return 0;
  std::lock_guard<std::mutex> lk(g_mu);
  hipStream_t s = g_ctx.stream;
  const int ld = pad_ld(n);
  const size_t wb = ormtr_work_bytes(n, ncols);
  void *ws;
  rc = workspace(2 * al((size_t)ld * ld * 8) + al((size_t)ld * ncols * 8) + al(wb) + al((size_t)ld * 8), &ws);
  if (rc) return rc;
  Arena a(ws, g_ctx.ws_bytes);
  double *dA = a.get<double>((size_t)ld * ld);
  double *dV = a.get<double>((size_t)ld * ld);
  double *dZ = a.get<double>((size_t)ld * ncols);
  char *work = a.get<char>(wb);
  double *dt = a.get<double>(ld);
  rc = h2d_matrix(n, n, A_loc, desc_A[8], dA, ld, s); if (rc) return rc;
  rc = h2d_matrix(n, ncols, Z_loc, desc_Z[8], dZ, ld, s); if (rc) return rc;
  EK_HIP_CHECK(hipMemsetAsync(dt, 0, (size_t)ld * 8, s));
  EK_HIP_CHECK(hipMemcpyAsync(dt, tau, (size_t)(n - 1) * 8, hipMemcpyHostToDevice, s));
  EK_HIP_CHECK(hipMemsetAsync(dV, 0, (size_t)ld * ld * 8, s));
  build_explicit_v(s, n, dA, ld, dV, ld);
  ormtr_lower(s, n, ncols, dV, ld, dt, dZ, ld, work);
  EK_HIP_CHECK(hipGetLastError());
  rc = d2h_matrix(n, ncols, dZ, ld, Z_loc, desc_Z[8], s); if (rc) return rc;
  EK_HIP_CHECK(hipStreamSynchronize(s));
  return 0;
}

int ek_hip_synth_matrix_device(int n, unsigned long long seed, double *dM, int ldm) {
  if (n < 0) return -1;
  if (ldm < (n > 1 ? n : 1)) return -4;
  int rc = ensure_init(); if (rc) return rc;
  std::lock_guard<std::mutex> lk(g_mu);
  synth_matrix(g_ctx.stream, n, seed, dM, ldm);
  EK_HIP_CHECK(hipGetLastError());
  EK_HIP_CHECK(hipStreamSynchronize(g_ctx.stream));
  return 0;
}


// ---- acceptance checks on the GPU: device-resident and host-array forms ----
int ek_hip_residual_device(int problem, int n, int n_check, const double *dA, int lda, const double *dB,
                           int ldb, const double *dw, const double *dZ, int ldz, double *a_norm,
                           double *res_ave, double *res_max) {
  if (problem != 0 && problem != 1) return -1;
  if (n < 0) return -2;
  if (n_check < 0 || n_check > n) return -3;
  if (n > 0 && !dA) return -4;
  if (problem == 1 && n > 0 && !dB) return -6;
  int rc = ensure_init(); if (rc) return rc;
  if (n == 0 || n_check == 0) return 0;
  std::lock_guard<std::mutex> lk(g_mu);
  hipStream_t s = g_ctx.stream;
  void *ws;
  rc = workspace(verify_work_bytes(n, n_check) + 256, &ws); if (rc) return rc;
  double *d_res = (double *)((char *)ws + verify_work_bytes(n, n_check));
  residual_norms(s, n, n_check, dA, lda, problem ? dB : nullptr, ldb, dw, dZ, ldz, d_res, ws);
  EK_HIP_CHECK(hipGetLastError());
  double r[3];
  EK_HIP_CHECK(hipMemcpyAsync(r, d_res, sizeof(r), hipMemcpyDeviceToHost, s));
  EK_HIP_CHECK(hipStreamSynchronize(s));
  if (a_norm) *a_norm = r[2];
  if (res_ave) *res_ave = r[0] / r[2] / (double)n_check;      /* verifier.f90:198 */
  if (res_max) *res_max = r[1] / r[2];                        /* verifier.f90:199 */
  return 0;
}

int ek_hip_orthogonality_device(int problem, int n, int index1, int index2, const double *dB, int ldb,
                                const double *dZ, int ldz, double *orthogonality) {
  if (problem != 0 && problem != 1) return -1;
  if (n < 0) return -2;
  if (index1 < 1 || index1 > n) return -3;
  if (index2 < index1 || index2 > n) return -4;
  if (problem == 1 && !dB) return -5;
  int rc = ensure_init(); if (rc) return rc;
  std::lock_guard<std::mutex> lk(g_mu);
  hipStream_t s = g_ctx.stream;
  const int nc = index2 - index1 + 1;
  void *ws;
  rc = workspace(verify_work_bytes(n, nc) + 256, &ws); if (rc) return rc;
  double *d_res = (double *)((char *)ws + verify_work_bytes(n, nc));
  ek::orthogonality(s, n, index1 - 1, nc, problem ? dB : nullptr, ldb, dZ, ldz, d_res, ws);
  EK_HIP_CHECK(hipGetLastError());
  double r[3];
  EK_HIP_CHECK(hipMemcpyAsync(r, d_res, sizeof(r), hipMemcpyDeviceToHost, s));
  EK_HIP_CHECK(hipStreamSynchronize(s));
  if (orthogonality) *orthogonality = r[2];
  return 0;
}

int ek_hip_ipratios_device(int problem, int n, int n_vec, const double *dB, int ldb, const double *dZ,
                           int ldz, double *ipratios_host) {
  if (problem != 0 && problem != 1) return -1;
  if (n < 0) return -2;
  if (n_vec < 0 || n_vec > n) return -3;
  if (problem == 1 && n > 0 && !dB) return -4;
  int rc = ensure_init(); if (rc) return rc;
  if (n == 0 || n_vec == 0) return 0;
  std::lock_guard<std::mutex> lk(g_mu);
  hipStream_t s = g_ctx.stream;
  void *ws;
  rc = workspace(verify_work_bytes(n, n_vec) + al((size_t)n * 8), &ws); if (rc) return rc;
  double *d_ipr = (double *)((char *)ws + verify_work_bytes(n, n_vec));
  ek::ipratios(s, n, n_vec, problem ? dB : nullptr, ldb, dZ, ldz, d_ipr, ws);
  EK_HIP_CHECK(hipGetLastError());
  EK_HIP_CHECK(hipMemcpyAsync(ipratios_host, d_ipr, (size_t)n_vec * 8, hipMemcpyDeviceToHost, s));
  EK_HIP_CHECK(hipStreamSynchronize(s));
  return 0;
}

// Host-array form: what the Fortran host calls in place of eval_residual_norm (verifier.f90:207),
// eval_orthogonality (:333) and get_ipratios (distribute_matrix.f90:18). what: 0 residual
// (out[0..2] = A_norm, res_ave, res_max; uses n_check columns), 1 orthogonality (out[0]; columns
// index1..index2, 1-based), 2 IPR (out[0..n_vec-1]).
int ek_hip_check(int what, int problem, int n, int n_cols, int index1, int index2, const double *A_loc,
                 const int desc_A[9], const double *B_loc, const int desc_B[9], const double *w,
                 const double *Z_loc, const int desc_Z[9], double *out) {
  if (what < 0 || what > 2) return -1;
  if (problem != 0 && problem != 1) return -2;
  if (n < 0) return -3;
  if (n_cols < 0 || n_cols > n) return -4;
  int rc;
  if (what == 0) { if (!A_loc) return -7; rc = check_desc(desc_A, 8, n, n); if (rc) return rc; }
  if (problem == 1) { if (!B_loc) return -9; rc = check_desc(desc_B, 10, n, n); if (rc) return rc; }
  if (!Z_loc) return -12;
  rc = check_desc(desc_Z, 13, n, n); if (rc) return rc;
  if (!out) return -14;
  rc = ensure_init(); if (rc) return rc;
  if (n == 0) return 0;
  double *uA = nullptr, *uB = nullptr, *uZ = nullptr, *uw = nullptr;
  const size_t nn = (size_t)n * n * 8;
  DevMem mem;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    hipStream_t s = g_ctx.stream;
    rc = mem.alloc(&uZ, nn); if (rc) return rc;
    rc = h2d_matrix(n, n, Z_loc, desc_Z[8], uZ, n, s); if (rc) return rc;
    if (what == 0) {
      rc = mem.alloc(&uA, nn); if (rc) return rc;
      rc = mem.alloc(&uw, (size_t)n * 8); if (rc) return rc;
      rc = h2d_matrix(n, n, A_loc, desc_A[8], uA, n, s); if (rc) return rc;
      EK_HIP_CHECK(hipMemcpyAsync(uw, w, (size_t)n * 8, hipMemcpyHostToDevice, s));
    }
    if (problem == 1) {
      rc = mem.alloc(&uB, nn); if (rc) return rc;
      rc = h2d_matrix(n, n, B_loc, desc_B[8], uB, n, s); if (rc) return rc;
    }
    EK_HIP_CHECK(hipStreamSynchronize(s));
  }
  if (what == 0) rc = ek_hip_residual_device(problem, n, n_cols, uA, n, uB, n, uw, uZ, n, &out[0], &out[1], &out[2]);
  else if (what == 1) rc = ek_hip_orthogonality_device(problem, n, index1, index2, uB, n, uZ, n, &out[0]);
  else rc = ek_hip_ipratios_device(problem, n, n_cols, uB, n, uZ, n, out);
  return rc;
}

}  // extern "C"
