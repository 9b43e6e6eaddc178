// ek_api.hip -- the C-ABI of libek_hip.so (declared in include/ek_hip.h).
//
// Host side of the drop-in boundary: validates arguments the way a LAPACK-style routine
// does (info = -k), stages host arrays into padded device work arrays, runs the stage
// kernels on one HIP stream and hands results back.  No numerical work happens on the CPU.
#include "../../include/ek_hip.h"
#include "ek_common.h"

#include <chrono>
#include <cstring>
#include <mutex>
#include <vector>

namespace {

using namespace ek;

struct Context {
  bool ready = false;
  int device = 0;
  hipStream_t stream = nullptr;
  // cached device workspace (grown on demand, never shrunk until finalize)
  void *ws = nullptr;
  size_t ws_bytes = 0;
  int *d_info = nullptr;
};
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
  EK_HIP_CHECK(hipMalloc((void **)&g_ctx.d_info, 64 * sizeof(int)));
  g_ctx.ready = true;
  return 0;
}

int workspace(size_t bytes, void **p) {
  if (bytes > g_ctx.ws_bytes) {
    if (g_ctx.ws) EK_HIP_CHECK(hipFree(g_ctx.ws));
    g_ctx.ws = nullptr; g_ctx.ws_bytes = 0;
    EK_HIP_CHECK(hipMalloc(&g_ctx.ws, bytes));
    g_ctx.ws_bytes = bytes;
  }
  *p = g_ctx.ws;
  return 0;
}

// simple bump allocator over the cached workspace, 256-byte aligned pieces
struct Arena {
  char *base; size_t off = 0, cap;
  Arena(void *p, size_t c) : base((char *)p), cap(c) {}
  template <typename T> T *get(size_t count) {
    size_t bytes = (count * sizeof(T) + 255) & ~(size_t)255;
    T *r = (T *)(base + off);
    off += bytes;
    return r;
  }
};
inline size_t al(size_t bytes) { return (bytes + 255) & ~(size_t)255; }

// descriptor checks for the 1x1 grid this round implements; returns 0 or the LAPACK-style
// 100*argpos + field code ScaLAPACK uses (-(argpos*100 + field)).
int check_desc(const int *desc, int argpos, int m, int n) {
  if (!desc) return -argpos;
  if (desc[0] != 1) return -(argpos * 100 + 1);
  if (desc[2] != m) return -(argpos * 100 + 3);
  if (desc[3] != n) return -(argpos * 100 + 4);
  if (desc[4] < 1 || desc[4] != desc[5]) return -(argpos * 100 + 5);
  if (desc[6] != 0) return -(argpos * 100 + 7);
  if (desc[7] != 0) return -(argpos * 100 + 8);
  if (desc[8] < (m > 1 ? m : 1)) return -(argpos * 100 + 9);
  return 0;
}

inline int pad_ld(int n) { return round_up(n > 0 ? n : 1, 128); }

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

}  // namespace

extern "C" {

int ek_hip_version(void) { return 1; }

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
  if (g_ctx.ws) (void)hipFree(g_ctx.ws);
  g_ctx.ws = nullptr; g_ctx.ws_bytes = 0;
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
  rc = workspace(al((size_t)ld * n * 8) + al((size_t)nblk * kDiagNB * kDiagNB * 8) +
                 al((size_t)128 * ld * 8), &ws);
  if (rc) return rc;
  Arena a(ws, g_ctx.ws_bytes);
  double *dB = a.get<double>((size_t)ld * n);
  double *dInv = a.get<double>((size_t)nblk * kDiagNB * kDiagNB);
  double *work = a.get<double>((size_t)128 * ld);
  rc = h2d_matrix(n, n, B_loc, desc_B[8], dB, ld, s); if (rc) return rc;
  EK_HIP_CHECK(hipMemsetAsync(g_ctx.d_info, 0, sizeof(int), s));
  potrf_lower(s, n, dB, ld, dInv, g_ctx.d_info, work);
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
                 al((size_t)128 * ld * 8), &ws);
  if (rc) return rc;
  Arena a(ws, g_ctx.ws_bytes);
  double *dA = a.get<double>((size_t)ld * n);
  double *dL = a.get<double>((size_t)ld * n);
  double *dInv = a.get<double>((size_t)nblk * kDiagNB * kDiagNB);
  double *work = a.get<double>((size_t)128 * ld);
  rc = h2d_matrix(n, n, A_loc, desc_A[8], dA, ld, s); if (rc) return rc;
  rc = h2d_matrix(n, n, L_loc, desc_B[8], dL, ld, s); if (rc) return rc;
  trtri_diag_blocks(s, n, dL, ld, dInv);
  sygst_lower(s, n, dA, ld, dL, ld, dInv, work);
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

}  // extern "C"
