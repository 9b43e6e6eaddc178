// ek_chol.hip -- Cholesky factorisation, triangular solves and the reduction of the
// generalized problem to standard form.
//
// Replaces, on a 1x1 grid:
//   PDPOTRF('L')          generalized_to_standard.f90:24    B = L L^T
//   PDSYGST(1,'L')        generalized_to_standard.f90:37    A <- L^-1 A L^-T
//   PDTRTRS('L','T','N')  generalized_to_standard.f90:103   V <- L^-T V
//
// MI355X shape of the algorithm: everything above a 128x128 diagonal block is a *recursive*
// blocked algorithm whose off-diagonal work is handed to the MFMA GEMM in the largest
// possible pieces (half of the remaining matrix at every level), instead of the fixed
// NB=64 panels of the reference's ScaLAPACK path.  The 128x128 diagonal blocks are factored
// AND explicitly inverted by one workgroup entirely inside LDS (L in the lower triangle,
// inv(L)^T in the upper triangle of the same 128 KB image), so that every triangular solve
// against a diagonal block becomes one more MFMA GEMM.
#include "ek_common.h"

namespace ek {
namespace {

constexpr int NB = kDiagNB;   // 128

// One workgroup: B(0:nb,0:nb) = L L^T in LDS, then inv(L).  s is the 128x128 column-major
// image; positions outside nb are the identity so that short edge blocks need no special
// casing.  info (device): first non-positive pivot (1-based, global numbering) or 0.
// FACTOR = false: B already holds L; only the inverses are formed, one workgroup per
// diagonal block (blockIdx.x), for stage-level calls that receive L from the host.
template <bool FACTOR>
__global__ __launch_bounds__(256) void potrf_diag_kernel(int nb_or_n, double *B, int ldb,
                                                         double *inv, int *info, int info_base) {
  extern __shared__ double s[];
  const int t = threadIdx.x;
  int nb = nb_or_n;
  if (!FACTOR) {
    const int off = blockIdx.x * NB;
    nb = nb_or_n - off < NB ? nb_or_n - off : NB;
    B += (size_t)off + (size_t)off * ldb;
    inv += (size_t)blockIdx.x * NB * NB;
  }
  for (int idx = t; idx < NB * NB; idx += 256) {
    const int i = idx & (NB - 1), j = idx >> 7;
    double v = (i == j) ? 1.0 : 0.0;
    if (i < nb && j < nb && i >= j) v = B[(size_t)i + (size_t)j * ldb];
    s[idx] = v;
  }
  bool failed = false;
  for (int j = 0; FACTOR && j < nb; ++j) {
    __syncthreads();
    const double d = s[j + NB * j];
    if (!(d > 0.0)) {
      if (t == 0) atomicCAS(info, 0, info_base + j + 1);
      failed = true;
      break;
    }
    const double l = sqrt(d), r = 1.0 / l;
    __syncthreads();
    if (t < NB) {
      if (t > j) s[t + NB * j] *= r;
      else if (t == j) s[t + NB * j] = l;
    }
    __syncthreads();
    const int i = j + 1 + (t & (NB - 1));
    if (i < nb) {
      const double lij = s[i + NB * j];
      for (int k = j + 1 + (t >> 7); k <= i; k += 2) s[i + NB * k] -= lij * s[k + NB * j];
    }
  }
  __syncthreads();
  if (FACTOR)
    for (int idx = t; idx < NB * NB; idx += 256) {
      const int i = idx & (NB - 1), j = idx >> 7;
      if (i < nb && j < nb && i >= j) B[(size_t)i + (size_t)j * ldb] = s[idx];
    }
  if (failed) {   // leave a harmless inverse so later kernels stay finite
    for (int idx = t; idx < NB * NB; idx += 256)
      inv[idx] = ((idx & (NB - 1)) == (idx >> 7)) ? 1.0 : 0.0;
    return;
  }
  // inv(L): thread c owns column c of X = inv(L); X(i,c), i > c, lives at s[c + NB*i]
  // (the mirrored, strictly-upper position), so lanes touch consecutive LDS words while
  // L(i,k) is one broadcast word.  A column only reads its own earlier entries: no barrier.
  if (t < NB) {
    const int c = t;
    const double dinv = 1.0 / s[c + NB * c];
    for (int i = 1; i < NB; ++i) {
      double acc = 0.0;
      for (int k = 0; k < i; ++k) {
        const double lik = s[i + NB * k];
        const double xkc = (k > c) ? s[c + NB * k] : (k == c ? dinv : 0.0);
        acc += lik * xkc;
      }
      if (i > c) s[c + NB * i] = -acc / s[i + NB * i];
    }
  }
  __syncthreads();
  for (int idx = t; idx < NB * NB; idx += 256) {
    const int i = idx & (NB - 1), c = idx >> 7;
    double v = 0.0;
    if (i > c) v = s[c + NB * i];
    else if (i == c) v = 1.0 / s[c + NB * c];
    inv[idx] = v;
  }
}

inline int split(int n) {   // first part of a recursive split, multiple of 128
  int n1 = round_up(n / 2, NB);
  if (n1 >= n) n1 -= NB;
  if (n1 < NB) n1 = NB;
  return n1;
}

void potrf_rec(hipStream_t s, int n, double *B, int ldb, int off, double *invdiag, int *d_info,
               double *work) {
  double *Bd = B + (size_t)off + (size_t)off * ldb;
  if (n <= NB) {
    hipLaunchKernelGGL(potrf_diag_kernel<true>, dim3(1), dim3(256), NB * NB * sizeof(double), s, n,
                       Bd, ldb, invdiag + (size_t)(off / NB) * NB * NB, d_info, off);
    return;
  }
  const int n1 = split(n), n2 = n - n1;
  potrf_rec(s, n1, B, ldb, off, invdiag, d_info, work);
  double *B21 = Bd + n1, *B22 = Bd + (size_t)n1 + (size_t)n1 * ldb;
  trsm_rlt(s, n2, n1, Bd, ldb, invdiag + (size_t)(off / NB) * NB * NB, B21, ldb, work);
  gemm(s, false, true, n2, n2, n1, -1.0, B21, ldb, B21, ldb, 1.0, B22, ldb, /*lower_only=*/true);
  potrf_rec(s, n2, B, ldb, off + n1, invdiag, d_info, work);
}

}  // namespace

static void set_attrs() {
  static bool attr_set = false;
  if (attr_set) return;
  (void)hipFuncSetAttribute((const void *)potrf_diag_kernel<true>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, NB * NB * sizeof(double));
  (void)hipFuncSetAttribute((const void *)potrf_diag_kernel<false>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, NB * NB * sizeof(double));
  attr_set = true;
}

void potrf_lower(hipStream_t s, int n, double *B, int ldb, double *invdiag, int *d_info,
                 double *work) {
  set_attrs();
  if (n <= 0) return;
  potrf_rec(s, n, B, ldb, 0, invdiag, d_info, work);
}

void trtri_diag_blocks(hipStream_t s, int n, const double *L, int ldl, double *invdiag) {
  set_attrs();
  if (n <= 0) return;
  hipLaunchKernelGGL(potrf_diag_kernel<false>, dim3(ceil_div(n, NB)), dim3(256),
                     NB * NB * sizeof(double), s, n, const_cast<double *>(L), ldl, invdiag,
                     (int *)nullptr, 0);
}

// invdiag points at the inverse of the FIRST diagonal block of L (blocks follow at
// 128*128-double intervals); L must start on a 128-aligned diagonal position.

// X <- X L^-T  (X: m x n, L: n x n lower)
void trsm_rlt(hipStream_t s, int m, int n, const double *L, int ldl, const double *invdiag,
              double *X, int ldx, double *work) {
  if (m <= 0 || n <= 0) return;
  if (n <= NB) {
    gemm(s, false, true, m, n, n, 1.0, X, ldx, invdiag, NB, 0.0, work, m);
    copy_matrix(s, m, n, work, m, X, ldx);
    return;
  }
  const int n1 = split(n), n2 = n - n1;
  double *X2 = X + (size_t)n1 * ldx;
  const double *L21 = L + n1, *L22 = L + (size_t)n1 + (size_t)n1 * ldl;
  trsm_rlt(s, m, n1, L, ldl, invdiag, X, ldx, work);
  gemm(s, false, true, m, n2, n1, -1.0, X, ldx, L21, ldl, 1.0, X2, ldx);
  trsm_rlt(s, m, n2, L22, ldl, invdiag + (size_t)(n1 / NB) * NB * NB, X2, ldx, work);
}

// X <- L^-1 X  (X: n x m)
void trsm_lln(hipStream_t s, int n, int m, const double *L, int ldl, const double *invdiag,
              double *X, int ldx, double *work) {
  if (m <= 0 || n <= 0) return;
  if (n <= NB) {
    gemm(s, false, false, n, m, n, 1.0, invdiag, NB, X, ldx, 0.0, work, n);
    copy_matrix(s, n, m, work, n, X, ldx);
    return;
  }
  const int n1 = split(n), n2 = n - n1;
  double *X2 = X + n1;
  const double *L21 = L + n1, *L22 = L + (size_t)n1 + (size_t)n1 * ldl;
  trsm_lln(s, n1, m, L, ldl, invdiag, X, ldx, work);
  gemm(s, false, false, n2, m, n1, -1.0, L21, ldl, X, ldx, 1.0, X2, ldx);
  trsm_lln(s, n2, m, L22, ldl, invdiag + (size_t)(n1 / NB) * NB * NB, X2, ldx, work);
}

// X <- L^-T X  (X: n x m)
void trsm_llt(hipStream_t s, int n, int m, const double *L, int ldl, const double *invdiag,
              double *X, int ldx, double *work) {
  if (m <= 0 || n <= 0) return;
  if (n <= NB) {
    gemm(s, true, false, n, m, n, 1.0, invdiag, NB, X, ldx, 0.0, work, n);
    copy_matrix(s, n, m, work, n, X, ldx);
    return;
  }
  const int n1 = split(n), n2 = n - n1;
  double *X2 = X + n1;
  const double *L21 = L + n1, *L22 = L + (size_t)n1 + (size_t)n1 * ldl;
  trsm_llt(s, n2, m, L22, ldl, invdiag + (size_t)(n1 / NB) * NB * NB, X2, ldx, work);
  gemm(s, true, false, n1, m, n2, -1.0, L21, ldl, X2, ldx, 1.0, X, ldx);
  trsm_llt(s, n1, m, L, ldl, invdiag, X, ldx, work);
}

// A <- L^-1 A L^-T.  The lower triangle of A is the input (as PDSYGST 'L'); the result is
// returned in full storage (both triangles), of which later stages reference the lower.
// Two recursive triangular solves on the whole matrix: 2 N^3 MFMA flops, all in GEMMs
// of order N/2, N/4, ... rather than NB-wide panels.
void sygst_lower(hipStream_t s, int n, double *A, int lda, const double *L, int ldl,
                 const double *invdiag, double *work) {
  if (n <= 0) return;
  symmetrize_lower(s, n, A, lda);
  trsm_lln(s, n, n, L, ldl, invdiag, A, lda, work);
  trsm_rlt(s, n, n, L, ldl, invdiag, A, lda, work);
}

}  // namespace ek
