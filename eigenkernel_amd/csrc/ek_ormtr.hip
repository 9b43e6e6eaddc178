// ek_ormtr.hip -- back-transformation Z <- Q Z, Q = H(0) H(1) ... H(n-2).
// Replaces PDORMTR('L','L','N') at solver_scalapack_all.f90:115 (1x1 grid).
//
// Compact-WY on the matrix cores: reflectors are grouped KB = 128 at a time,
// Q_b = I - V_b T_b V_b^T, and each group costs three GEMMs
//     W1 = V_b^T Z      W2 = T_b W1      Z -= V_b W2
// applied from the last group to the first.  All Gram matrices V_b^T V_b are formed by ONE
// batched GEMM and all triangular factors T_b by ONE batched launch (a 128x128 LDS image
// per workgroup) before the sweep, so the sweep itself is nothing but large GEMMs.
// V is the explicit unit-lower-trapezoidal reflector matrix the tridiagonalisation writes
// (zeros above the unit diagonal), so no masking is needed inside the GEMMs.
#include "ek_common.h"

namespace ek {
namespace {

constexpr int KB = 128;

// T_b from G_b = V_b^T V_b and tau (forward, columnwise: DLARFT):
//   T(i,i) = tau_i,  T(0:i, i) = -tau_i * T(0:i,0:i) * G(0:i, i)
__global__ __launch_bounds__(128) void larft_kernel(int nrefl, const double *__restrict__ G,
                                                    const double *__restrict__ tau,
                                                    double *__restrict__ T) {
  extern __shared__ double s[];          // KB x KB image of T + one column of G
  double *sT = s, *sg = s + KB * KB;
  const int b = blockIdx.x, t = threadIdx.x;
  const int c0 = b * KB;
  const int kb = (nrefl - c0 < KB) ? nrefl - c0 : KB;
  const double *Gb = G + (size_t)b * KB * KB;
  double *Tb = T + (size_t)b * KB * KB;
  for (int idx = t; idx < KB * KB; idx += 128) sT[idx] = 0.0;
  __syncthreads();
  for (int i = 0; i < kb; ++i) {
    const double ti = tau[c0 + i];
    if (t < i) sg[t] = Gb[(size_t)t + (size_t)i * KB];
    __syncthreads();
    if (t < i) {
      double acc = 0.0;
      for (int c = t; c < i; ++c) acc += sT[t + KB * c] * sg[c];
      sT[t + KB * i] = -ti * acc;
    } else if (t == i) {
      sT[t + KB * i] = ti;
    }
    __syncthreads();
  }
  for (int idx = t; idx < KB * KB; idx += 128) Tb[idx] = sT[idx];
}

// Explicit V from the PDSYTRD storage (reflectors below the sub-diagonal of A).
__global__ void build_v_kernel(int n, const double *__restrict__ A, int lda, double *__restrict__ V,
                               int ldv) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  for (int j = blockIdx.y; j < n; j += gridDim.y) {
    double v = 0.0;
    if (r == j + 1) v = 1.0;
    else if (r > j + 1) v = A[(size_t)r + (size_t)j * lda];
    V[(size_t)r + (size_t)j * ldv] = v;
  }
}

inline size_t al256(size_t b) { return (b + 255) & ~(size_t)255; }

}  // namespace

void build_explicit_v(hipStream_t s, int n, const double *A, int lda, double *V, int ldv) {
  if (n <= 0) return;
  hipLaunchKernelGGL(build_v_kernel, dim3(ceil_div(n, 256), n < 4096 ? n : 4096), dim3(256), 0, s,
                     n, A, lda, V, ldv);
}

size_t ormtr_work_bytes(int n, int ncols) {
  const int nblk = ceil_div(n > 1 ? n - 1 : 1, KB);
  return 2 * al256((size_t)nblk * KB * KB * 8) + 2 * al256((size_t)KB * (ncols > 0 ? ncols : 1) * 8);
}

void ormtr_lower(hipStream_t s, int n, int ncols, const double *V, int ldv, const double *tau,
                 double *Z, int ldz, void *work) {
  const int nrefl = n - 1;
  if (nrefl <= 0 || ncols <= 0) return;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void *)larft_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (KB * KB + KB) * sizeof(double));
    attr = true;
  }
  const int nblk = ceil_div(nrefl, KB);
  char *w = (char *)work;
  double *G = (double *)w; w += al256((size_t)nblk * KB * KB * 8);
  double *T = (double *)w; w += al256((size_t)nblk * KB * KB * 8);
  double *W1 = (double *)w; w += al256((size_t)KB * ncols * 8);
  double *W2 = (double *)w;

  // all Gram matrices in one batched GEMM (rows above a block's reflectors are zero in V)
  {
    GemmDesc g{};
    g.M = KB; g.N = KB; g.K = n; g.transA = true; g.transB = false; g.alpha = 1.0; g.beta = 0.0;
    g.A = V; g.lda = ldv; g.strideA = (long long)KB * ldv;
    g.B = V; g.ldb = ldv; g.strideB = (long long)KB * ldv;
    g.C = G; g.ldc = KB; g.strideC = (long long)KB * KB;
    g.batch = nblk; g.lower_only = false;
    if (nrefl % KB != 0) {      // last block is short: run it separately with its true width
      g.batch = nblk - 1;
      if (g.batch > 0) gemm(s, g);
      const int c0 = (nblk - 1) * KB, kb = nrefl - c0;
      gemm(s, true, false, kb, kb, n, 1.0, V + (size_t)c0 * ldv, ldv, V + (size_t)c0 * ldv, ldv, 0.0,
           G + (size_t)(nblk - 1) * KB * KB, KB);
    } else {
      gemm(s, g);
    }
  }
  hipLaunchKernelGGL(larft_kernel, dim3(nblk), dim3(128), (KB * KB + KB) * sizeof(double), s, nrefl, G,
                     tau, T);
  for (int b = nblk - 1; b >= 0; --b) {
    const int c0 = b * KB;
    const int kb = (nrefl - c0 < KB) ? nrefl - c0 : KB;
    const int row0 = c0 + 1, m = n - row0;
    const double *Vb = V + (size_t)row0 + (size_t)c0 * ldv;
    double *Zb = Z + row0;
    gemm(s, true, false, kb, ncols, m, 1.0, Vb, ldv, Zb, ldz, 0.0, W1, KB);
    gemm(s, false, false, kb, ncols, kb, 1.0, T + (size_t)b * KB * KB, KB, W1, KB, 0.0, W2, KB);
    gemm(s, false, false, m, ncols, kb, -1.0, Vb, ldv, W2, KB, 1.0, Zb, ldz);
  }
}

}  // namespace ek
