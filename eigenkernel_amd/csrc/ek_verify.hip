// ek_verify.hip -- the reference's acceptance checks and IPR on GPU-resident data
// (SURVEY.md 8(f) rows 1-2): the same quantities, the same normalisations, the two
// SYMM/GEMM passes on the matrix cores.
//
//   residual       verifier.f90:75-204   R = A V - B V diag(w); avg/max of ||r_j||_2 / ||A||_F
//   orthogonality  verifier.f90:233-330  G = V^T B V, scaled by 1/sqrt(G_jj), zero diagonal, ||G||_F
//   IPR            distribute_matrix.f90:18-78   sum_i v_ij^4 / (sum_i v_ij (S v)_ij)^2
//
// All reductions run in a fixed order (one workgroup per column, no atomics).
#include "ek_common.h"

namespace ek {
namespace {

__device__ __forceinline__ double wg_sum(double v, double *red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// out[j] = sum_i A(i,j)^2  (one workgroup per column)
__global__ __launch_bounds__(256) void col_sumsq_kernel(int m, const double *__restrict__ A, int lda,
                                                        double *__restrict__ out) {
  __shared__ double red[4];
  const double *col = A + (size_t)blockIdx.x * lda;
  double s = 0.0;
  for (int i = threadIdx.x; i < m; i += 256) s += col[i] * col[i];
  s = wg_sum(s, red);
  if (threadIdx.x == 0) out[blockIdx.x] = s;
}

// R(:,j) *= -w[j]   (verifier.f90:160-163)
__global__ void scale_cols_kernel(int m, int n, double *R, int ldr, const double *__restrict__ w) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  for (int j = blockIdx.y; j < n; j += gridDim.y) R[(size_t)i + (size_t)j * ldr] *= -w[j];
}

// result[0] = sum, result[1] = max of sqrt(v[j]) over j < n; result[2] = sqrt(sum_j a[j])
__global__ __launch_bounds__(256) void finish_residual_kernel(int n, const double *__restrict__ colsq,
                                                              int na, const double *__restrict__ asq,
                                                              double *result) {
  __shared__ double red[4];
  __shared__ double mx[256];
  double s = 0.0, m = 0.0, a = 0.0;
  for (int j = threadIdx.x; j < n; j += 256) { const double r = sqrt(colsq[j]); s += r; m = fmax(m, r); }
  for (int j = threadIdx.x; j < na; j += 256) a += asq[j];
  s = wg_sum(s, red);
  a = wg_sum(a, red);
  mx[threadIdx.x] = m;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) mx[threadIdx.x] = fmax(mx[threadIdx.x], mx[threadIdx.x + o]); __syncthreads(); }
  if (threadIdx.x == 0) { result[0] = s; result[1] = mx[0]; result[2] = sqrt(a); }
}

// G(i,j) <- G(i,j) / sqrt(G_ii G_jj), diagonal zeroed; colsq[j] = sum_i of the scaled squares
__global__ __launch_bounds__(256) void ortho_scale_kernel(int n, const double *__restrict__ G, int ldg,
                                                          double *__restrict__ colsq) {
  __shared__ double red[4];
  const int j = blockIdx.x;
  const double sj = 1.0 / sqrt(G[(size_t)j + (size_t)j * ldg]);
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) {
    if (i == j) continue;
    const double g = G[(size_t)i + (size_t)j * ldg] * sj / sqrt(G[(size_t)i + (size_t)i * ldg]);
    s += g * g;
  }
  s = wg_sum(s, red);
  if (threadIdx.x == 0) colsq[j] = s;
}

// ipr[j] = sum v^4 / (sum v * sv)^2
__global__ __launch_bounds__(256) void ipr_kernel(int m, const double *__restrict__ V, int ldv,
                                                  const double *__restrict__ SV, int ldsv,
                                                  double *__restrict__ ipr) {
  __shared__ double red[4];
  const double *v = V + (size_t)blockIdx.x * ldv, *sv = SV + (size_t)blockIdx.x * ldsv;
  double p4 = 0.0, p2 = 0.0;
  for (int i = threadIdx.x; i < m; i += 256) {
    const double x = v[i], x2 = x * x;
    p4 += x2 * x2; p2 += x * sv[i];
  }
  p4 = wg_sum(p4, red);
  p2 = wg_sum(p2, red);
  if (threadIdx.x == 0) ipr[blockIdx.x] = p4 / (p2 * p2);
}

inline size_t al256(size_t b) { return (b + 255) & ~(size_t)255; }

}  // namespace

size_t verify_work_bytes(int n, int ncols) {
  const size_t nn = (size_t)n * (n > 0 ? n : 1), nc = (size_t)n * (ncols > 0 ? ncols : 1);
  return 2 * al256(nn * 8) + 2 * al256(nc * 8) + al256((size_t)(ncols > n ? ncols : n) * (size_t)(ncols > 0 ? ncols : 1) * 8) +
         4 * al256((size_t)(n + ncols + 8) * 8);
}

// d_result (device, 3 doubles): sum_j ||r_j||, max_j ||r_j||, ||A||_F
void residual_norms(hipStream_t s, int n, int n_check, const double *A, int lda, const double *B, int ldb,
                    const double *w, const double *V, int ldv, double *d_result, void *work) {
  char *p = (char *)work;
  const size_t nn = (size_t)n * n;
  double *As = (double *)p; p += al256(nn * 8);
  double *Bs = (double *)p; p += al256(nn * 8);
  double *R = (double *)p; p += al256((size_t)n * n_check * 8);
  p += al256((size_t)n * n_check * 8);
  p += al256((size_t)(n_check > n ? n_check : n) * n_check * 8);
  double *colsq = (double *)p; p += al256((size_t)(n + n_check + 8) * 8);
  double *asq = (double *)p;
  // 'L' semantics of PDSYMM: only the lower triangles of A and B are referenced
  copy_matrix(s, n, n, A, lda, As, n);
  symmetrize_lower(s, n, As, n);
  if (B) {
    copy_matrix(s, n, n, B, ldb, Bs, n);
    symmetrize_lower(s, n, Bs, n);
    gemm(s, false, false, n, n_check, n, 1.0, Bs, n, V, ldv, 0.0, R, n);   // Residual <- B V  (:142)
  } else {
    copy_matrix(s, n, n_check, V, ldv, R, n);                               // Residual <- V    (:151)
  }
  hipLaunchKernelGGL(scale_cols_kernel, dim3(ceil_div(n, 256), n_check < 1024 ? n_check : 1024), dim3(256), 0, s,
                     n, n_check, R, n, w);                                  // * -lambda_j      (:160)
  gemm(s, false, false, n, n_check, n, 1.0, As, n, V, ldv, 1.0, R, n);      // += A V           (:169)
  hipLaunchKernelGGL(col_sumsq_kernel, dim3(n_check), dim3(256), 0, s, n, R, n, colsq);
  hipLaunchKernelGGL(col_sumsq_kernel, dim3(n), dim3(256), 0, s, n, As, n, asq);
  hipLaunchKernelGGL(finish_residual_kernel, dim3(1), dim3(256), 0, s, n_check, colsq, n, asq, d_result);
}

// d_result[0] = || scaled (V^T B V) with zero diagonal ||_F for columns [c0, c0+nc)
void orthogonality(hipStream_t s, int n, int c0, int nc, const double *B, int ldb, const double *V, int ldv,
                   double *d_result, void *work) {
  char *p = (char *)work;
  const size_t nn = (size_t)n * n;
  p += al256(nn * 8);
  double *Bs = (double *)p; p += al256(nn * 8);
  p += al256((size_t)n * nc * 8);
  double *BV = (double *)p; p += al256((size_t)n * nc * 8);
  double *G = (double *)p; p += al256((size_t)(nc > n ? nc : n) * nc * 8);
  double *colsq = (double *)p; p += al256((size_t)(n + nc + 8) * 8);
  double *zero = (double *)p;
  const double *Vs = V + (size_t)c0 * ldv;
  if (B) {
    copy_matrix(s, n, n, B, ldb, Bs, n);
    symmetrize_lower(s, n, Bs, n);
    gemm(s, false, false, n, nc, n, 1.0, Bs, n, Vs, ldv, 0.0, BV, n);       // BV <- B V        (:279)
    gemm(s, true, false, nc, nc, n, 1.0, Vs, ldv, BV, n, 0.0, G, nc);       // V^T BV           (:289)
  } else {
    gemm(s, true, false, nc, nc, n, 1.0, Vs, ldv, Vs, ldv, 0.0, G, nc);     // V^T V            (:299)
  }
  hipLaunchKernelGGL(ortho_scale_kernel, dim3(nc), dim3(256), 0, s, nc, G, nc, colsq);
  (void)hipMemsetAsync(zero, 0, 8, s);
  // reuse finish: result[2] = sqrt(sum colsq) -> Frobenius norm
  hipLaunchKernelGGL(finish_residual_kernel, dim3(1), dim3(256), 0, s, 0, colsq, nc, colsq, d_result);
}

void ipratios(hipStream_t s, int n, int n_vec, const double *B, int ldb, const double *V, int ldv,
              double *d_ipr, void *work) {
  char *p = (char *)work;
  const size_t nn = (size_t)n * n;
  p += al256(nn * 8);
  double *Bs = (double *)p; p += al256(nn * 8);
  double *SV = (double *)p;
  if (B) {
    copy_matrix(s, n, n, B, ldb, Bs, n);
    symmetrize_lower(s, n, Bs, n);
    gemm(s, false, false, n, n_vec, n, 1.0, Bs, n, V, ldv, 0.0, SV, n);     // SV <- S V (distribute_matrix.f90:47)
    hipLaunchKernelGGL(ipr_kernel, dim3(n_vec), dim3(256), 0, s, n, V, ldv, SV, n, d_ipr);
  } else {
    hipLaunchKernelGGL(ipr_kernel, dim3(n_vec), dim3(256), 0, s, n, V, ldv, V, ldv, d_ipr);
  }
}

}  // namespace ek
