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
#include "ek_block64.h"

#include <cstdlib>
#include <vector>

namespace ek {
namespace {

constexpr int NB = kDiagNB;   // 128
constexpr size_t kDiagLds = 2 * b64::IMG * sizeof(double);   // dynamic LDS of potrf_diag_kernel

// One workgroup: B(0:nb,0:nb) = L L^T and inv(L), for a 128x128 diagonal block (nb <= 128; positions
// outside nb behave as the identity so that short edge blocks need no special casing).
// info (device): first non-positive pivot (1-based, global numbering) or 0.
//
// The block is handled as 2 x 2 blocks of 64 in terms of the UPPER factor R = L^T held in row-major
// LDS images (ek_block64.h): the two factorisations and the two triangular inverses go by row blocks of 16
// (the serial steps of a block inside one wave, the rest of the image on the matrix cores),
// everything between them is a 64x64 product on the matrix cores:
//   R11 = chol(B11), X11 = R11^-1, R12 = X11^T B12, R22 = chol(B22 - R12^T R12), X22 = R22^-1,
//   X12 = -X11 R12 X22;      L = R^T,  inv(L) = [X11 X12; 0 X22]^T.
// This kernel is the serial chain of the factorisation (one launch per 128 columns, each waiting for
// the previous trailing update), which is why it is built for latency -- and for a small footprint: TWO LDS
// images (76 KB with the scratch; X11 R12 waits in registers for X22).  With four images (144 KB) the
// workgroup could only start on a CU that both resident workgroups of the trailing update's GEMM had left,
// and the GEMM refilled every slot that came free first: the launch took 150 - 750 us instead of 60
// (tools/potrf_trace.sh) for as long as an update was running beside it; it takes 110 - 250 now.  (Measured
// without effect on that: s_setprio(3) in this kernel; the updates on a stream whose CU mask leaves 1, 2 or 4
// CUs per XCD to the panel stream -- 36.6 -> 37.5 - 37.8 ms for the factorisation; the block column solved in
// place by 128-row workgroups instead of gemm + copy -- 36.4 vs 36.6; the trailing update held to one workgroup
// per CU by an LDS pad -- 41.4.)
// inv (128x128, column-major): inv(L)(i,c) for i > c, 1/L(c,c) on the diagonal, zeros above.
__global__ __launch_bounds__(256) void potrf_diag_kernel(int nb, double *B, int ldb, double *inv, int *info,
                                                         int info_base) {
  using namespace b64;
  extern __shared__ double smem[];
  double *sP = smem, *sQ = smem + IMG;
  __shared__ double s_scr[kScratch];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l15 = lane & 15, l4 = lane >> 4;
  // element (gr, gc), gr <= gc, of the symmetric block from its lower triangle
  auto upper = [&](int gr, int gc) -> double {
    if (gc < nb) return B[(size_t)gc + (size_t)gr * ldb];
    return (gr == gc) ? 1.0 : 0.0;
  };
  auto fail_exit = [&](int j) {
    if (t == 0) atomicCAS(info, 0, info_base + j + 1);
    for (int idx = t; idx < NB * NB; idx += 256)      // a harmless inverse so later kernels stay finite
      inv[idx] = ((idx & (NB - 1)) == (idx >> 7)) ? 1.0 : 0.0;
  };
  for (int idx = t; idx < SB * SB; idx += 256) {
    const int c = idx & 63, r = idx >> 6;
    sP[r * LD + c] = (c >= r) ? upper(r, c) : 0.0;        // B11
  }
  __syncthreads();
  int fail = chol64_upper_wg(sP, s_scr);                   // P = R11
  if (fail >= 0) { fail_exit(fail); return; }
  triinv64_upper_wg(sP, sQ, s_scr);                        // Q = X11
  for (int idx = t; idx < SB * SB; idx += 256) {
    const int i = idx & 63, j = idx >> 6;
    if (i >= j && i < nb) B[(size_t)i + (size_t)j * ldb] = sP[j * LD + i];                  // L11
    inv[i + NB * j] = (i >= j) ? sQ[j * LD + i] : 0.0;
    inv[i + NB * (64 + j)] = 0.0;
  }
  __syncthreads();
  for (int idx = t; idx < SB * SB; idx += 256) {
    const int c = idx & 63, r = idx >> 6;
    sP[r * LD + c] = upper(r, 64 + c);                     // P = B12
  }
  __syncthreads();
  double4_t acc[4], Z[4];
  mm64_acc(sQ, true, sP, false, acc);                      // R12 = X11^T B12
  __syncthreads();
  mm64_store(acc, sP);                                     // P = R12
  __syncthreads();
  for (int idx = t; idx < SB * SB; idx += 256) {
    const int i = idx & 63, j = idx >> 6;
    if (64 + i < nb) B[(size_t)(64 + i) + (size_t)j * ldb] = sP[j * LD + i];                // L21
  }
  mm64_acc(sQ, false, sP, false, Z);                       // Z = X11 R12, kept in registers until X22 exists
  mm64_acc(sP, true, sP, false, acc);                      // R12^T R12
  __syncthreads();
#pragma unroll
  for (int jt = 0; jt < 4; ++jt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {                          // Q = B22 - R12^T R12 (upper part)
      const int i = 16 * wave + l4 + 4 * r, j = 16 * jt + l15;
      sQ[i * LD + j] = (j >= i) ? upper(64 + i, 64 + j) - acc[jt][r] : 0.0;
    }
  __syncthreads();
  fail = chol64_upper_wg(sQ, s_scr);                       // Q = R22
  if (fail >= 0) { fail_exit(64 + fail); return; }
  triinv64_upper_wg(sQ, sP, s_scr);                        // P = X22
  for (int idx = t; idx < SB * SB; idx += 256) {
    const int i = idx & 63, j = idx >> 6;
    if (i >= j && 64 + i < nb) B[(size_t)(64 + i) + (size_t)(64 + j) * ldb] = sQ[j * LD + i];   // L22
    inv[(64 + i) + NB * (64 + j)] = (i >= j) ? sP[j * LD + i] : 0.0;
  }
  __syncthreads();
  mm64_store(Z, sQ);                                       // Q = X11 R12
  __syncthreads();
  mm64_acc(sQ, false, sP, false, acc);                     // X11 R12 X22 = -X12
  __syncthreads();
  mm64_store(acc, sQ);
  __syncthreads();
  for (int idx = t; idx < SB * SB; idx += 256) {
    const int i = idx & 63, c = idx >> 6;
    inv[(64 + i) + NB * c] = -sQ[c * LD + i];
  }
}

// inv(L) of every 128x128 diagonal block of a given factor L (one workgroup per block, all in LDS), for
// stage-level calls that receive L from the host.  s is the 128x128 column-major image; positions
// outside nb are the identity.  Blocked by 16: diagonal 16x16 blocks by forward substitution (a column
// per thread), then block row by block row  X_IJ = -inv(L_II) * sum_K L_IK X_KJ.
// inv(L)(i,c), i > c, lives at s[c + 128*i] (the mirrored, strictly-upper position); its diagonal
// 1/L(c,c) in sd[].
constexpr int PB = 16;

__global__ __launch_bounds__(256) void trtri_diag_kernel(int n, const double *B, int ldb, double *inv) {
  extern __shared__ double s[];
  double *sd = s + NB * NB;            // 1 / L(i,i)
  const int t = threadIdx.x;
  const int off = blockIdx.x * NB;
  const int nb = n - off < NB ? n - off : NB;
  B += (size_t)off + (size_t)off * ldb;
  inv += (size_t)blockIdx.x * NB * NB;
  for (int idx = t; idx < NB * NB; idx += 256) {
    const int i = idx & (NB - 1), j = idx >> 7;
    double v = (i == j) ? 1.0 : 0.0;
    if (i < nb && j < nb && i >= j) v = B[(size_t)i + (size_t)j * ldb];
    s[idx] = v;
  }
  __syncthreads();
  if (t < NB) sd[t] = 1.0 / s[t + NB * t];
  __syncthreads();
  // step 1: inverses of the diagonal 16x16 blocks, a column per thread
  if (t < NB) {
    const int c = t, cl = c & (PB - 1), b0 = c - cl;
    double x[PB];
#pragma unroll
    for (int i = 0; i < PB; ++i) x[i] = 0.0;
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      if (i == cl) x[i] = sd[c];
      else if (i > cl) {
        double a = 0.0;
#pragma unroll
        for (int k = 0; k < PB; ++k)
          if (k >= cl && k < i) a += s[(b0 + i) + NB * (b0 + k)] * x[k];
        x[i] = -a * sd[b0 + i];
      }
    }
#pragma unroll
    for (int i = 0; i < PB; ++i)
      if (i > cl) s[c + NB * (b0 + i)] = x[i];
  }
  __syncthreads();
  // step 2: block rows I = 1..7; X_IJ = -inv(L_II) * T, T = sum_{K=J..I-1} L_IK X_KJ
  for (int I = 1; I < NB / PB; ++I) {
    const int ib = I * PB, ncol = ib;       // columns c < ib
    // T(i, c) -> s[c + NB*i] (the final location of X(i,c))
    for (int idx = t; idx < PB * ncol; idx += 256) {
      const int il = idx & (PB - 1), c = idx >> 4;
      const int i = ib + il;
      double a = 0.0;
      for (int k = c; k < ib; ++k) {
        const double xkc = (k == c) ? sd[c] : s[c + NB * k];
        a += s[i + NB * k] * xkc;
      }
      s[c + NB * i] = a;
    }
    __syncthreads();
    // X(ib.., c) = -inv(L_II) T(ib.., c): a column segment per thread (in registers)
    if (t < ncol) {
      const int c = t;
      double tv[PB], xv[PB];
#pragma unroll
      for (int i = 0; i < PB; ++i) tv[i] = s[c + NB * (ib + i)];
#pragma unroll
      for (int i = 0; i < PB; ++i) {
        double a = sd[ib + i] * tv[i];
#pragma unroll
        for (int k = 0; k < PB; ++k)
          if (k < i) a += s[(ib + k) + NB * (ib + i)] * tv[k];   // inv(L_II)(i,k), mirrored
        xv[i] = -a;
      }
#pragma unroll
      for (int i = 0; i < PB; ++i) s[c + NB * (ib + i)] = xv[i];
    }
    __syncthreads();
  }
  for (int idx = t; idx < NB * NB; idx += 256) {
    const int i = idx & (NB - 1), c = idx >> 7;
    double v = 0.0;
    if (i > c) v = s[c + NB * i];
    else if (i == c) v = sd[c];
    inv[idx] = v;
  }
}

inline int split(int n) {   // first part of a recursive split, multiple of 128
  int n1 = round_up(n / 2, NB);
  if (n1 >= n) n1 -= NB;
  if (n1 < NB) n1 = NB;
  return n1;
}

// Leaves of 256 for the triangular solves (round 4).  A solve against a 256 x 256 diagonal block of L through the
// two 128-block inverses is leaf, rank-128 correction, leaf: three small GEMMs and two copies of ~25 us each.  With the
// explicit inverse of the 256-block, inv256 = [I11 0; -I22 L21 I11  I22] (two 128^3 products per block, batched, once
// per factorisation), it is ONE product with K = 256 and one copy.  The whole-path call registers the array for the L it
// has just factored (trsm_register_inv256); a solve whose `invdiag` pointer lies in the registered array of 128-block
// inverses, at an even block, with n = 256, takes it; everything else is as before.
struct Inv256 { const double *base128 = nullptr; const double *inv256 = nullptr; int nblk256 = 0; };
Inv256 g_inv256;
inline const double *leaf256(const double *invdiag, int n) {
  if (!g_inv256.inv256 || n != 2 * NB || invdiag < g_inv256.base128) return nullptr;
  const size_t d = (size_t)(invdiag - g_inv256.base128);
  if (d % ((size_t)2 * NB * NB) != 0) return nullptr;
  const size_t b = d / ((size_t)2 * NB * NB);
  return b < (size_t)g_inv256.nblk256 ? g_inv256.inv256 + b * (size_t)4 * NB * NB : nullptr;
}
inline int split_t(int n) {   // the solves split on multiples of 256 where they can, so that aligned 256-leaves appear
  if (n <= 2 * NB) return NB;
  int n1 = round_up(n / 2, 2 * NB);
  if (n1 >= n) n1 -= 2 * NB;
  if (n1 < 2 * NB) n1 = 2 * NB;
  return n1;
}
__global__ void inv256_assemble_kernel(const double *__restrict__ inv128, double *__restrict__ inv256) {
  const int b = blockIdx.x;
  double *D = inv256 + (size_t)b * 4 * NB * NB;
  const double *I11 = inv128 + (size_t)(2 * b) * NB * NB, *I22 = I11 + NB * NB;
  for (int idx = threadIdx.x; idx < NB * NB; idx += blockDim.x) {
    const int i = idx & (NB - 1), j = idx >> 7;
    D[i + (size_t)j * 2 * NB] = I11[idx];
    D[(NB + i) + (size_t)(NB + j) * 2 * NB] = I22[idx];
    D[i + (size_t)(NB + j) * 2 * NB] = 0.0;
  }
}

void potrf_rec(hipStream_t s, int n, double *B, int ldb, int off, double *invdiag, int *d_info,
               double *work) {
  double *Bd = B + (size_t)off + (size_t)off * ldb;
  if (n <= NB) {
    hipLaunchKernelGGL(potrf_diag_kernel, dim3(1), dim3(256), kDiagLds, s, n,
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
  (void)hipFuncSetAttribute((const void *)potrf_diag_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kDiagLds);
  (void)hipFuncSetAttribute((const void *)trtri_diag_kernel,
                            hipFuncAttributeMaxDynamicSharedMemorySize, (NB * NB + NB) * sizeof(double));
  attr_set = true;
}

void potrf_lower(hipStream_t s, int n, double *B, int ldb, double *invdiag, int *d_info,
                 double *work) {
  set_attrs();
  if (n <= 0) return;
  potrf_rec(s, n, B, ldb, 0, invdiag, d_info, work);
}

// inv256 (n / 256 blocks of 256 x 256, ld 256) from L and the 128-block inverses; scratch: >= (n / 256) * 128 * 128 doubles
void trtri256_blocks(hipStream_t s, int n, const double *L, int ldl, const double *invdiag, double *inv256, double *scratch) {
  const int nb2 = n / (2 * NB);
  if (nb2 <= 0) return;
  GemmDesc g{};
  g.M = NB; g.N = NB; g.K = NB; g.transA = false; g.transB = false; g.alpha = 1.0; g.beta = 0.0; g.batch = nb2; g.lower_only = false;
  g.A = L + NB; g.lda = ldl; g.strideA = (long long)2 * NB * ((long long)ldl + 1);             // L21 of block b
  g.B = invdiag; g.ldb = NB; g.strideB = (long long)2 * NB * NB;                              // I11
  g.C = scratch; g.ldc = NB; g.strideC = (long long)NB * NB;
  gemm(s, g);
  g.alpha = -1.0;
  g.A = invdiag + (size_t)NB * NB; g.lda = NB; g.strideA = (long long)2 * NB * NB;            // I22
  g.B = scratch; g.ldb = NB; g.strideB = (long long)NB * NB;
  g.C = inv256 + NB; g.ldc = 2 * NB; g.strideC = (long long)4 * NB * NB;                      // lower-left quarter
  gemm(s, g);
  hipLaunchKernelGGL(inv256_assemble_kernel, dim3(nb2), dim3(256), 0, s, invdiag, inv256);
}
void trsm_register_inv256(const double *invdiag, const double *inv256, int n) {
  g_inv256.base128 = inv256 ? invdiag : nullptr; g_inv256.inv256 = inv256; g_inv256.nblk256 = inv256 ? n / (2 * NB) : 0;
}

void trtri_diag_blocks(hipStream_t s, int n, const double *L, int ldl, double *invdiag) {
  set_attrs();
  if (n <= 0) return;
  hipLaunchKernelGGL(trtri_diag_kernel, dim3(ceil_div(n, NB)), dim3(256),
                     (NB * NB + NB) * sizeof(double), s, n, L, ldl, invdiag);
}

// invdiag points at the inverse of the FIRST diagonal block of L (blocks follow at
// 128*128-double intervals); L must start on a 128-aligned diagonal position.

// X <- X L^-T  (X: m x n, L: n x n lower)
void trsm_rlt(hipStream_t s, int m, int n, const double *L, int ldl, const double *invdiag,
              double *X, int ldx, double *work) {
  if (m <= 0 || n <= 0) return;
  if (const double *i2 = leaf256(invdiag, n)) {
    gemm(s, false, true, m, n, n, 1.0, X, ldx, i2, 2 * NB, 0.0, work, m);
    copy_matrix(s, m, n, work, m, X, ldx);
    return;
  }
  if (n <= NB) {
    gemm(s, false, true, m, n, n, 1.0, X, ldx, invdiag, NB, 0.0, work, m);
    copy_matrix(s, m, n, work, m, X, ldx);
    return;
  }
  const int n1 = split_t(n), n2 = n - n1;
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
  if (const double *i2 = leaf256(invdiag, n)) {
    gemm(s, false, false, n, m, n, 1.0, i2, 2 * NB, X, ldx, 0.0, work, n);
    copy_matrix(s, n, m, work, n, X, ldx);
    return;
  }
  if (n <= NB) {
    gemm(s, false, false, n, m, n, 1.0, invdiag, NB, X, ldx, 0.0, work, n);
    copy_matrix(s, n, m, work, n, X, ldx);
    return;
  }
  const int n1 = split_t(n), n2 = n - n1;
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
  if (const double *i2 = leaf256(invdiag, n)) {
    gemm(s, true, false, n, m, n, 1.0, i2, 2 * NB, X, ldx, 0.0, work, n);
    copy_matrix(s, n, m, work, n, X, ldx);
    return;
  }
  if (n <= NB) {
    gemm(s, true, false, n, m, n, 1.0, invdiag, NB, X, ldx, 0.0, work, n);
    copy_matrix(s, n, m, work, n, X, ldx);
    return;
  }
  const int n1 = split_t(n), n2 = n - n1;
  double *X2 = X + n1;
  const double *L21 = L + n1, *L22 = L + (size_t)n1 + (size_t)n1 * ldl;
  trsm_llt(s, n2, m, L22, ldl, invdiag + (size_t)(n1 / NB) * NB * NB, X2, ldx, work);
  gemm(s, true, false, n1, m, n2, -1.0, L21, ldl, X2, ldx, 1.0, X, ldx);
  trsm_llt(s, n1, m, L, ldl, invdiag, X, ldx, work);
}

// dst <- dst + alpha * src  (m x n)
__global__ void axpy_matrix_kernel(int m, int n, double alpha, const double *__restrict__ src, int lds,
                                   double *__restrict__ dst, int ldd) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  for (int j = blockIdx.y; j < n; j += gridDim.y)
    dst[(size_t)i + (size_t)j * ldd] += alpha * src[(size_t)i + (size_t)j * lds];
}
// dst (full, ld n) <- symmetric matrix given by the lower triangle of src
__global__ void full_from_lower_kernel(int n, const double *__restrict__ src, int lds,
                                       double *__restrict__ dst, int ldd) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  for (int j = blockIdx.y; j < n; j += gridDim.y)
    dst[(size_t)i + (size_t)j * ldd] = (i >= j) ? src[(size_t)i + (size_t)j * lds] : src[(size_t)j + (size_t)i * lds];
}
// A(i, j) -= M(i, j) + M(j, i) for i >= j (n x n): the two halves of a SYR2K whose product M = X Y^T was formed
// once, in full.  32 x 32 tiles, the mirrored tile through LDS.
__global__ __launch_bounds__(256) void syr2k_fold_kernel(int n, const double *__restrict__ M, int ldm,
                                                         double *__restrict__ A, int lda) {
  __shared__ double sT[32][33];
  const int bi = blockIdx.x, bj = blockIdx.y;
  if (bj > bi) return;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
#pragma unroll
  for (int q = 0; q < 4; ++q) {                                    // mirrored tile: rows 32 bj.., columns 32 bi..
    const int r = 32 * bj + tx, c = 32 * bi + ty + 8 * q;
    sT[ty + 8 * q][tx] = (r < n && c < n) ? M[(size_t)r + (size_t)c * ldm] : 0.0;
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int i = 32 * bi + tx, j = 32 * bj + ty + 8 * q;
    if (i < n && j < n && i >= j)
      A[(size_t)i + (size_t)j * lda] -= M[(size_t)i + (size_t)j * ldm] + sT[tx][ty + 8 * q];
  }
}
static inline dim3 grid_mn(int m, int n) { return dim3(ceil_div(m, 256), n < 2048 ? (n > 0 ? n : 1) : 2048); }

// X <- X L^-T where only the lower triangle of the (square, n x n) result is wanted: column
// block 2 is then only needed on rows >= n1, which removes ~43% of the flops of a full solve.
static void trsm_rlt_lower(hipStream_t s, int n, const double *L, int ldl, const double *invdiag,
                           double *X, int ldx, double *work) {
  if (n <= 0) return;
  if (n <= NB || leaf256(invdiag, n)) { trsm_rlt(s, n, n, L, ldl, invdiag, X, ldx, work); return; }
  const int n1 = split_t(n), n2 = n - n1;
  const double *L21 = L + n1, *L22 = L + (size_t)n1 + (size_t)n1 * ldl;
  double *X21 = X + n1, *X22 = X + (size_t)n1 + (size_t)n1 * ldx;
  trsm_rlt(s, n, n1, L, ldl, invdiag, X, ldx, work);                          // all rows of block column 1
  gemm(s, false, true, n2, n2, n1, -1.0, X21, ldx, L21, ldl, 1.0, X22, ldx);  // rows >= n1 only
  trsm_rlt_lower(s, n2, L22, ldl, invdiag + (size_t)(n1 / NB) * NB * NB, X22, ldx, work);
}

// Below this order a block is reduced by two triangular solves (1.57 n^3 flops but the fewest
// and largest GEMMs); above it the blocked recursion (n^3).  Measured on MI355X at N = 16384
// (sygst stage), round 1: no recursion 0.133 s, threshold 2048 -> 0.115 s, 1024 -> 0.120 s, 512 -> 0.126 s;
// with the 16-byte-load GEMM of round 2: 1024 -> 0.110, 2048 -> 0.103, 4096 -> 0.099, 8192 -> 0.099, none 0.115.
static int sygst_direct() { return 4096; }      // (the scan above; the tuning switch went in round 6)

// Recursive blocked DSYGST(itype = 1, 'L'):  with A = [A11 .; A21 A22], L = [L11 0; L21 L22]
//   C11 = sygst(A11, L11)
//   A21 <- A21 L11^-T;  A21 <- A21 - 1/2 L21 C11
//   A22 <- A22 - A21 L21^T - L21 A21^T            (SYR2K, lower)
//   A21 <- A21 - 1/2 L21 C11;  A21 <- L22^-1 A21
//   C22 = sygst(A22, L22)
// ~n^3 flops in total (the count of the blocked LAPACK routine) down to blocks of order 2048,
// every update a GEMM of order n/2, n/4, ...  scratch: n^2/2 doubles (the symmetric C11 in full storage + the product L21 C11).
static void sygst_rec(hipStream_t s, int n, double *A, int lda, const double *L, int ldl,
                      const double *invdiag, double *work, double *scratch) {
  if (n <= 0) return;
  if (n <= sygst_direct()) {
    // small block: the recursion would drown in tiny launches; a full left solve plus a right
    // solve restricted to the lower triangle (1.57 n^3 flops, but few and larger GEMMs)
    symmetrize_lower(s, n, A, lda);
    trsm_lln(s, n, n, L, ldl, invdiag, A, lda, work);
    trsm_rlt_lower(s, n, L, ldl, invdiag, A, lda, work);
    return;
  }
  const int n1 = split_t(n), n2 = n - n1;
  double *A21 = A + n1, *A22 = A + (size_t)n1 + (size_t)n1 * lda;
  const double *L21 = L + n1, *L22 = L + (size_t)n1 + (size_t)n1 * ldl;
  const double *inv2 = invdiag + (size_t)(n1 / NB) * NB * NB;
  sygst_rec(s, n1, A, lda, L, ldl, invdiag, work, scratch);
  trsm_rlt(s, n2, n1, L, ldl, invdiag, A21, lda, work);
  double *C11 = scratch, *M = scratch + (size_t)n1 * n1;          // n1 x n1, n2 x n1
  hipLaunchKernelGGL(full_from_lower_kernel, grid_mn(n1, n1), dim3(256), 0, s, n1, A, lda, C11, n1);
  gemm(s, false, false, n2, n1, n1, 1.0, L21, ldl, C11, n1, 0.0, M, n2);
  hipLaunchKernelGGL(axpy_matrix_kernel, grid_mn(n2, n1), dim3(256), 0, s, n2, n1, -0.5, M, n2, A21, lda);
  // A22 -= A21 L21^T + L21 A21^T.  As two lower-only products each loses a fifth to its tail (2080 tiles of 2 ms on
  // 512 workgroup slots at n2 = 8192: 45 TFLOP/s); the same flops as ONE full product P = A21 L21^T (68 TFLOP/s, into
  // the space of C11, which is spent) and a pass A22 -= P + P^T over the lower triangle.
  static int fold = -1;
  if (fold < 0) { const char *e = getenv("EK_SYGST_FOLD"); fold = e ? atoi(e) : 1; }
  if (fold && n2 <= n1) {
    double *P = C11;
    gemm(s, false, true, n2, n2, n1, 1.0, A21, lda, L21, ldl, 0.0, P, n2);
    hipLaunchKernelGGL(syr2k_fold_kernel, dim3(ceil_div(n2, 32), ceil_div(n2, 32)), dim3(256), 0, s, n2, P, n2, A22, lda);
  } else {
    gemm(s, false, true, n2, n2, n1, -1.0, A21, lda, L21, ldl, 1.0, A22, lda, /*lower_only=*/true);
    gemm(s, false, true, n2, n2, n1, -1.0, L21, ldl, A21, lda, 1.0, A22, lda, /*lower_only=*/true);
  }
  hipLaunchKernelGGL(axpy_matrix_kernel, grid_mn(n2, n1), dim3(256), 0, s, n2, n1, -0.5, M, n2, A21, lda);
  trsm_lln(s, n2, n1, L22, ldl, inv2, A21, lda, work);
  sygst_rec(s, n2, A22, lda, L22, ldl, inv2, work, scratch);
}

// A <- L^-1 A L^-T, lower triangles in and out (as PDSYGST 'L').  scratch: >= max(n^2/2 + n, 2*128^2) doubles.
void sygst_lower(hipStream_t s, int n, double *A, int lda, const double *L, int ldl,
                 const double *invdiag, double *work, double *scratch) {
  if (n <= 0) return;
  sygst_rec(s, n, A, lda, L, ldl, invdiag, work, scratch);
}

namespace {
inline size_t al256c(size_t b) { return (b + 255) & ~(size_t)255; }
struct PotrfDistLayout {
  int NRB, maxb;
  size_t off_pbuf, off_pbuf2, off_infos, off_infod, off_offs, off_dims, total;
  PotrfDistLayout(int n, int ld, int P) {
    NRB = ceil_div(n > 0 ? n : 1, NB); maxb = ceil_div(NRB, P) + 1;
    size_t o = 0;
    off_pbuf = o; o += al256c((size_t)(2 * NB * NB + (size_t)ld * NB) * 8);
    off_pbuf2 = o; o += al256c((size_t)(2 * NB * NB + (size_t)ld * NB) * 8);     // (look-ahead of the team form: the next strip's message)
    off_infos = o; o += al256c((size_t)NRB * sizeof(int));
    off_infod = o; o += al256c((size_t)NRB * 8);
    off_offs = o; o += al256c((size_t)NRB * maxb * 3 * sizeof(long long));
    off_dims = o; o += al256c((size_t)NRB * maxb * 3 * sizeof(int));
    total = o;
  }
};

// per step k and owned strip j > k: A = B = panel rows from j*128 (inside strip k), C = block (j, j)
__global__ void potrf_table_kernel(int n, int ldb, int P, int rank, int NRB, int maxb, long long *offs, int *dims) {
  const int k = blockIdx.x, b = threadIdx.x;
  if (b >= maxb) return;
  const int jf = k + 1 + ((rank - (k + 1)) % P + P) % P;
  const int j = jf + b * P;
  int M = 0, N = 0;
  const long long r0 = (long long)j * NB;
  if (j < NRB && r0 < n) { M = n - (int)r0; N = (M < NB) ? M : NB; }
  const int nbk = (n - k * NB < NB) ? n - k * NB : NB;
  const size_t e = (size_t)k * maxb + b;
  offs[3 * e] = r0 + (long long)k * NB * ldb; offs[3 * e + 1] = offs[3 * e]; offs[3 * e + 2] = r0 * ((long long)ldb + 1);
  dims[3 * e] = M; dims[3 * e + 1] = N; dims[3 * e + 2] = nbk;
}
__global__ void info_to_double_kernel(int nrb, const int *infos, double *out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nrb) out[i] = (double)infos[i];
}
__global__ void first_info_kernel(int nrb, const double *v, int *d_info) {
  if (threadIdx.x != 0 || blockIdx.x != 0 || *d_info != 0) return;
  for (int i = 0; i < nrb; ++i)
    if (v[i] != 0.0) { *d_info = (int)v[i]; return; }
}
}  // namespace

size_t potrf_dist_work_bytes(int n, int ld, int nranks) { return PotrfDistLayout(n, ld, nranks > 0 ? nranks : 1).total; }

// optional timing of the team form (tools/team_timing.py): HIP events around every owner's chain (factor + invert the
// diagonal block, solve the panel) and around every "rest of the update" section, on the streams they run on
namespace {
struct PotrfProf {
  bool on = false;
  std::vector<hipEvent_t> ev;      // pairs (begin, end)
  std::vector<int> kind;           // per pair: 0 = chain, 1 = rest of the update
  hipEvent_t mark(hipStream_t st) {
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    (void)hipEventRecord(e, st);
    ev.push_back(e);
    return e;
  }
};
PotrfProf g_pprof;
int g_potrf_la = -1;               // look-ahead of the team form: 0 off, 1 on, -1 default (on)
}  // namespace
void potrf_dist_set_lookahead(int on) { g_potrf_la = on; }
void potrf_dist_profile(bool on) {
  for (auto &e : g_pprof.ev) (void)hipEventDestroy(e);
  g_pprof.ev.clear(); g_pprof.kind.clear();
  g_pprof.on = on;
}
// after the streams have been synchronised (a rehearsal WITHOUT look-ahead: the sections do not overlap): seconds[0] = all
// chains, [1] = all rest-of-update sections, [2] = the first chain + sum over the strips of max(chain of strip k + 1,
// (update of strip k) / P): what the two cost a rank of a real team of P when the chain (one rank, the others wait for
// its broadcast) runs beside the update (every rank its 1 / P)
void potrf_dist_profile_collect(double *seconds, int P) {
  seconds[0] = seconds[1] = seconds[2] = 0.0;
  double last_update = -1.0;
  for (size_t q = 0; q < g_pprof.kind.size(); ++q) {
    float ms = 0.f;
    if (!(g_pprof.ev[2 * q] && g_pprof.ev[2 * q + 1] && hipEventElapsedTime(&ms, g_pprof.ev[2 * q], g_pprof.ev[2 * q + 1]) == hipSuccess)) continue;
    const double t = ms * 1e-3;
    seconds[g_pprof.kind[q]] += t;
    if (g_pprof.kind[q] == 1) {
      if (last_update >= 0.0) seconds[2] += last_update / (P > 0 ? P : 1);
      last_update = t;
    } else {
      const double u = last_update >= 0.0 ? last_update / (P > 0 ? P : 1) : 0.0;
      seconds[2] += (t > u) ? t : u;
      last_update = -1.0;
    }
  }
  if (last_update >= 0.0) seconds[2] += last_update / (P > 0 ? P : 1);
  potrf_dist_profile(g_pprof.on);
}

// Look-ahead (round 6; the single-GPU form and the team's dense -> band stage have had it): once strip k is in place
// everywhere, the owner of strip k + 1 updates THAT strip first; its chain (factor + invert the diagonal block, solve the
// panel), the broadcast of [inverse | block | panel] and the unpacking on every member then run on the second stream s2
// while all members apply strip k to the rest of their strips on s.  The message buffer is double-buffered for that.
// Until round 5 the owner's chain (0.1 - 0.15 ms, 256 of them at N = 32768) stood between every two updates while P - 1
// ranks waited.  Same arithmetic on every element as without (s2 == nullptr or potrf_dist_set_lookahead(0)): same bits.
void potrf_lower_dist(hipStream_t s, hipStream_t s2, int n, int nmem, const PotrfMember *mem, const SytrdExchange &x) {
  set_attrs();
  if (n <= 0 || nmem <= 0 || nmem > kMaxTeam) return;
  const int P = x.nranks;
  const PotrfDistLayout Ly(n, mem[0].ldb, P);
  const int NRB = Ly.NRB;
  static bool evs = false;
  static hipEvent_t evA[2], evB[2];
  if (!evs) {
    for (int q = 0; q < 2; ++q) {
      (void)hipEventCreateWithFlags(&evA[q], hipEventDisableTiming);
      (void)hipEventCreateWithFlags(&evB[q], hipEventDisableTiming);
    }
    evs = true;
  }
  const bool la_on = s2 != nullptr && g_potrf_la != 0 && NRB > 1;
  double *pbuf[2][kMaxTeam], *infod[kMaxTeam];
  int *infos[kMaxTeam]; long long *offs[kMaxTeam]; int *dims[kMaxTeam];
  for (int m = 0; m < nmem; ++m) {
    char *w = (char *)mem[m].work;
    pbuf[0][m] = (double *)(w + Ly.off_pbuf); pbuf[1][m] = (double *)(w + Ly.off_pbuf2); infos[m] = (int *)(w + Ly.off_infos);
    infod[m] = (double *)(w + Ly.off_infod); offs[m] = (long long *)(w + Ly.off_offs); dims[m] = (int *)(w + Ly.off_dims);
    (void)hipMemsetAsync(infos[m], 0, (size_t)NRB * sizeof(int), s);
    hipLaunchKernelGGL(potrf_table_kernel, dim3(NRB), dim3(round_up(Ly.maxb, 64)), 0, s, n, mem[m].ldb, P,
                       mem[m].rank, NRB, Ly.maxb, offs[m], dims[m]);
  }
  // strip k on stream sp through message buffer `buf`: the owner factors it and solves the panel; one broadcast; every
  // member stores the inverse, the block and the panel (L and the block inverses end up complete on all ranks)
  auto issue_strip = [&](hipStream_t sp, int k, int buf) {
    const int off = k * NB, nbk = (n - off < NB) ? n - off : NB, mrows = n - off - nbk;
    const int owner = k % P;
    const size_t count = (size_t)2 * NB * NB + (size_t)mrows * nbk;
    for (int m = 0; m < nmem; ++m) {
      if (mem[m].rank != owner) continue;
      const PotrfMember &M = mem[m];
      double *Bd = M.B + (size_t)off + (size_t)off * M.ldb;
      double *pinv = pbuf[buf][m], *pdiag = pbuf[buf][m] + NB * NB, *ppan = pbuf[buf][m] + 2 * NB * NB;
      hipEvent_t e0 = g_pprof.on ? g_pprof.mark(sp) : nullptr;
      hipLaunchKernelGGL(potrf_diag_kernel, dim3(1), dim3(256), kDiagLds, sp, nbk,
                         Bd, M.ldb, pinv, infos[m] + k, off);
      copy_matrix(sp, nbk, nbk, Bd, M.ldb, pdiag, NB);
      if (mrows > 0) gemm(sp, false, true, mrows, nbk, nbk, 1.0, Bd + nbk, M.ldb, pinv, NB, 0.0, ppan, mrows);
      if (e0) { g_pprof.mark(sp); g_pprof.kind.push_back(0); }
    }
    size_t zoffs[kMaxTeam], cnts[kMaxTeam];
    for (int r = 0; r < P; ++r) { zoffs[r] = 0; cnts[r] = (r == owner) ? count : 0; }
    x.allgatherv(sp, nmem, mem[0].rank, pbuf[buf], zoffs, cnts, P, x.user);
    for (int m = 0; m < nmem; ++m) {
      const PotrfMember &M = mem[m];
      double *Bd = M.B + (size_t)off + (size_t)off * M.ldb;
      const double *pinv = pbuf[buf][m], *pdiag = pbuf[buf][m] + NB * NB, *ppan = pbuf[buf][m] + 2 * NB * NB;
      copy_matrix(sp, NB, NB, pinv, NB, M.invdiag + (size_t)k * NB * NB, NB);
      if (M.rank != owner) copy_matrix(sp, nbk, nbk, pdiag, NB, Bd, M.ldb);
      if (mrows > 0) copy_matrix(sp, mrows, nbk, ppan, mrows, Bd + nbk, M.ldb);
    }
  };
  // member m applies strip k to its own strips: entries first .. of its table for step k (entry 0 = its first owned strip
  // beyond k)
  auto update_strips = [&](int m, int k, int first, int count) {
    const PotrfMember &M = mem[m];
    const int nbk = (n - k * NB < NB) ? n - k * NB : NB;
    const int jf = k + 1 + ((M.rank - (k + 1)) % P + P) % P;
    if (count <= 0 || jf + first * P >= NRB || (jf + first * P) * NB >= n) return;
    GemmDesc g{};
    g.M = n - (jf + first * P) * NB; g.N = NB; g.K = nbk; g.transA = false; g.transB = true; g.alpha = -1.0; g.beta = 1.0;
    g.A = M.B; g.lda = M.ldb; g.B = M.B; g.ldb = M.ldb; g.C = M.B; g.ldc = M.ldb;
    g.batch = count; g.lower_only = true;
    g.d_offs = offs[m] + ((size_t)k * Ly.maxb + first) * 3; g.d_dims = dims[m] + ((size_t)k * Ly.maxb + first) * 3;
    gemm(s, g);
  };
  auto owned_after = [&](int m, int k) {                 // strips member m owns beyond strip k
    const int jf = k + 1 + ((mem[m].rank - (k + 1)) % P + P) % P;
    return (jf < NRB && jf * NB < n) ? ceil_div(NRB - jf, P) : 0;
  };

  issue_strip(s, 0, 0);
  bool waited = true;
  for (int k = 0; k < NRB; ++k) {
    const int cur = k & 1;
    if (!waited) (void)hipStreamWaitEvent(s, evB[cur], 0);
    const int mrows = n - k * NB - ((n - k * NB < NB) ? n - k * NB : NB);
    if (mrows <= 0) break;
    const bool has_next = k + 1 < NRB && (k + 1) * NB < n;
    if (la_on && has_next) {
      const int owner_next = (k + 1) % P;
      for (int m = 0; m < nmem; ++m)                      // the next strip first, on its owner
        if (mem[m].rank == owner_next) update_strips(m, k, 0, 1);
      (void)hipEventRecord(evA[cur], s);
      hipEvent_t e0 = g_pprof.on ? g_pprof.mark(s) : nullptr;
      for (int m = 0; m < nmem; ++m) {
        const bool own = mem[m].rank == owner_next;
        update_strips(m, k, own ? 1 : 0, owned_after(m, k) - (own ? 1 : 0));
      }
      if (e0) { g_pprof.mark(s); g_pprof.kind.push_back(1); }
      (void)hipStreamWaitEvent(s2, evA[cur], 0);
      issue_strip(s2, k + 1, cur ^ 1);
      (void)hipEventRecord(evB[cur ^ 1], s2);
      waited = false;
    } else {
      hipEvent_t e0 = g_pprof.on ? g_pprof.mark(s) : nullptr;
      for (int m = 0; m < nmem; ++m) update_strips(m, k, 0, owned_after(m, k));
      if (e0) { g_pprof.mark(s); g_pprof.kind.push_back(1); }
      if (has_next) issue_strip(s, k + 1, cur ^ 1);
      waited = true;
    }
  }
  // (the loop leaves at the last strip, whose chain stream s has waited for at the top of that iteration)
  // first failing pivot, known to every rank
  for (int m = 0; m < nmem; ++m)
    hipLaunchKernelGGL(info_to_double_kernel, dim3(ceil_div(NRB, 256)), dim3(256), 0, s, NRB, infos[m], infod[m]);
  x.allreduce(s, nmem, infod, (size_t)NRB, x.user);
  for (int m = 0; m < nmem; ++m)
    hipLaunchKernelGGL(first_info_kernel, dim3(1), dim3(64), 0, s, NRB, infod[m], mem[m].d_info);
}

// Single-GPU right-looking Cholesky with look-ahead: the chain "factor + invert the 128x128 diagonal
// block in LDS (one workgroup, ~0.2 ms), solve the panel below it" of block column k+1 runs on a
// second stream while the first stream is still applying block column k to the rest of the
// matrix, so the latency-bound chain (a third of the recursive form's time) hides behind the GEMMs.
size_t potrf_rl_work_bytes(int n, int ld) { return PotrfDistLayout(n, ld, 1).total; }

void potrf_lower_rl(hipStream_t s, hipStream_t s2, int n, double *B, int ldb, double *invdiag, int *d_info,
                    void *work) {
  set_attrs();
  if (n <= 0) return;
  const PotrfDistLayout Ly(n, ldb, 1);
  const int NRB = Ly.NRB;
  char *w = (char *)work;
  double *ppan = (double *)(w + Ly.off_pbuf);
  int *infos = (int *)(w + Ly.off_infos);
  double *infod = (double *)(w + Ly.off_infod);
  long long *offs = (long long *)(w + Ly.off_offs);
  int *dims = (int *)(w + Ly.off_dims);
  static hipEvent_t evU[2], evP[2], evFork;
  static bool ev_ready = false;
  if (!ev_ready) {
    for (int i = 0; i < 2; ++i) {
      (void)hipEventCreateWithFlags(&evU[i], hipEventDisableTiming);
      (void)hipEventCreateWithFlags(&evP[i], hipEventDisableTiming);
    }
    (void)hipEventCreateWithFlags(&evFork, hipEventDisableTiming);
    ev_ready = true;
  }
  (void)hipMemsetAsync(infos, 0, (size_t)NRB * sizeof(int), s);
  hipLaunchKernelGGL(potrf_table_kernel, dim3(NRB), dim3(round_up(Ly.maxb, 64)), 0, s, n, ldb, 1, 0, NRB, Ly.maxb,
                     offs, dims);
  auto factor_panel = [&](hipStream_t st, int k) {
    const int off = k * NB, nbk = (n - off < NB) ? n - off : NB, mrows = n - off - nbk;
    double *Bd = B + (size_t)off + (size_t)off * ldb;
    double *inv = invdiag + (size_t)k * NB * NB;
    hipLaunchKernelGGL(potrf_diag_kernel, dim3(1), dim3(256), kDiagLds, st, nbk, Bd,
                       ldb, inv, infos + k, off);
    if (mrows > 0) {
      gemm(st, false, true, mrows, nbk, nbk, 1.0, Bd + nbk, ldb, inv, NB, 0.0, ppan, mrows);
      copy_matrix(st, mrows, nbk, ppan, mrows, Bd + nbk, ldb);
    }
  };
  // Two levels of blocking.  An OUTER panel of `ob` block columns (512 columns) is factored on stream s2 by the
  // right-looking steps of width 128 restricted to the panel (diagonal kernel, block column through the inverse
  // of the diagonal block, rank-128 update of the panel's remaining block columns); the matrix behind the panel
  // then takes ONE rank-512 update on stream s -- the next panel's columns first, so that its factorisation
  // runs beside the rest of the update.  With rank-128 updates of the whole trailing matrix (ob = 1, the earlier
  // form) every 128 columns read and wrote the whole trailing triangle: 92 GB of C traffic at N = 16384.
  static int ob = -1;
  if (ob < 0) { const char *e = getenv("EK_POTRF_OB"); ob = e ? atoi(e) : 4; if (ob < 1) ob = 1; }
  const int NOB = ceil_div(NRB, ob);
  auto factor_outer = [&](hipStream_t st, int o) {
    const int k0 = o * ob, k1 = (k0 + ob < NRB) ? k0 + ob : NRB;
    for (int k = k0; k < k1; ++k) {
      factor_panel(st, k);
      const int cnt = k1 - 1 - k;                                 // block columns of the panel behind k
      if (cnt > 0) {
        const int nbk = (n - k * NB < NB) ? n - k * NB : NB;
        GemmDesc g{};
        g.N = NB; g.K = nbk; g.transA = false; g.transB = true; g.alpha = -1.0; g.beta = 1.0;
        g.A = B; g.lda = ldb; g.B = B; g.ldb = ldb; g.C = B; g.ldc = ldb; g.lower_only = true;
        g.even_offs = true;                                       // offsets are multiples of 128 rows and columns
        g.M = n - (k + 1) * NB; g.batch = cnt;
        g.d_offs = offs + (size_t)k * Ly.maxb * 3; g.d_dims = dims + (size_t)k * Ly.maxb * 3;
        gemm(st, g);
      }
    }
  };
  (void)hipEventRecord(evFork, s);
  (void)hipStreamWaitEvent(s2, evFork, 0);
  factor_outer(s2, 0);
  (void)hipEventRecord(evP[0], s2);
  for (int o = 0; o + 1 < NOB; ++o) {
    const int c0 = o * ob * NB, c1 = (o + 1) * ob * NB, kw = c1 - c0;       // c1 < n here
    const int m2 = n - c1;
    const int wnext = (m2 < ob * NB) ? m2 : ob * NB;                        // width of the next outer panel
    (void)hipStreamWaitEvent(s, evP[o & 1], 0);                             // panel o is factored
    const double *P = B + (size_t)c1 + (size_t)c0 * ldb;                    // L(c1:, c0:c1)
    double *C = B + (size_t)c1 + (size_t)c1 * ldb;
    gemm(s, false, true, m2, wnext, kw, -1.0, P, ldb, P, ldb, 1.0, C, ldb, true);
    (void)hipEventRecord(evU[(o + 1) & 1], s);
    // (host order: the rest of the update is submitted before the sixteen launches of the next panel's chain, behind
    // which it would start ~0.1 ms late)
    if (m2 > wnext)
      gemm(s, false, true, m2 - wnext, m2 - wnext, kw, -1.0, P + wnext, ldb, P + wnext, ldb, 1.0,
           C + (size_t)wnext + (size_t)wnext * ldb, ldb, true);
    (void)hipStreamWaitEvent(s2, evU[(o + 1) & 1], 0);
    factor_outer(s2, o + 1);
    (void)hipEventRecord(evP[(o + 1) & 1], s2);
  }
  (void)hipStreamWaitEvent(s, evP[(NOB - 1) & 1], 0);
  hipLaunchKernelGGL(info_to_double_kernel, dim3(ceil_div(NRB, 256)), dim3(256), 0, s, NRB, infos, infod);
  hipLaunchKernelGGL(first_info_kernel, dim3(1), dim3(64), 0, s, NRB, infod, d_info);
}

size_t sygst_dist_scratch_doubles(int n, int ld, int nranks) {
  const int NRB = ceil_div(n > 0 ? n : 1, NB);
  return (size_t)ld * NB * ceil_div(NRB, nranks > 0 ? nranks : 1);
}

void sygst_lower_dist(hipStream_t s, int n, int nmem, const SygstMember *mem, const SytrdExchange &x) {
  if (n <= 0 || nmem <= 0 || nmem > kMaxTeam) return;
  const int P = x.nranks;
  const int NRB = ceil_div(n, NB);
  const int wblk = ceil_div(NRB, P) * NB;       // width of a rank's contiguous column block
  // step 1: Y(:, C_r) = L^-1 A(:, C_r)
  size_t offs[kMaxTeam], counts[kMaxTeam];
  for (int r = 0; r < P; ++r) {
    long long c0 = (long long)r * wblk, c1 = c0 + wblk;
    if (c0 > n) c0 = n;
    if (c1 > n) c1 = n;
    offs[r] = (size_t)c0 * mem[0].lda; counts[r] = (size_t)(c1 - c0) * mem[0].lda;
  }
  double *bufs[kMaxTeam];
  for (int m = 0; m < nmem; ++m) {
    const SygstMember &M = mem[m];
    bufs[m] = M.A;
    symmetrize_lower(s, n, M.A, M.lda);
    const int c0 = (int)(offs[M.rank] / M.lda), w = (int)(counts[M.rank] / M.lda);
    if (w > 0) trsm_lln(s, n, w, M.L, M.ldl, M.invdiag, M.A + (size_t)c0 * M.lda, M.lda, M.work);
  }
  x.allgatherv(s, nmem, mem[0].rank, bufs, offs, counts, P, x.user);
  // step 2: A'(:, S) = L^-1 (Y(S, :))^T for the owned strips
  for (int m = 0; m < nmem; ++m) {
    const SygstMember &M = mem[m];
    int wtot = 0;
    for (int S = M.rank; S < NRB; S += P) {
      const int cols = (n - S * NB < NB) ? n - S * NB : NB;
      transpose_rows(s, n, cols, M.A, M.lda, S * NB, M.scratch + (size_t)wtot * M.lda, M.lda);
      wtot += cols;
    }
    if (wtot == 0) continue;
    trsm_lln(s, n, wtot, M.L, M.ldl, M.invdiag, M.scratch, M.lda, M.work);
    wtot = 0;
    for (int S = M.rank; S < NRB; S += P) {
      const int cols = (n - S * NB < NB) ? n - S * NB : NB;
      copy_matrix(s, n, cols, M.scratch + (size_t)wtot * M.lda, M.lda, M.A + (size_t)S * NB * M.lda, M.lda);
      wtot += cols;
    }
  }
}

}  // namespace ek
