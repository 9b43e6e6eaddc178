// ek_block64.h -- 64x64 building blocks for kernels of ONE workgroup of 256 threads (4 waves):
// products on the matrix cores and the serial factorisations (Cholesky, triangular inverse, LU
// without pivoting, unit-triangular solve) of 64x64 LDS images.  Device code only; used by the
// panel chain of ek_sy2sb.hip and by the diagonal blocks of the Cholesky factorisation (ek_chol.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace ek {
namespace b64 {

typedef double double4_t __attribute__((ext_vector_type(4)));

constexpr int SB = 64;       // order of a block
constexpr int LD = 66;       // leading dimension of 64x64 LDS images, row-major s[r * LD + c]
constexpr int IMG = SB * LD; // doubles per image

// C = op(A) op(B), all 64x64 LDS images, on the matrix cores, by the 4 waves of the workgroup:
// wave w owns rows 16 w .. 16 w + 15 of C.  Callers synchronise before and after.
// out_g != nullptr: C goes to global memory (column-major, ld 64), row i scaled by rs[i] if rs.
__device__ __forceinline__ void mm64(const double *sA, bool ta, const double *sB, bool tb, double *sC,
                                     double *out_g = nullptr, const double *rs = nullptr) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l15 = lane & 15, l4 = lane >> 4;
  const int i0 = 16 * wave;
  double4_t acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] = (double4_t){0.0, 0.0, 0.0, 0.0};
  for (int kk = 0; kk < SB; kk += 4) {
    const double x = ta ? sA[(kk + l4) * LD + i0 + l15] : sA[(i0 + l15) * LD + kk + l4];
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) {
      const double y = tb ? sB[(16 * jt + l15) * LD + kk + l4] : sB[(kk + l4) * LD + 16 * jt + l15];
      acc[jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc[jt], 0, 0, 0);
    }
  }
#pragma unroll
  for (int jt = 0; jt < 4; ++jt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = i0 + l4 + 4 * r, j = 16 * jt + l15;
      if (out_g) out_g[i + SB * j] = rs ? rs[i] * acc[jt][r] : acc[jt][r];
      else sC[i * LD + j] = acc[jt][r];
    }
}

// The 64-step factorisations below are the serial part of a panel.  They run on all 256 threads of
// the workgroup with the 64x64 matrix in REGISTERS: thread (w = t >> 6, c = t & 63) holds the entries
// (w + 4 i, c), i = 0 .. 15.  A step publishes one row (and, for LU, one column) through a
// double-buffered LDS line, costs ONE workgroup barrier, and updates the trailing block with 16
// register FMAs per thread (right-looking).  After every four steps the registers shift by one, so
// that the row being finished is always register 0 (15 when going upwards) and the body is the same
// for every block of four rows: the code stays a few hundred instructions (a fully unrolled
// 64-step body does not fit the instruction cache).  Slots that have shifted out keep being
// "updated" with values nobody reads.  Finished rows go straight to the LDS image.
// The left-looking single-wave forms these replace spent ~1650 cycles per step in LDS reads.
// srow / scol: kLine doubles each (two lines of 128: row indices of shifted-out slots run past 63).
constexpr int kLine = 2 * 128;

__device__ __forceinline__ double fast_rcp(double p) {     // |p| well inside the normal range
  double r = __builtin_amdgcn_rcp(p);
  double e = __builtin_fma(-p, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-p, r, 1.0);
  return __builtin_fma(r, e, r);
}

// G = R^T R, in place: image -> upper factor (zeros below; only the upper triangle of G is used).
// Returns (in all threads) the index of the first pivot that is not positive, -1 if there is none.
__device__ __forceinline__ int chol64_upper_wg(double *sG, double *srow) {
  const int t = threadIdx.x, c = t & 63, w = t >> 6;
  double a[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = sG[(w + 4 * i) * LD + c];
  int fail = -1;
  __syncthreads();
#pragma unroll 1
  for (int jb = 0; jb < 16; ++jb) {
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int j = 4 * jb + jj;
      double *row = srow + jj % 2 * 128;
      if (w == jj) row[c] = a[0];
      __syncthreads();
      double d = row[j];
      if (!(d > 0.0) || !(d < 1.7e308)) { fail = (fail < 0) ? j : fail; d = 1.0; }
      const double rinv = rsqrt(d);
      const double rc = row[c] * rinv;
      if (w == jj) sG[j * LD + c] = (c > j) ? rc : (c == j ? d * rinv : 0.0);
#pragma unroll
      for (int i = 0; i < 16; ++i) a[i] -= (row[w + 4 * (i + jb)] * rinv) * rc;
    }
#pragma unroll
    for (int i = 0; i < 15; ++i) a[i] = a[i + 1];
  }
  __syncthreads();
  return fail;
}

// X = R^-1 for upper triangular R (only its upper triangle is read); every entry of sX is written.
__device__ __forceinline__ void triinv64_upper_wg(const double *sR, double *sX, double *srow) {
  const int t = threadIdx.x, c = t & 63, w = t >> 6;
  double x[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) x[i] = (w + 4 * i == c) ? 1.0 : 0.0;
#pragma unroll 1
  for (int kb = 15; kb >= 0; --kb) {
#pragma unroll
    for (int kk = 3; kk >= 0; --kk) {
      const int k = 4 * kb + kk;
      double *row = srow + kk % 2 * 128;
      if (w == kk) {
        const double xk = (c >= k) ? x[15] * fast_rcp(sR[k * LD + k]) : 0.0;
        row[c] = xk;
        sX[k * LD + c] = xk;
      }
      __syncthreads();
      const double xc = row[c];
#pragma unroll
      for (int i = 0; i < 16; ++i) x[i] -= sR[((w + 4 * (i + kb - 15)) & 63) * LD + k] * xc;
    }
#pragma unroll
    for (int i = 15; i > 0; --i) x[i] = x[i - 1];
  }
  __syncthreads();
}

// LU of (A - S) without pivoting, S(j,j) = -sign(pivot) so that |pivot| >= 1; in place (strictly
// lower = L, upper = U), signs to s_sign.
__device__ __forceinline__ void lu64_signed_wg(double *sD, double *s_sign, double *srow, double *scol) {
  const int t = threadIdx.x, c = t & 63, w = t >> 6;
  double a[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = sD[(w + 4 * i) * LD + c];
  __syncthreads();
#pragma unroll 1
  for (int jb = 0; jb < 16; ++jb) {
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int j = 4 * jb + jj;
      double *row = srow + jj % 2 * 128, *col = scol + jj % 2 * 128;
      if (w == jj) row[c] = a[0];
      if (c == j) {
#pragma unroll
        for (int i = 0; i < 16; ++i) col[w + 4 * (i + jb)] = a[i];
      }
      __syncthreads();
      double piv = row[j];
      const double sj = (piv >= 0.0) ? -1.0 : 1.0;
      piv -= sj;
      const double pinv = fast_rcp(piv);
      if (w == jj) {          // row j is final: L entries kept in the registers left of the diagonal, U right of it
        sD[j * LD + c] = (c == j) ? piv : a[0];
        if (c == j) s_sign[j] = sj;
      }
      const double uc = (c > j) ? row[c] : 0.0;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const double l = col[w + 4 * (i + jb)] * pinv;
        a[i] -= l * uc;
        a[i] = (c == j) ? l : a[i];
      }
    }
#pragma unroll
    for (int i = 0; i < 15; ++i) a[i] = a[i + 1];
  }
  __syncthreads();
}

// Y = L^-1 C for the UNIT lower triangular L held strictly below the diagonal of sL, with
// C(r, c) = -U(c, r) sign(r) for c <= r (U = upper part of sL): Y(j, i) = T(i, j) of T = -U S L^-T.
__device__ __forceinline__ void tsolve64_wg(const double *sL, const double *s_sign, double *sY, double *srow) {
  const int t = threadIdx.x, c = t & 63, w = t >> 6;
  double y[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int r = w + 4 * i;
    y[i] = (c <= r) ? -sL[c * LD + r] * s_sign[r] : 0.0;
  }
#pragma unroll 1
  for (int kb = 0; kb < 16; ++kb) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int k = 4 * kb + kk;
      double *row = srow + kk % 2 * 128;
      if (w == kk) { row[c] = y[0]; sY[k * LD + c] = y[0]; }
      __syncthreads();
      const double yc = row[c];
#pragma unroll
      for (int i = 0; i < 16; ++i) y[i] -= sL[((w + 4 * (i + kb)) & 63) * LD + k] * yc;
    }
#pragma unroll
    for (int i = 0; i < 15; ++i) y[i] = y[i + 1];
  }
  __syncthreads();
}

}  // namespace b64
}  // namespace ek
