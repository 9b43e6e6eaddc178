// ek_sy2sb.hip -- stage 1 of the two-stage tridiagonalisation: dense symmetric -> symmetric band
// (half bandwidth SBW = 64), A = Q1 Bd Q1^T, entirely on the matrix cores.
//
// Together with ek_sb2st.hip (band -> tridiagonal by bulge chasing, and the back-transformation of
// its reflectors) this replaces the ONE-stage PDSYTRD of solver_scalapack_all.f90:59 inside the
// whole-path call for large orders.  Why: the one-stage reduction reads the trailing matrix once
// per Householder column (4 n^3 / 3 bytes, HBM-bound: 0.73 s of the 1.8 s solve at N = 16384 even at
// the full 8 TB/s); the two-stage form reads it once per PANEL of 64 columns and does all O(n^3)
// work as GEMM-shaped products.  The results contract of the path (eigenvalues, eigenvectors) is
// unchanged; the stage-level entry ek_hip_sytrd keeps the one-stage PDSYTRD semantics.
//
// Per panel p (columns c0 = 64 p .. c0+63, rows r0 = c0+64 .. n-1, m = n - r0):
//   1. panel factorisation  P = A(r0:n, c0:c0+64) = Q R  with Q = I - V T V^T:
//        m > 192 : CholeskyQR2 (two Gram + triangular-solve passes: everything is a tall-skinny
//                  GEMM) followed by Householder reconstruction (Ballard et al. 2014: an LU of the
//                  top 64x64 block of Q - S without pivoting gives V, T and the signs S), checked on
//                  the device (Cholesky pivots, ||Q1^T Q1 - I||): a panel CholeskyQR2 cannot handle
//                  (rank deficient / cond > 1e7) sets THAT PANEL's flag and is factored by Householder
//                  reflections inside the stage (house_tall_kernel and its companions: the per-panel rescue);
//                  bits 8.. of *d_flag count the panels that took it, the low byte stays 0;
//        m <= 192: Householder QR of the panel inside one workgroup's LDS (handles any rank).
//   2. Y = A22 V            (SYMM on the lower triangle, split-K partial sums)
//   3. W = Y T - 1/2 V (T^T V^T Y T)
//   4. A22 -= W V^T + V W^T (one GEMM with K = 128 on the image [W | V | W])
// The band is left in the lower band of A, R (with the signs of the reconstruction) in the panel.
#include "ek_common.h"
#include "ek_block64.h"

#include <chrono>
#include <cstdlib>
#include <vector>

namespace ek {
namespace {

using namespace b64;          // 64x64 LDS images and the workgroup-wide factorisations on them
static_assert(SB == kBandW, "the panel width is the width of the 64x64 building blocks");
// rows per workgroup of the tall-skinny kernels (slabs of 64).  One slab: the chain of a panel is ~14 dependent
// launches, and 64-row chunks put twice as many workgroups on each of them as 128-row chunks did
// (stage 1 at N = 4096: 19.7 -> 17.7 ms, N = 16384: 215.7 -> 210.0 ms, N = 32768: 1.387 -> 1.382 s)
constexpr int CH = 64;
#ifndef EK_CHOLQR_MAXDEV
#define EK_CHOLQR_MAXDEV 0.25      // largest |Q1^T Q1 - I| after the first CholeskyQR pass that the second pass is trusted with
#endif

__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): this wave's LDS operations have completed
  __builtin_amdgcn_wave_barrier();
}

// ---------------------------------------------------------------- tall-skinny passes over a panel
struct PanelArgs {
  int m;                    // rows of the panel
  const double *src; int lds_;   // source (m x 64, column-major)
  const double *M;          // 64x64 multiplier (column-major, ld 64) when MUL
  double *dst; int ldd;     // destination of src * M when MUL (may be null)
  double *Gpart;            // one 64x64 partial Gram (of the OUTPUT rows) per workgroup when GRAM
  // FINAL pass (the reflectors of the panel): extra outputs
  const double *L1;         // top 64x64 block of V (column-major, ld 64)
  const double *Rband;      // S R, upper triangular (column-major, ld 64)
  double *Vall; int ldv;    // explicit reflector matrix of the back-transformation, at (r0, c0)
  double *Apanel; int lda;  // the panel inside A, at (r0, c0)
  double *Vd1, *Vd2; int ldi;   // where V goes in the update's operand images (column 0 of the panel's block; Vd2 may be null)
  const int *pflag;         // FINAL pass: non-zero = this panel was factored by the rescue (house_tall_kernel): its outputs
  const double *tau;        //   are V (from the panel's storage), R alone left in the panel, and T from tau and V^T V (the last
  double *Tout;             //   workgroup to finish sums the chunks' partial products in a fixed order)
  unsigned *arrived;        //   (a counter that is zero between launches)
  int *nz;                  // MODE 0: *nz = 1 + the last 64-row chunk of the panel that is not all zeros (atomic max)
};

// out slab (64 x 64, LDS, row-major) = in slab * M^T-image; see mm64 for the operand convention
__device__ __forceinline__ void slab_mul(const double *sS, const double *sMT, double *sO) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l15 = lane & 15, l4 = lane >> 4;
  const int i0 = 16 * wave;
  double4_t acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] = (double4_t){0.0, 0.0, 0.0, 0.0};
  for (int kk = 0; kk < SB; kk += 4) {
    const double x = sS[(i0 + l15) * LD + kk + l4];
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
      acc[jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, sMT[(16 * jt + l15) * LD + kk + l4], acc[jt], 0, 0, 0);
  }
#pragma unroll
  for (int jt = 0; jt < 4; ++jt)
#pragma unroll
    for (int r = 0; r < 4; ++r) sO[(i0 + l4 + 4 * r) * LD + 16 * jt + l15] = acc[jt][r];
}

// out slab X (64 x 64) with X R = in slab, R upper triangular (sR[i * LD + j] = R(i, j), srd[j] = 1 / R(j, j)): row by
// row a forward substitution -- backward stable whatever the condition of R, where the product with an explicitly formed
// R^-1 leaves eps cond(R) (a panel of condition 1e7 lost seven digits that way).  Wave w owns rows 16 w .. 16 w + 15: the
// 16 x 16 diagonal blocks by substitution (lane = row; R's entries are broadcast reads), the blocks to their right on
// the matrix cores; nothing crosses a wave, so no workgroup barrier inside.  The in slab is overwritten.
__device__ __forceinline__ void slab_solve_upper(double *sS, const double *sR, const double *srd, double *sO) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, l4 = lane >> 4;
  const int i0 = 16 * wave;
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    if (lane < 16) {
      double a[16];
      const double *row = sS + (i0 + lane) * LD + 16 * b;
#pragma unroll
      for (int c = 0; c < 16; ++c) a[c] = row[c];
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const double x = a[j] * srd[16 * b + j];
        a[j] = x;
        const double *rr = sR + (16 * b + j) * LD + 16 * b;
#pragma unroll
        for (int c = j + 1; c < 16; ++c) a[c] = __builtin_fma(-x, rr[c], a[c]);
      }
      double *out = sO + (i0 + lane) * LD + 16 * b;
#pragma unroll
      for (int c = 0; c < 16; ++c) out[c] = a[c];
    }
    wave_sync();
#pragma unroll
    for (int bp = b + 1; bp < 4; ++bp) {
      double4_t acc;
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] = sS[(i0 + l4 + 4 * r) * LD + 16 * bp + l15];
#pragma unroll
      for (int kk = 0; kk < 16; kk += 4)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-sO[(i0 + l15) * LD + 16 * b + kk + l4], sR[(16 * b + kk + l4) * LD + 16 * bp + l15], acc, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) sS[(i0 + l4 + 4 * r) * LD + 16 * bp + l15] = acc[r];
    }
    wave_sync();
  }
}

// G(i, j) += sum_r X(r, i) Y(r, j) over the 64 rows of two LDS slabs; wave w owns rows 16 w .. of G
__device__ __forceinline__ void slab_gram(const double *sX, const double *sY, double4_t (&acc)[4]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, l4 = lane >> 4;
  for (int kk = 0; kk < SB; kk += 4) {
    const double x = sX[(kk + l4) * LD + 16 * wave + l15];
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
      acc[jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, sY[(kk + l4) * LD + 16 * jt + l15], acc[jt], 0, 0, 0);
  }
}

__device__ __forceinline__ void store_gram(const double4_t (&acc)[4], double *G) {
  // G(i, j) with i = 16 wave + l4 + 4 r, j = 16 jt + l15, stored at j + 64 i (lanes contiguous)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, l4 = lane >> 4;
#pragma unroll
  for (int jt = 0; jt < 4; ++jt)
#pragma unroll
    for (int r = 0; r < 4; ++r) G[(16 * jt + l15) + SB * (16 * wave + l4 + 4 * r)] = acc[jt][r];
}

// MODE 0: Gram of src.  MODE 1: dst = src M^-1 for the upper triangular M = R1 (by substitution), Gram of dst.
// MODE 2: final pass (V = src M below the top block, L1 in it) with the writes of the reflectors.
template <int MODE>
__global__ __launch_bounds__(256) void panel_kernel(PanelArgs p) {
  __shared__ double sS[IMG], sO[IMG], sMT[MODE ? IMG : 1], s_rd[MODE == 1 ? SB : 1];
  const int t = threadIdx.x, r = t & 63, cg = t >> 6;
  if (MODE == 2 && p.pflag && *p.pflag) {
    // ---- the outputs of a rescued panel (DGEQR2's storage in the panel, tau)
    double4_t ga[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) ga[j] = (double4_t){0.0, 0.0, 0.0, 0.0};
    for (int slab = 0; slab < CH / SB; ++slab) {
      const int row0 = blockIdx.x * CH + slab * SB;
      if (row0 >= p.m) break;
      const int row = row0 + r;
      __syncthreads();
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const int col = 16 * cg + c;
        double v = 0.0;
        if (row < p.m) {
          const double x = p.Apanel[(size_t)row + (size_t)col * p.lda];
          v = (row > col) ? x : (row == col ? 1.0 : 0.0);
          p.Apanel[(size_t)row + (size_t)col * p.lda] = (row <= col) ? x : 0.0;
          p.Vall[(size_t)row + (size_t)col * p.ldv] = v;
          p.Vd1[(size_t)row + (size_t)col * p.ldi] = v;
          if (p.Vd2) p.Vd2[(size_t)row + (size_t)col * p.ldi] = v;
        }
        sS[r * LD + col] = v;
      }
      __syncthreads();
      slab_gram(sS, sS, ga);
    }
    store_gram(ga, p.Gpart + (size_t)blockIdx.x * SB * SB);
    // the last workgroup to arrive forms T = DLARFT(V, tau) from the sum of the partial products
    __shared__ unsigned s_last;
    __threadfence();
    __syncthreads();
    if (t == 0) s_last = (atomicAdd(p.arrived, 1u) + 1u == gridDim.x) ? 1u : 0u;
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    if (t == 0) *p.arrived = 0u;
    double *sG = sO, *sT = sMT;
    __shared__ double s_tau[SB];
    const int npart = (int)gridDim.x;
    for (int idx = t; idx < SB * SB; idx += 256) {
      double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
      int q = 0;
      for (; q + 4 <= npart; q += 4) {
        a0 += p.Gpart[(size_t)q * 4096 + idx]; a1 += p.Gpart[(size_t)(q + 1) * 4096 + idx];
        a2 += p.Gpart[(size_t)(q + 2) * 4096 + idx]; a3 += p.Gpart[(size_t)(q + 3) * 4096 + idx];
      }
      for (; q < npart; ++q) a0 += p.Gpart[(size_t)q * 4096 + idx];
      sG[(idx >> 6) * LD + (idx & 63)] = (a0 + a1) + (a2 + a3); sT[(idx >> 6) * LD + (idx & 63)] = 0.0;
    }
    if (t < SB) s_tau[t] = p.tau[t];
    __syncthreads();
    for (int i = 0; i < SB; ++i) {
      const double ti = s_tau[i];
      if (t < i) {
        double a = 0.0;
        for (int l = t; l < i; ++l) a += sT[t * LD + l] * sG[l * LD + i];
        sT[t * LD + i] = -ti * a;
      } else if (t == i) sT[i * LD + i] = ti;
      __syncthreads();
    }
    for (int idx = t; idx < SB * SB; idx += 256) {
      const int i = idx & 63, j = idx >> 6;
      p.Tout[idx] = sT[i * LD + j];
    }
    return;
  }
  if (MODE == 1) {                                   // sMT(i, j) = R1(i, j) as stored (column-major in memory), 1 / diagonal
    for (int idx = t; idx < SB * SB; idx += 256) sMT[(idx & 63) * LD + (idx >> 6)] = p.M[idx];
    if (t < SB) s_rd[t] = 1.0 / p.M[t + SB * t];
  }
  if (MODE == 2) {
    for (int idx = t; idx < SB * SB; idx += 256) {
      const int k = idx & 63, j = idx >> 6;
      sMT[j * LD + k] = p.M[k + SB * j];
    }
  }
  double4_t acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] = (double4_t){0.0, 0.0, 0.0, 0.0};
  for (int slab = 0; slab < CH / SB; ++slab) {
    const int row0 = blockIdx.x * CH + slab * SB;
    if (row0 >= p.m) break;
    const int row = row0 + r;
    __syncthreads();
    bool any = false;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const int col = 16 * cg + c;
      const double v = (row < p.m) ? p.src[(size_t)row + (size_t)col * p.lds_] : 0.0;
      sS[r * LD + col] = v;
      any = any || v != 0.0;
    }
    if (MODE == 0 && p.nz && __any(any) && (t & 63) == 0) atomicMax(p.nz, (int)blockIdx.x * (CH / SB) + slab + 1);
    __syncthreads();
    const double *sOut = sS;
    if (MODE == 1) {
      slab_solve_upper(sS, sMT, s_rd, sO);
      __syncthreads();
      sOut = sO;
    }
    if (MODE == 2) {
      slab_mul(sS, sMT, sO);
      __syncthreads();
      sOut = sO;
    }
    if (MODE == 1 && p.dst && row < p.m) {
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const int col = 16 * cg + c;
        p.dst[(size_t)row + (size_t)col * p.ldd] = sO[r * LD + col];
      }
    }
    if (MODE == 2 && row < p.m) {
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const int col = 16 * cg + c;
        double v = sO[r * LD + col], a = 0.0;
        if (row0 == 0) {          // top block: V = L1 (unit lower triangular), panel = S R (upper)
          v = (r > col) ? p.L1[r + SB * col] : (r == col ? 1.0 : 0.0);
          a = (r <= col) ? p.Rband[r + SB * col] : 0.0;
        }
        p.Vall[(size_t)row + (size_t)col * p.ldv] = v;
        p.Vd1[(size_t)row + (size_t)col * p.ldi] = v;
        if (p.Vd2) p.Vd2[(size_t)row + (size_t)col * p.ldi] = v;
        p.Apanel[(size_t)row + (size_t)col * p.lda] = a;
      }
    }
    if (MODE != 2) slab_gram(sOut, sOut, acc);
  }
  if (MODE != 2) store_gram(acc, p.Gpart + (size_t)blockIdx.x * SB * SB);
}

// sum of `npart` 64x64 partials in a fixed order: grid 128 x 256 threads.  A workgroup owns 32 entries;
// thread (q = t / 32, e) adds the partials q, q + 8, q + 16, .. (two accumulators), the eight sums of an
// entry meet in LDS.  (One thread per entry over all partials was a chain of npart dependent-latency loads:
// with 64-row chunks a panel of 16 000 rows has 250 partials.)
__global__ __launch_bounds__(256) void reduce_parts_kernel(int npart, const double *__restrict__ part,
                                                           double *__restrict__ out, const int *only_if = nullptr) {
  __shared__ double s_sum[8][32];
  if (only_if && !*only_if) return;
  const int t = threadIdx.x, q = t >> 5, el = t & 31, e = blockIdx.x * 32 + el;
  double a0 = 0.0, a1 = 0.0;
  int p = q;
  for (; p + 8 < npart; p += 16) { a0 += part[(size_t)p * 4096 + e]; a1 += part[(size_t)(p + 8) * 4096 + e]; }
  if (p < npart) a0 += part[(size_t)p * 4096 + e];
  s_sum[q][el] = a0 + a1;
  __syncthreads();
  if (t < 32)
    out[e] = ((s_sum[0][t] + s_sum[1][t]) + (s_sum[2][t] + s_sum[3][t])) + ((s_sum[4][t] + s_sum[5][t]) + (s_sum[6][t] + s_sum[7][t]));
}

// first CholeskyQR pass: G (64x64, symmetric, stored j + 64 i) -> R1 (column-major) and R1^-1
// (*pflag: the flag of THIS panel, written here -- 0 or 1 -- and possibly raised by hr_kernel: a panel CholeskyQR2
// cannot factor goes to the Householder rescue below)
__global__ __launch_bounds__(256) void chol_kernel(const double *__restrict__ G, double *__restrict__ R,
                                                   double *__restrict__ Rinv, int *pflag, const int *nz) {
  __shared__ double sA[IMG];                               // (one image: 42 KB fit beside both workgroups of an update)
  __shared__ double s_inv[kScratch];
  const int t = threadIdx.x;
  double gv[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) gv[k] = G[t + 256 * k];
#pragma unroll
  for (int k = 0; k < 16; ++k) { const int idx = t + 256 * k; sA[(idx >> 6) * LD + (idx & 63)] = gv[k]; }
  __syncthreads();
  // (an all-zero panel -- an input that is already banded: flag 4, the rescue's short cut.  The test is the first
  // pass's scan of the entries themselves (*nz = 1 + the last chunk with a non-zero entry), not the Gram diagonal: the
  // squares of a column of entries below 1e-162 underflow to a zero diagonal, and such a panel is not zero)
  __shared__ int s_nz;
  if (t == 0) s_nz = nz ? (*nz != 0) : 0;
  __syncthreads();
  if (!nz && t < SB && sA[t * LD + t] != 0.0) atomicOr(&s_nz, 1);
  __syncthreads();
  const int zero_panel = !s_nz;
  const int bad = chol64_upper_wg(sA, s_inv);
  if (t == 0) *pflag = zero_panel ? 4 : ((bad >= 0) ? 1 : 0);
  (void)Rinv;                                              // (no explicit inverse: the panel is divided by R1 by substitution)
  __syncthreads();
  for (int idx = t; idx < SB * SB; idx += 256) {
    const int i = idx & 63, j = idx >> 6;
    R[idx] = sA[i * LD + j];
  }
}

// second pass + Householder reconstruction.  In: G2 = Qt^T Qt, the top 64 rows of Qt, R1.
// Out: M2 = R2^-1 U^-1, T, L1, Rband = S R2 R1, tau.
struct HrArgs {
  const double *G2; const double *Qt; int ldq; const double *R1;
  double *M2, *T, *L1, *Rband, *tau;
  int *flag;
  long long *prof;   // optional: shader cycles per phase
};
// LDS: THREE 64 x 64 images (round 5; four until round 4).  With 135 KB the workgroup could only start on a CU that BOTH
// resident workgroups of a trailing update had left, and a freed slot went to the update's next workgroup first: beside
// an 8-wave update the launch lasted until the update's grid had drained (trace: 475 us against 44 alone), so the chain
// of the look-ahead ended AFTER the update it was to hide behind.  110 KB fit a CU as soon as one of the two has left.
// And 256 registers instead of 357 (the four waves of an update's workgroup leave 256 per SIMD lane).  What made room: Q top
// passes through the accumulators, R1 arrives late, and R2 R1 is formed early and waits in its place in memory for the
// signs.  The same operations on the same operands: same bits.
__global__ __launch_bounds__(256, 2) void hr_kernel(HrArgs p) {
  extern __shared__ double smem[];
  double *sA = smem, *sB = smem + IMG, *sC = smem + 2 * IMG;
  __shared__ double s_sign[SB];
  __shared__ double s_red[4];
  __shared__ double s_inv[kScratch];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  long long tc[10]; int nt = 0;
  const bool prof = p.prof && t == 0;
  if (prof) tc[nt++] = clock64();
  // G2 and its distance from the identity: the loss of orthogonality of the first pass
  double dev = 0.0;
  double gv[16], qv[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int idx = t + 256 * k;
    gv[k] = p.G2[idx];
    qv[k] = p.Qt[(size_t)(idx & 63) + (size_t)(idx >> 6) * p.ldq];
  }
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int idx = t + 256 * k, lo = idx & 63, hi = idx >> 6;
    sA[hi * LD + lo] = gv[k];                                             // G2(i = hi, j = lo)
    const double e = fabs(gv[k] - (lo == hi ? 1.0 : 0.0));
    dev = (e > dev || e != e) ? e : dev;
    sC[lo * LD + hi] = qv[k];                                             // top block of Qt, (i = lo, j = hi)
  }
  for (int o = 32; o > 0; o >>= 1) { const double y = __shfl_down(dev, o, 64); dev = (y > dev || y != y) ? y : dev; }
  if (lane == 0) s_red[wave] = dev;
  __syncthreads();
  double dmax = 0.0;
  for (int w = 0; w < 4; ++w) dmax = (s_red[w] > dmax || s_red[w] != s_red[w]) ? s_red[w] : dmax;
  if (t == 0) {
    // max |Q1^T Q1 - I| after the first pass is eps kappa^2 (up to a modest factor): up to 0.25 (kappa ~ 1e7) the
    // second pass orthogonalises to rounding.  (Q1 = A R1^-1 is formed by SUBSTITUTION: with the explicit inverse of R1,
    // as until the end of round 3, its error was eps kappa(R1) relative to A -- a band of half width 65, whose panels are
    // random triangles of kappa ~ 1e7, came out with eigenvalues 1.5e-9 off.)  The rest goes to the rescue.
    if (!(dmax <= EK_CHOLQR_MAXDEV)) atomicOr(p.flag, 2);
  }
  if (prof) tc[nt++] = clock64();
  if (dmax <= 1e-10) {                                                   // (uniform)
    // G2 = I + E with |E| <= 1e-10: the factor is R2 = I + U, U = striu(E) + diag(E) / 2, and R2^-1 = I - U, both to
    // O(E^2) = 64 x 1e-20 entrywise, far below the rounding of the products they enter -- the 64 serial steps of a
    // Cholesky factorisation and of a triangular inverse (40 000 of this kernel's 135 000 cycles) are a formula
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int idx = t + 256 * k, j = idx & 63, i = idx >> 6;         // gv[k] = G2(i, j)
      const double u = (i < j) ? gv[k] : (i == j ? 0.5 * (gv[k] - 1.0) : 0.0), one = (i == j) ? 1.0 : 0.0;
      sA[i * LD + j] = one + u;                                          // sA = R2
      sB[i * LD + j] = one - u;                                          // sB = R2^-1
    }
    __syncthreads();
    if (prof) { tc[nt++] = clock64(); tc[nt++] = clock64(); }
  } else {
    if (chol64_upper_wg(sA, s_inv) >= 0 && t == 0) atomicOr(p.flag, 1);      // sA = R2
    if (prof) tc[nt++] = clock64();
    triinv64_upper_wg(sA, sB, s_inv);                                    // sB = R2^-1
    if (prof) tc[nt++] = clock64();
  }
  // Q top = Qt_top R2^-1 into the accumulators; then R1 takes the place of Qt's top block and R2 R1 is formed while R2
  // is still there (the signs it is scaled by come from the LU below); then Q top takes R2's place
  double r1v[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) r1v[k] = p.R1[t + 256 * k];               // (in flight during the first product)
  double4_t qacc[4], rr[4];
  mm64_acc(sC, false, sB, false, qacc);
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 16; ++k) {                                        // sC = R1
    const int idx = t + 256 * k;
    sC[(idx & 63) * LD + (idx >> 6)] = r1v[k];
  }
  __syncthreads();
  mm64_acc(sA, false, sC, false, rr);                                   // R2 R1
  {                                                                     // (parked where it belongs; this thread scales its
    const int l15 = lane & 15, l4 = lane >> 4, i0 = 16 * wave;          // own entries once the signs are known)
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
      for (int r = 0; r < 4; ++r) p.Rband[(i0 + l4 + 4 * r) + SB * (16 * jt + l15)] = rr[jt][r];
  }
  __syncthreads();
  double *sD = sA;                                                       // (R2 has served)
  mm64_store(qacc, sD);                                                 // sD = Q top
  __syncthreads();
  if (prof) tc[nt++] = clock64();
  lu64_signed_wg(sD, s_sign, s_inv);                               // sD = L1 \ U of (Q_top - S)
  if (prof) tc[nt++] = clock64();
  for (int idx = t; idx < SB * SB; idx += 256) {
    const int i = idx & 63, j = idx >> 6;
    p.L1[idx] = (i > j) ? sD[i * LD + j] : (i == j ? 1.0 : 0.0);
  }
  // T = -U S L1^-T (into sC transposed: sC(j, i) = T(i, j)); the top block of Qt is no longer needed
  tsolve64_wg(sD, s_sign, sC, s_inv);
  for (int idx = t; idx < SB * SB; idx += 256) {
    const int i = idx & 63, j = idx >> 6;
    p.T[idx] = sC[j * LD + i];
    if (i == j) p.tau[i] = sC[i * LD + i];
  }
  __syncthreads();
  triinv64_upper_wg(sD, sC, s_inv);                                      // sC = U^-1
  if (prof) { tc[nt++] = clock64(); tc[nt++] = clock64(); }
  __syncthreads();
  mm64(sB, false, sC, false, nullptr, p.M2);                            // M2 = R2^-1 U^-1
  {                                                                     // S R2 R1
    const int l15 = lane & 15, l4 = lane >> 4, i0 = 16 * wave;
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = i0 + l4 + 4 * r, j = 16 * jt + l15;
        p.Rband[i + SB * j] = s_sign[i] * p.Rband[i + SB * j];
      }
  }
  if (prof) { tc[nt++] = clock64(); for (int q = 0; q + 1 < nt; ++q) p.prof[q] += tc[q + 1] - tc[q]; p.prof[9] += 1; }
}

// ---------------------------------------------------------------- panel with few rows: Householder
// QR inside one workgroup (any rank, any m <= 256).  Same outputs as the CholeskyQR2 chain.
struct SmallArgs {
  int m;
  double *Apanel; int lda; double *Vall; int ldv; double *Vd1, *Vd2; int ldi;
  double *T, *tau;
};
constexpr int SMALL_MAX = 127;              // panels of 128 rows and more go through the CholeskyQR2 chain
constexpr int SMALL_ROWS = 128;             // rows of the LDS image (two slabs of 64)
__global__ __launch_bounds__(256) void house_small_kernel(SmallArgs p) {
  extern __shared__ double smem[];
  double *sP = smem;                       // 128 x 64 panel, row-major (LD); rows >= m are zero
  double *sT = smem + SMALL_ROWS * LD;     // T, row-major
  double *sG = sT + IMG;                   // V^T V, row-major
  __shared__ double s_tau[SB], s_w[4][SB], s_red[2];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int m = p.m;
  for (int idx = t; idx < SMALL_ROWS * SB; idx += 256) {
    const int r = idx % SMALL_ROWS, c = idx / SMALL_ROWS;
    sP[r * LD + c] = (r < m) ? p.Apanel[(size_t)r + (size_t)c * p.lda] : 0.0;
  }
  for (int idx = t; idx < IMG; idx += 256) sT[idx] = 0.0;
  __syncthreads();
  for (int j = 0; j < SB; ++j) {
    double tau = 0.0;
    if (j < m - 1) {                       // uniform
      // DLARFG on x = P(j:m, j): thread t < 128 holds row t
      double x = (t > j && t < m) ? sP[t * LD + j] : 0.0;
      double ssq = x * x;
      if (wave < 2) {
        for (int o = 32; o > 0; o >>= 1) ssq += __shfl_down(ssq, o, 64);
        if (lane == 0) s_red[wave] = ssq;
      }
      __syncthreads();
      ssq = s_red[0] + s_red[1];
      const double alpha = sP[j * LD + j];
      if (ssq != 0.0 && alpha * alpha + ssq > 1e-290) {   // uniform (a column of denormal squares is dropped: reflector_of, ek_sb2st.hip)
        const double beta = -copysign(hypot(alpha, sqrt(ssq)), alpha);
        tau = (beta - alpha) / beta;
        const double scale = 1.0 / (alpha - beta);
        __syncthreads();                   // everybody has read alpha
        if (t > j && t < m) sP[t * LD + j] = x * scale;
        if (t == j) sP[j * LD + j] = beta;
        __syncthreads();
        // H_j on the columns c > j: wave q takes the rows r = j + q, j + q + 4, ..; v(j) = 1
        const int c = lane;
        double w = 0.0;
        for (int r = j + wave; r < m; r += 4) {
          const double v = (r == j) ? 1.0 : sP[r * LD + j];
          w += v * sP[r * LD + c];
        }
        s_w[wave][c] = w;
        __syncthreads();
        w = tau * ((s_w[0][c] + s_w[1][c]) + (s_w[2][c] + s_w[3][c]));
        if (c > j)
          for (int r = j + wave; r < m; r += 4) {
            const double v = (r == j) ? 1.0 : sP[r * LD + j];
            sP[r * LD + c] -= w * v;
          }
      }
    }
    if (t == 0) s_tau[j] = tau;
    __syncthreads();
  }
  // the triangle R goes out, V takes its place: unit diagonal, zeros above (no reflector for c >= m - 1)
  for (int idx = t; idx < m * SB; idx += 256) {
    const int r = idx % m, c = idx / m;
    const double x = sP[r * LD + c];
    p.Apanel[(size_t)r + (size_t)c * p.lda] = (r <= c || c >= m - 1) ? x : 0.0;
  }
  __syncthreads();
  for (int idx = t; idx < SMALL_ROWS * SB; idx += 256) {
    const int r = idx % SMALL_ROWS, c = idx / SMALL_ROWS;
    if (c >= m - 1 || r < c) sP[r * LD + c] = 0.0;
    else if (r == c) sP[r * LD + c] = 1.0;
  }
  __syncthreads();
  {
    double4_t acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = (double4_t){0.0, 0.0, 0.0, 0.0};
    slab_gram(sP, sP, acc);
    slab_gram(sP + SB * LD, sP + SB * LD, acc);
    const int l15 = lane & 15, l4 = lane >> 4;
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
      for (int q = 0; q < 4; ++q) sG[(16 * wave + l4 + 4 * q) * LD + 16 * jt + l15] = acc[jt][q];
  }
  __syncthreads();
  // T (DLARFT, forward columnwise): T(i,i) = tau_i, T(0:i, i) = -tau_i T(0:i,0:i) G(0:i, i)
  for (int i = 0; i < SB; ++i) {
    const double ti = s_tau[i];
    if (t < i) {
      double a = 0.0;
      for (int l = t; l < i; ++l) a += sT[t * LD + l] * sG[l * LD + i];
      sT[t * LD + i] = -ti * a;
    } else if (t == i) sT[i * LD + i] = ti;
    __syncthreads();
  }
  for (int idx = t; idx < SB * SB; idx += 256) {
    const int i = idx & 63, j = idx >> 6;
    p.T[idx] = sT[i * LD + j];
    if (i == j) p.tau[i] = s_tau[i];
  }
  for (int idx = t; idx < m * SB; idx += 256) {
    const int r = idx % m, c = idx / m;
    const double v = sP[r * LD + c];
    p.Vall[(size_t)r + (size_t)c * p.ldv] = v;
    p.Vd1[(size_t)r + (size_t)c * p.ldi] = v;
    if (p.Vd2) p.Vd2[(size_t)r + (size_t)c * p.ldi] = v;
  }
}

// ---------------------------------------------------------------- rescue: Householder QR of a tall panel
// A panel CholeskyQR2 cannot factor (rank deficient or cond > 1e7: a zero or repeated column, an input that is
// already banded, low rank plus identity, ...) is factored by Householder reflections instead -- any rank, any
// conditioning, like PDSYTRD's own panels (solver_scalapack_all.f90:59) -- by ONE workgroup that walks the panel in
// blocks of 8 columns: the block is factored column by column (norm, reflector, application to the rest of the
// block: three passes over its rows), then the block's compact-WY factor is applied to the columns to its right, 8 at
// a time (two passes).  About 15 passes of the panel's bytes through one compute unit: 1 - 3 ms for the largest
// panels, paid only by the panels that need it (the kernels below leave at once when the panel's flag is clear).
// In place: R in the upper triangle, the reflectors below the diagonal (DGEQR2's storage), tau.
constexpr int HT = 256;                 // threads of the rescue workgroup (see house_tall_kernel)
constexpr int HB = 8;                   // columns per block
template <int K>
__device__ __forceinline__ void ht_reduce(double (&a)[K], double *s_part /* [HT / 64][K] */, double *s_out /* [K] */) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    double v = a[k];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if (lane == 0) s_part[wave * K + k] = v;
  }
  __syncthreads();
  if (t < K) {
    double v = 0.0;
    for (int w = 0; w < HT / 64; ++w) v += s_part[w * K + t];
    s_out[t] = v;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < K; ++k) a[k] = s_out[k];
  __syncthreads();
}

// nz: 1 + the last 64-row chunk of the panel with a non-zero entry (panel_kernel<0>): the rows below are exactly zero and
// stay zero under every reflector, so the walk stops there -- the panels of a matrix that is nearly a band (the
// reference's sparse Hamiltonians) cost a few hundred rows each instead of the full height.
// (Round 5: ONE predicated launch in the chain instead of three.  Until round 4 two more -- tall_finish_kernel over the
// chunks and t_from_gram_kernel -- cost EVERY panel of every matrix two empty launches; the rescued panel's outputs are now
// written by the chain's last pass, panel_kernel<2>, which runs anyway: chunk by chunk on all its workgroups, the last one
// to finish forms T.  And the workgroup has four waves of at most 256 registers, not eight: eight needed a CU that BOTH
// workgroups of a trailing update had left, so beside an update the EMPTY launch lasted until the update's grid had
// drained (trace: 219 us); four fit as soon as one has left (15 us), and the factorisation itself is no slower -- it is
// bound by one CU's memory pipe (band of half width 65 at N = 16384, 70 panels rescued: 1.13 x the dense solve, as before).)
__global__ __launch_bounds__(HT, 2) void house_tall_kernel(int m_full, double *__restrict__ P, int ldp, double *__restrict__ tau_out,
                                                           const int *pflag, int *d_flag, const int *nz) {
  __shared__ double s_part[(HT / 64) * HB * HB], s_out[HB * HB];
  __shared__ double s_T[HB * HB], s_X[HB * HB], s_tau[SB];
  if (!*pflag) return;
  const int t = threadIdx.x;
  int m = m_full;
  if (nz) { const int me = *nz * SB; if (me < m) m = (me > SB + 1) ? me : ((SB + 1 < m) ? SB + 1 : m); }
  if (t == 0) atomicAdd(d_flag, 256);                      // bits 8..: panels that took this path (informational)
  if (*pflag & 4) {                                        // the panel is exactly zero: R = 0, H = I
    if (t < SB) tau_out[t] = 0.0;
    return;
  }
  auto vget = [&](int r, int j) -> double {                // entry (r, j) of the unit lower trapezoidal V
    return (r > j) ? P[(size_t)r + (size_t)j * ldp] : (r == j ? 1.0 : 0.0);
  };
  for (int jb = 0; jb < SB; jb += HB) {
    // ---- the block, column by column
    bool block_active = false;
    for (int j = jb; j < jb + HB; ++j) {
      double a1[1] = {0.0};
      for (int r = j + 1 + t; r < m; r += HT) { const double x = P[(size_t)r + (size_t)j * ldp]; a1[0] += x * x; }
      ht_reduce<1>(a1, s_part, s_out);
      const double ssq = a1[0], alpha = P[(size_t)j + (size_t)j * ldp];
      double tau = 0.0, beta = alpha, scale = 0.0;
      if (ssq != 0.0 && alpha * alpha + ssq > 1e-290) {     // (a column of denormal squares is dropped: see reflector_of, ek_sb2st.hip)
        beta = -copysign(sqrt(alpha * alpha + ssq), alpha);
        tau = (beta - alpha) / beta;
        scale = 1.0 / (alpha - beta);
      }
      __syncthreads();                                     // everybody has read alpha
      if (t == 0) { s_tau[j] = tau; tau_out[j] = tau; P[(size_t)j + (size_t)j * ldp] = beta; }
      if (tau == 0.0) {                                    // nothing to annihilate in this column: H = I (uniform)
        if (ssq != 0.0)                                    // (a dropped column of denormal squares: its entries are zeros now)
          for (int r = j + 1 + t; r < m; r += HT) P[(size_t)r + (size_t)j * ldp] = 0.0;
        continue;
      }
      block_active = true;
      // v = x * scale (stored), w_c = v^T p_c for the columns of the block to the right of j
      double w[HB - 1];
#pragma unroll
      for (int c = 0; c < HB - 1; ++c) w[c] = 0.0;
      const int nc = jb + HB - 1 - j;                      // columns j+1 .. jb+HB-1
      for (int r = j + t; r < m; r += HT) {
        double v = 1.0;
        if (r > j) { v = P[(size_t)r + (size_t)j * ldp] * scale; P[(size_t)r + (size_t)j * ldp] = v; }
#pragma unroll
        for (int c = 0; c < HB - 1; ++c)
          if (c < nc) w[c] += v * P[(size_t)r + (size_t)(j + 1 + c) * ldp];
      }
      ht_reduce<HB - 1>(w, s_part, s_out);
      for (int r = j + t; r < m; r += HT) {
        const double v = (r > j) ? P[(size_t)r + (size_t)j * ldp] : 1.0;
#pragma unroll
        for (int c = 0; c < HB - 1; ++c)
          if (c < nc) P[(size_t)r + (size_t)(j + 1 + c) * ldp] -= tau * w[c] * v;
      }
      __syncthreads();
    }
    if (!block_active) continue;                           // a block without a reflector (a panel that is already triangular)
    if (jb + HB >= SB) break;
    // ---- T of the block (DLARFT, forward columnwise) from G = V^T V
    {
      double g[HB * HB];
#pragma unroll
      for (int q = 0; q < HB * HB; ++q) g[q] = 0.0;
      for (int r = jb + t; r < m; r += HT) {
        double v[HB];
#pragma unroll
        for (int k = 0; k < HB; ++k) v[k] = vget(r, jb + k);
#pragma unroll
        for (int a = 0; a < HB; ++a)
#pragma unroll
          for (int b = a + 1; b < HB; ++b) g[a * HB + b] += v[a] * v[b];
      }
      ht_reduce<HB * HB>(g, s_part, s_out);
      if (t == 0) {
        for (int q = 0; q < HB * HB; ++q) s_T[q] = 0.0;
        for (int i = 0; i < HB; ++i) {
          const double ti = s_tau[jb + i];
          for (int a = 0; a < i; ++a) {
            double acc = 0.0;
            for (int l = a; l < i; ++l) acc += s_T[a * HB + l] * g[l * HB + i];
            s_T[a * HB + i] = -ti * acc;
          }
          s_T[i * HB + i] = ti;
        }
      }
      __syncthreads();
    }
    // ---- (I - V T V^T)^T C = C - V T^T (V^T C) on the columns to the right, 8 at a time
    for (int c0 = jb + HB; c0 < SB; c0 += HB) {
      double wv[HB * HB];                                  // W(k, c) = sum_r V(r, k) C(r, c)
#pragma unroll
      for (int q = 0; q < HB * HB; ++q) wv[q] = 0.0;
      for (int r = jb + t; r < m; r += HT) {
        double v[HB], cc[HB];
#pragma unroll
        for (int k = 0; k < HB; ++k) { v[k] = vget(r, jb + k); cc[k] = P[(size_t)r + (size_t)(c0 + k) * ldp]; }
#pragma unroll
        for (int k = 0; k < HB; ++k)
#pragma unroll
          for (int c = 0; c < HB; ++c) wv[k * HB + c] += v[k] * cc[c];
      }
      ht_reduce<HB * HB>(wv, s_part, s_out);
      if (t < HB * HB) {                                   // X = T^T W
        const int k = t / HB, c = t % HB;
        double acc = 0.0;
        for (int l = 0; l <= k; ++l) acc += s_T[l * HB + k] * wv[l * HB + c];
        s_X[k * HB + c] = acc;
      }
      __syncthreads();
      double x[HB * HB];
#pragma unroll
      for (int q = 0; q < HB * HB; ++q) x[q] = s_X[q];
      for (int r = jb + t; r < m; r += HT) {
        double v[HB];
#pragma unroll
        for (int k = 0; k < HB; ++k) v[k] = vget(r, jb + k);
#pragma unroll
        for (int c = 0; c < HB; ++c) {
          double acc = 0.0;
#pragma unroll
          for (int k = 0; k < HB; ++k) acc += v[k] * x[k * HB + c];
          P[(size_t)r + (size_t)(c0 + c) * ldp] -= acc;
        }
      }
      __syncthreads();
    }
  }
}

// ---------------------------------------------------------------- Y = A22 V on the lower triangle
// Workgroup (rb, ks): rows 128 rb .. of Ypart[ks] = sum over the K tiles kt of its split of
// tile(rb, kt) V(kt rows), where tile(rb, kt) is read as stored for kt < rb, transposed from
// tile(kt, rb) for kt > rb, and mixed on the diagonal: every tile of the lower triangle is read
// twice per panel (once by its row's workgroups, once by its column's), never the upper triangle.
struct SymmArgs {
  int m; const double *A; int lda;      // A22 (m x m, lower)
  const double *V; int ldv;             // m x 64
  double *Ypart; int ldy; long long sY; // per split: m x 64
  int T, tiles_per_split;
  int P, rank, r0;                      // DIST: this member of a team of P sums only the entries of A22 whose COLUMN lies
                                        // in a 128-wide global strip it owns (strip S on rank S mod P; A22 starts at global
                                        // row and column r0); the members' Y add up to A22 V
  int SD = 0, ST = 0;                   // DIST: workgroups (rb, 0 .. SD-1) share the tiles up to the diagonal of block row rb,
                                        // (rb, SD .. SD+ST-1) the transposed tiles below it (see symm_lower_kernel)
};
// DIST: a member's entries are a 1 / P share of the triangle but an uneven one per block row: a block row whose rows lie in
// an owned strip takes ALL the transposed tiles below its diagonal (up to T of them), any other only every P-th direct
// slab.  Cut over K like on one GPU, the launch lasts as long as its heaviest workgroup -- as long as the undistributed
// SYMM (round 3: a team of 8 spent 0.066 s per rank in these launches at N = 16384, 1 / 8 of the flops each).  So the two
// kinds of work have their own workgroups: SD chunks of the tiles up to the diagonal, ST chunks of the transposed run of an
// owning block row (the others leave at once); yred_dist_kernel sums exactly the partials that were written.
__host__ __device__ inline int symm_dist_chunk(int count, int parts) { return (count + parts - 1) / parts; }
constexpr int BK = 32, MC_LD = 128 + 16, KC_LD = BK + 2;   // K-contiguous images: 34 keeps (x, k) and (x + 1, k - 1) in different banks and row pairs 16-byte aligned
constexpr int A_TILE = (BK * MC_LD > 128 * KC_LD) ? BK * MC_LD : 128 * KC_LD;
constexpr int V_LD = BK + 2;            // V slab: 64 x 32, K-contiguous image s[n][34]
typedef double double2_t __attribute__((ext_vector_type(2)));

// K is walked in slabs of 32 (one barrier pair per slab); slabs that lie completely inside the matrix
// and off the diagonal tile are fetched as 16-byte pairs without predicates (A22 starts on a multiple of
// 64 rows and the leading dimensions are even, so the pairs are aligned), the others element by element.
template <bool DIST>
__global__ __launch_bounds__(256, 2) void symm_lower_kernel(SymmArgs p) {
  __shared__ __attribute__((aligned(16))) double sA[A_TILE];
  __shared__ __attribute__((aligned(16))) double sV[SB * V_LD];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l15 = lane & 15, l4 = lane >> 4;
  int rb = blockIdx.x, ks = blockIdx.y;
  if (DIST) {
    // 1-D grid: T * SD workgroups for the direct chunks, then ST per block row that has owned rows (a grid over all T
    // block rows launched six thousand workgroups of which five thousand left at once)
    const int b = blockIdx.x, nd = p.T * p.SD;
    if (b < nd) { rb = b / p.SD; ks = b % p.SD; }
    else {
      const int h = (b - nd) / p.ST, j = (b - nd) % p.ST, q = p.r0 >> 7;
      ks = p.SD + j;
      if (p.P == 1) rb = h;
      else {
        const int first = q + ((p.rank - q) % p.P + p.P) % p.P;            // first owned strip at or beyond q
        if ((p.r0 & 127) == 0) rb = first + h * p.P - q;                       // block row = strip
        else rb = first + (h >> 1) * p.P - q - 1 + (h & 1);                    // a strip lies in two block rows
      }
      if (rb < 0 || rb >= p.T) return;
    }
  }
  int kt0 = ks * p.tiles_per_split;
  int kt1 = kt0 + p.tiles_per_split; if (kt1 > p.T) kt1 = p.T;
  const int m0 = rb * 128;
  if (DIST) {
    const bool heavy = ((((p.r0 + m0) >> 7) % p.P) == p.rank) || ((((p.r0 + m0 + 64) >> 7) % p.P) == p.rank);
    if (ks < p.SD) {                      // the tiles up to and including the diagonal one
      const int td = symm_dist_chunk(p.T, p.SD);
      kt0 = ks * td; kt1 = kt0 + td; if (kt1 > rb + 1) kt1 = rb + 1;
      if (kt0 > rb) return;
    } else {                              // the transposed tiles below the diagonal of an owning block row
      const int cnt = p.T - rb - 1, len = symm_dist_chunk(cnt > 0 ? cnt : 1, p.ST);
      kt0 = rb + 1 + (ks - p.SD) * len; kt1 = kt0 + len; if (kt1 > p.T) kt1 = p.T;
      if (!heavy || kt0 >= kt1) return;
    }
  }
  const int wm = (wave & 1) * 64, wn = (wave >> 1) * 32;
  double4_t acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (double4_t){0.0, 0.0, 0.0, 0.0};
  double2_t ra[8], rv[4];
  // entry a(x, k) of the symmetric matrix lives in column min(x, k) of the lower triangle: direct tiles (k < x) belong to
  // the owner of their K index, transposed tiles (x < k) to the owner of their output row
  auto own = [&](int rel) -> bool { return !DIST || (((p.r0 + rel) >> 7) % p.P) == p.rank; };
  const bool rowown[2] = {own(m0), own(m0 + 64)};
  auto needed = [&](int kt, int k0) -> bool {
    if (!DIST) return true;
    if (kt < rb) return own(kt * 128 + k0);
    if (kt > rb) return rowown[0] || rowown[1];
    return true;
  };
  // odd leading dimensions (odd n) or an unaligned base take the element path everywhere
  const bool vec_ok = (((p.lda | p.ldv) & 1) == 0) && ((((size_t)p.A | (size_t)p.V) & 15) == 0);
  const bool rows_in = vec_ok && m0 + 128 <= p.m;
  // slab (kt, k0): rows m0 .. m0+127 of the operand, K indices kt*128 + k0 .. + 31.
  //   direct tile (kt < rb): pair = rows (2 xp, 2 xp + 1) at one k;   thread -> xp = t & 63, k = t >> 6 + 4 i
  //   transposed (kt > rb):  pair = k indices (2 kp, 2 kp + 1) of one row; thread -> kp = t & 15, x = t >> 4 + 16 i
  auto load_a = [&](int kt, int k0) {
    const int mode = (kt < rb) ? 0 : (kt > rb ? 1 : 2);
    const int gk0 = kt * 128 + k0;
    if (mode == 0 && rows_in && gk0 + BK <= p.m) {
      const double *base = p.A + (size_t)(m0 + 2 * (t & 63)) + (size_t)(gk0 + (t >> 6)) * p.lda;
#pragma unroll
      for (int i = 0; i < 8; ++i) ra[i] = *reinterpret_cast<const double2_t *>(base + (size_t)(4 * i) * p.lda);
    } else if (mode == 1 && rows_in && gk0 + BK <= p.m) {
      const double *base = p.A + (size_t)(gk0 + 2 * (t & 15)) + (size_t)(m0 + (t >> 4)) * p.lda;
#pragma unroll
      for (int i = 0; i < 8; ++i) ra[i] = *reinterpret_cast<const double2_t *>(base + (size_t)(16 * i) * p.lda);
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        double v[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          int x, k;
          if (mode == 1) { k = 2 * (t & 15) + h; x = (t >> 4) + 16 * i; } else { x = 2 * (t & 63) + h; k = (t >> 6) + 4 * i; }
          const int gx = m0 + x, gk = gk0 + k;
          double e = 0.0;
          if (gx < p.m && gk < p.m && (!DIST || mode != 2 || own(gx < gk ? gx : gk))) {
            const bool low = (mode == 0) || (mode == 2 && gx >= gk);
            e = low ? p.A[(size_t)gx + (size_t)gk * p.lda] : p.A[(size_t)gk + (size_t)gx * p.lda];
          }
          v[h] = e;
        }
        ra[i] = (double2_t){v[0], v[1]};
      }
    }
    if (DIST && mode == 1) {          // rows (t >> 4) + 16 i: i < 4 the first 64 rows of the block, i >= 4 the others
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (!rowown[i >> 2]) ra[i] = (double2_t){0.0, 0.0};
    }
  };
  // V slab: pair = k indices (2 kp, 2 kp + 1) of one column n; thread -> kp = t & 15, n = t >> 4 + 16 i
  auto load_v = [&](int kt, int k0) {
    const int gk0 = kt * 128 + k0;
    if (vec_ok && gk0 + BK <= p.m) {
      const double *base = p.V + (size_t)(gk0 + 2 * (t & 15)) + (size_t)(t >> 4) * p.ldv;
#pragma unroll
      for (int i = 0; i < 4; ++i) rv[i] = *reinterpret_cast<const double2_t *>(base + (size_t)(16 * i) * p.ldv);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int gk = gk0 + 2 * (t & 15), nn = (t >> 4) + 16 * i;
        rv[i] = (double2_t){gk < p.m ? p.V[(size_t)gk + (size_t)nn * p.ldv] : 0.0,
                            gk + 1 < p.m ? p.V[(size_t)gk + 1 + (size_t)nn * p.ldv] : 0.0};
      }
    }
  };
  // the slabs of this workgroup in order (DIST: those that hold entries this member owns)
  auto advance = [&](int &kt, int &k0) -> bool {
    while (true) {
      k0 += BK;
      if (k0 >= 128 || kt * 128 + k0 >= p.m) { kt += 1; k0 = 0; }
      if (kt >= kt1 || kt * 128 >= p.m) return false;
      if (needed(kt, k0)) return true;
    }
  };
  int kt = kt0, k0 = 0;
  bool have = kt0 < kt1 && kt0 * 128 < p.m;
  if (have && !needed(kt, k0)) have = advance(kt, k0);
  if (have) { load_a(kt, k0); load_v(kt, k0); }
  while (have) {
    const bool kc = kt > rb;     // K-contiguous image of the A slab (transposed tile)
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (kc) { const int k = 2 * (t & 15), x = (t >> 4) + 16 * i; *reinterpret_cast<double2_t *>(&sA[x * KC_LD + k]) = ra[i]; }
      else    { const int x = 2 * (t & 63), k = (t >> 6) + 4 * i; *reinterpret_cast<double2_t *>(&sA[k * MC_LD + x]) = ra[i]; }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int k = 2 * (t & 15), nn = (t >> 4) + 16 * i; *reinterpret_cast<double2_t *>(&sV[nn * V_LD + k]) = rv[i]; }
    __syncthreads();
    // next slab (possibly of the next tile) in flight during the MFMAs
    int nkt = kt, nk0 = k0;
    const bool hn = advance(nkt, nk0);
    if (hn) { load_a(nkt, nk0); load_v(nkt, nk0); }
#pragma unroll
    for (int kk = 0; kk < BK; kk += 4) {
      double fa[4], fb[2];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int x = wm + i * 16 + l15, k = kk + l4;
        fa[i] = kc ? sA[x * KC_LD + k] : sA[k * MC_LD + x];
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) fb[i] = sV[(wn + i * 16 + l15) * V_LD + kk + l4];
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
          acc[ni][mi] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[ni], fa[mi], acc[ni][mi], 0, 0, 0);
    }
    kt = nkt; k0 = nk0; have = hn;
  }
  double *Y = p.Ypart + (size_t)ks * p.sY;
#pragma unroll
  for (int ni = 0; ni < 2; ++ni)
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      const int mrow = m0 + wm + mi * 16 + l15;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int nn = wn + ni * 16 + l4 + 4 * r;
        if (mrow < p.m) Y[(size_t)mrow + (size_t)nn * p.ldy] = acc[ni][mi][r];
      }
    }
}

// Y = sum of the split-K partials (kept), Gp = V_chunk^T Y_chunk per workgroup
struct YredArgs {
  int m, nsplit;
  const double *Ypart; int ldy; long long sY;
  double *Y;                 // m x 64, ld = ldyo (0: ldy)
  const double *V; int ldv;
  double *Gpart;
  int ldyo = 0;
  // second panel of a pair (yred_q_kernel): Ypart is A_stale V2 for a trailing matrix that the first panel has not
  // updated yet; Y = the sum - Wp (Vp^T V2) - Vp (Wp^T V2) with the first panel's W, V (rows of this trailing matrix, ld
  // ldc) and the two 64 x 64 products cG1 = Vp^T V2, cG2 = Wp^T V2 (stored (j + 64 i) = G(i, j)).  cW == null: none.
  const double *cW = nullptr, *cV = nullptr; int ldc = 0;
  const double *cG1 = nullptr, *cG2 = nullptr;
};
__global__ __launch_bounds__(256) void yred_kernel(YredArgs p) {
  __shared__ double sY[IMG], sV[IMG];
  const int t = threadIdx.x, r = t & 63, cg = t >> 6;
  double4_t acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] = (double4_t){0.0, 0.0, 0.0, 0.0};
  for (int slab = 0; slab < CH / SB; ++slab) {
    const int row0 = blockIdx.x * CH + slab * SB;
    if (row0 >= p.m) break;
    const int row = row0 + r;
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const int col = 16 * cg + c;
      double y = 0.0, v = 0.0;
      if (row < p.m) {
        const double *yp = p.Ypart + (size_t)row + (size_t)col * p.ldy;
        double ys[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) ys[s] = (s < p.nsplit) ? yp[(size_t)s * p.sY] : 0.0;
        y = ((ys[0] + ys[1]) + (ys[2] + ys[3])) + ((ys[4] + ys[5]) + (ys[6] + ys[7]));
        // (up to sixteen partials since round 4 -- short trailing matrices are cut finer; with eight or fewer the sum is
        // the one of rounds 2 - 3 bit for bit: adding +0.0 changes nothing)
        if (p.nsplit > 8) y += ((ys[8] + ys[9]) + (ys[10] + ys[11])) + ((ys[12] + ys[13]) + (ys[14] + ys[15]));
        p.Y[(size_t)row + (size_t)col * (p.ldyo ? p.ldyo : p.ldy)] = y;
        v = p.V[(size_t)row + (size_t)col * p.ldv];
      }
      sY[r * LD + col] = y; sV[r * LD + col] = v;
    }
    __syncthreads();
    slab_gram(sV, sY, acc);
  }
  store_gram(acc, p.Gpart + (size_t)blockIdx.x * SB * SB);
}

// The same with a workgroup per 64 rows AND 16 columns (grid (nch, 4)): a chunk's sixteen partial sums are 0.5 MB, which one
// workgroup per chunk took ~20 us to read while three quarters of the chip had nothing to do below m = 16384 (53
// workgroups at m = 3400).  Each element's sum and each entry of the Gram partial are formed as above, bit for bit.
__global__ __launch_bounds__(256) void yred_q_kernel(YredArgs p) {
  __shared__ double sYq[SB * 17], sV[IMG];
  const int t = threadIdx.x, r = t & 63, cg = t >> 6, q = blockIdx.y;
  const int lane = t & 63, wave = t >> 6, l15 = lane & 15, l4 = lane >> 4;
  const int row0 = blockIdx.x * CH;
  if (row0 >= p.m) return;
  const int row = row0 + r;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int lc = 4 * cg + c, col = 16 * q + lc;
    double y = 0.0;
    if (row < p.m) {
      const double *yp = p.Ypart + (size_t)row + (size_t)col * p.ldy;
      double ys[16];
#pragma unroll
      for (int s = 0; s < 16; ++s) ys[s] = (s < p.nsplit) ? yp[(size_t)s * p.sY] : 0.0;
      y = ((ys[0] + ys[1]) + (ys[2] + ys[3])) + ((ys[4] + ys[5]) + (ys[6] + ys[7]));
      if (p.nsplit > 8) y += ((ys[8] + ys[9]) + (ys[10] + ys[11])) + ((ys[12] + ys[13]) + (ys[14] + ys[15]));
      if (!p.cW) p.Y[(size_t)row + (size_t)col * (p.ldyo ? p.ldyo : p.ldy)] = y;
    }
    sYq[r * 17 + lc] = y;
  }
  if (p.cW) {                                           // (uniform) the pending update of the pair's first panel
    double4_t cacc = (double4_t){0.0, 0.0, 0.0, 0.0};
    for (int pass = 0; pass < 2; ++pass) {
      const double *src = pass ? p.cV : p.cW, *G = pass ? p.cG2 : p.cG1;     // Wp G1, then Vp G2
      __syncthreads();
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const int col = 16 * cg + c;
        sV[r * LD + col] = (row < p.m) ? src[(size_t)row + (size_t)col * p.ldc] : 0.0;
      }
      __syncthreads();
      for (int kk = 0; kk < SB; kk += 4)
        cacc = __builtin_amdgcn_mfma_f64_16x16x4f64(sV[(16 * wave + l15) * LD + kk + l4], G[(16 * q + l15) + SB * (kk + l4)], cacc, 0, 0, 0);
    }
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) sYq[(16 * wave + l4 + 4 * rr) * 17 + l15] -= cacc[rr];
    __syncthreads();
    if (row < p.m) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int lc = 4 * cg + c;
        p.Y[(size_t)row + (size_t)(16 * q + lc) * (p.ldyo ? p.ldyo : p.ldy)] = sYq[r * 17 + lc];
      }
    }
  }
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    const int col = 16 * cg + c;
    sV[r * LD + col] = (row < p.m) ? p.V[(size_t)row + (size_t)col * p.ldv] : 0.0;
  }
  __syncthreads();
  double4_t acc = (double4_t){0.0, 0.0, 0.0, 0.0};
  for (int kk = 0; kk < SB; kk += 4)
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sV[(kk + l4) * LD + 16 * wave + l15], sYq[(kk + l4) * 17 + l15], acc, 0, 0, 0);
  double *G = p.Gpart + (size_t)blockIdx.x * SB * SB;
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) G[(16 * q + l15) + SB * (16 * wave + l4 + 4 * rr)] = acc[rr];
}

// Second panel of a pair: per 64-row chunk the partial products Vp^T V2 and Wp^T V2 (the first panel's V and W against the
// second panel's V, rows of the second panel's trailing matrix); reduce_parts_kernel sums them
struct CorrGramArgs { int m; const double *Vp, *Wp, *V2; int ld; double *G1part, *G2part; };
__global__ __launch_bounds__(256) void corr_gram_kernel(CorrGramArgs p) {
  __shared__ double sX[IMG], sW[IMG], sY[IMG];
  const int t = threadIdx.x, r = t & 63, cg = t >> 6;
  const int row = blockIdx.x * CH + r;
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    const int col = 16 * cg + c;
    const size_t o = (size_t)row + (size_t)col * p.ld;
    const bool in = row < p.m;
    sX[r * LD + col] = in ? p.Vp[o] : 0.0; sW[r * LD + col] = in ? p.Wp[o] : 0.0; sY[r * LD + col] = in ? p.V2[o] : 0.0;
  }
  __syncthreads();
  double4_t a1[4], a2[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { a1[j] = (double4_t){0.0, 0.0, 0.0, 0.0}; a2[j] = (double4_t){0.0, 0.0, 0.0, 0.0}; }
  slab_gram(sX, sY, a1);
  slab_gram(sW, sY, a2);
  store_gram(a1, p.G1part + (size_t)blockIdx.x * SB * SB);
  store_gram(a2, p.G2part + (size_t)blockIdx.x * SB * SB);
}

// How many ways the K range of the SYMM is cut.  Rounds 2 - 4 took "as many splits as fill the chip" (ceil(512 / T) for T
// block rows: two workgroups fit a CU); but the T x nsplit workgroups of a launch all walk the same number of tiles, so a
// launch of 600 of them ran two rounds of which the second was a sixth full (tools/tail_quant.py put 32 of the kernel's
// 76 ms per solve at N = 16384 into such tails).  Measured per launch for every cut from 1 to 16 (round 4, profiles/r04_symm_split_table.csv;
// one solve each; kernel time summed over the 255 panels, + the partials yred_kernel then reads): at most 4 splits 83.8 ms,
// 8: 69.5, 12: 65.0, 16: 62.9 (+ 1.9 ms in yred_kernel); the old rule 76.7; the best cut per T picked from the table 61.9.
// The finer the cut, the better the late workgroups fill the gaps the early ones leave -- so: as fine as the buffer of
// partial sums allows (maxsplit).  Round 5 measured the other end too: the whole product as a list of (block row, K slab)
// units cut into 512 EQUAL consecutive ranges, one per resident workgroup, at most two partial block rows each
// ("stream-K": no tail at all, a third of the partial sums) -- 283 us per sampled launch against 269 at N = 16384 on one
// box, C2 38.8 against 38.4 ms: workgroups that all start together and walk in step lose more than the tails cost.
static void symm_split(int T, int maxsplit, int *nsplit_out, int *tps_out) {
  if (T < 1) T = 1;
  int nsplit = maxsplit;
  if (nsplit > T) nsplit = T;
  const int tps = ceil_div(T, nsplit);
  *nsplit_out = ceil_div(T, tps); *tps_out = tps;
}

// Team form: Y = the sum of the partials symm_lower_kernel<true> wrote for this block row, in a fixed order (the direct
// chunks that reach it, then -- for a block row with owned rows -- the transposed chunks), Gp as above
struct YredDistArgs {
  int m, T, SD, ST, P, rank, r0;
  const double *Ypart; int ldy; long long sY;
  double *Y; int ldyo;
  const double *V; int ldv;
  double *Gpart;
};
// (a workgroup per 64 rows and 16 columns, grid (nch, 4), since round 5 -- as yred_q_kernel: with a workgroup per chunk a
// member's reduction took 58 us per panel at N = 16384 reading up to 56 partial sums one batch of eight after the other)
__global__ __launch_bounds__(256) void yred_dist_kernel(YredDistArgs p) {
  __shared__ double sYq[SB * 17], sV[IMG];
  const int t = threadIdx.x, r = t & 63, cg = t >> 6, q = blockIdx.y;
  const int lane = t & 63, wave = t >> 6, l15 = lane & 15, l4 = lane >> 4;
  const int row0 = blockIdx.x * CH;
  if (row0 >= p.m) return;
  const int row = row0 + r;
  const int rb = row0 >> 7, m0 = rb * 128;
  const bool heavy = ((((p.r0 + m0) >> 7) % p.P) == p.rank) || ((((p.r0 + m0 + 64) >> 7) % p.P) == p.rank);
  const int td = symm_dist_chunk(p.T, p.SD);
  int nd = rb / td + 1; if (nd > p.SD) nd = p.SD;                    // direct chunks with kt0 <= rb
  const int cnt = p.T - rb - 1, len = symm_dist_chunk(cnt > 0 ? cnt : 1, p.ST);
  const int nt = (heavy && cnt > 0) ? symm_dist_chunk(cnt, len) : 0;  // transposed chunks with kt0 < T
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int lc = 4 * cg + c, col = 16 * q + lc;
    double y = 0.0;
    if (row < p.m) {
      const double *yp = p.Ypart + (size_t)row + (size_t)col * p.ldy;
      // (fixed order: direct chunks ascending, then transposed chunks ascending; eight loads in flight at a time)
      auto sum_run = [&](const double *base, int count) {
        int u0 = 0;
        for (; u0 + 8 <= count; u0 += 8) {
          double a[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) a[u] = base[(size_t)(u0 + u) * p.sY];
          y += ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
        }
        for (; u0 < count; ++u0) y += base[(size_t)u0 * p.sY];
      };
      sum_run(yp, nd);
      sum_run(yp + (size_t)p.SD * p.sY, nt);
      p.Y[(size_t)row + (size_t)col * p.ldyo] = y;
    }
    sYq[r * 17 + lc] = y;
  }
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    const int col = 16 * cg + c;
    sV[r * LD + col] = (row < p.m) ? p.V[(size_t)row + (size_t)col * p.ldv] : 0.0;
  }
  __syncthreads();
  double4_t acc = (double4_t){0.0, 0.0, 0.0, 0.0};
  for (int kk = 0; kk < SB; kk += 4)
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sV[(kk + l4) * LD + 16 * wave + l15], sYq[(kk + l4) * 17 + l15], acc, 0, 0, 0);
  double *G = p.Gpart + (size_t)blockIdx.x * SB * SB;
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) G[(16 * q + l15) + SB * (16 * wave + l4 + 4 * rr)] = acc[rr];
}

// W = [Y | V] [T ; -1/2 T^T G T], G = V^T Y (the sum of yred_kernel's partials), written to columns 0..63 and
// 128..191 of the image [W | V | W].  Every workgroup forms the 128 x 64 multiplier itself (two 64^3 products
// on the matrix cores: less than the launch of a kernel that would do it once).
struct WArgs {
  int m;
  const double *Y; int ldy; const double *V; int ldv;
  const double *Gred;        // G, stored (j + 64 i) = G(i, j)
  const double *T;           // column-major, ld 64
  double *W1, *W2; int ldi;  // the two places W goes to in the update's operand images
};
__global__ __launch_bounds__(256) void w_kernel(WArgs p) {
  __shared__ double sS[IMG], sM1[IMG], sM2[IMG];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l15 = lane & 15, l4 = lane >> 4, r = t & 63, cg = t >> 6;
  {
    double gv[16], tv[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) { gv[k] = p.Gred[t + 256 * k]; tv[k] = p.T[t + 256 * k]; }
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int idx = t + 256 * k, lo = idx & 63, hi = idx >> 6;
      sM2[hi * LD + lo] = gv[k];             // G(i = hi, j = lo)
      sS[lo * LD + hi] = tv[k];              // T(i = lo, j = hi), natural
      sM1[hi * LD + lo] = tv[k];             // image of T for the products below: sM1[j][k] = T(k, j)
    }
    __syncthreads();
    double4_t acc[4];
    mm64_acc(sM2, false, sS, false, acc);    // X = G T
    __syncthreads();
    mm64_store(acc, sM2);
    __syncthreads();
    mm64_acc(sS, true, sM2, false, acc);     // T^T X
    __syncthreads();
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
      for (int q = 0; q < 4; ++q) sM2[(16 * jt + l15) * LD + 16 * wave + l4 + 4 * q] = -0.5 * acc[jt][q];   // sM2[j][k] = -1/2 (T^T G T)(k, j)
  }
  for (int slab = 0; slab < CH / SB; ++slab) {
    const int row0 = blockIdx.x * CH + slab * SB;
    if (row0 >= p.m) break;
    const int row = row0 + r, i0 = 16 * wave;
    double4_t acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = (double4_t){0.0, 0.0, 0.0, 0.0};
    for (int pass = 0; pass < 2; ++pass) {
      const double *src = pass ? p.V : p.Y;
      const int lds_ = pass ? p.ldv : p.ldy;
      const double *sM = pass ? sM2 : sM1;
      __syncthreads();
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const int col = 16 * cg + c;
        sS[r * LD + col] = (row < p.m) ? src[(size_t)row + (size_t)col * lds_] : 0.0;
      }
      __syncthreads();
      for (int kk = 0; kk < SB; kk += 4) {
        const double x = sS[(i0 + l15) * LD + kk + l4];
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
          acc[jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, sM[(16 * jt + l15) * LD + kk + l4], acc[jt], 0, 0, 0);
      }
    }
    __syncthreads();
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
      for (int q = 0; q < 4; ++q) sS[(i0 + l4 + 4 * q) * LD + 16 * jt + l15] = acc[jt][q];
    __syncthreads();
    if (row < p.m) {
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const int col = 16 * cg + c;
        const double w = sS[r * LD + col];
        p.W1[(size_t)row + (size_t)col * p.ldi] = w;
        p.W2[(size_t)row + (size_t)col * p.ldi] = w;
      }
    }
  }
}

inline size_t al256(size_t b) { return (b + 255) & ~(size_t)255; }

struct Layout {
  int mpad, nparts, maxsplit;
  size_t off_img, off_img2, off_qt, off_y, off_ypart, off_gpart, off_gpart2, off_small, total;
  size_t off_opa[2] = {0, 0}, off_opb[2] = {0, 0}, off_cg1 = 0, off_cg2 = 0;   // one GPU: operand images of a pair of panels
  // team form (P > 0): the panel message [V | T | tau], the table of the strip updates
  int maxb = 0, npanels = 0;
  size_t off_msg = 0, off_offs = 0, off_dims = 0, off_offs2 = 0, off_dims2 = 0, msg_doubles = 0;
  explicit Layout(int n, int P = 0) {
    mpad = round_up(n > 0 ? n : 1, 128);
    nparts = mpad / CH + 1;
    maxsplit = P > 0 ? 56 : 16;          // (team form: up to 8 direct + 48 transposed chunks per block row)
    size_t o = 0;
    off_img = o; o += (P > 0) ? al256((size_t)mpad * 3 * SB * 8) : 0;      // team form: the [W | V | W] images
    off_img2 = o; o += (P > 0) ? al256((size_t)mpad * 3 * SB * 8) : 0;
    off_qt = o; o += al256((size_t)mpad * SB * 8);
    off_y = o; o += al256((size_t)mpad * SB * 8);
    off_ypart = o; o += al256((size_t)maxsplit * mpad * SB * 8);
    off_gpart = o; o += al256((size_t)nparts * SB * SB * 8);
    off_gpart2 = o; o += al256((size_t)nparts * SB * SB * 8);
    off_small = o; o += al256((size_t)24 * SB * SB * 8);   // [10 * 4096 ..): profile counters
    if (P == 0) {
      // the trailing update's operands for up to two pending panels: A-operand [W1 | V1 | W2 | V2], B-operand
      // [V1 | W1 | V2 | W2], double-buffered for the look-ahead (in place of the two [W | V | W] images)
      for (int q = 0; q < 2; ++q) { off_opa[q] = o; o += al256((size_t)mpad * 4 * SB * 8); off_opb[q] = o; o += al256((size_t)mpad * 4 * SB * 8); }
      off_cg1 = o; o += al256((size_t)nparts * SB * SB * 8);
      off_cg2 = o; o += al256((size_t)nparts * SB * SB * 8);
    }
    if (P > 0) {
      maxb = ceil_div(ceil_div(mpad, 128), P) + 1;
      npanels = ceil_div(n > 0 ? n : 1, SB);
      msg_doubles = (size_t)mpad * SB + SB * SB + SB;
      off_msg = o; o += al256(msg_doubles * 8);
      off_offs = o; o += al256((size_t)npanels * maxb * 3 * sizeof(long long));
      off_dims = o; o += al256((size_t)npanels * maxb * 3 * sizeof(int));
      off_offs2 = o; o += al256((size_t)npanels * maxb * 3 * sizeof(long long));
      off_dims2 = o; o += al256((size_t)npanels * maxb * 3 * sizeof(int));
    }
    total = o;
  }
};

// everything a panel factorisation needs besides the panel
struct ChainBufs {
  int n, mpad;
  double *Qt, *Gpart2, *Gred2, *R1, *R1inv, *M2, *L1, *Rband;
  long long *prof;
  int *pflag;               // the flag of the panel in flight (device)
  int *nzrows;              // [panel]: 1 + the last non-zero 64-row chunk of the panel (zeroed at the start of the stage)
};

void ensure_attrs() {
  static bool attr = false;
  if (attr) return;
  (void)hipFuncSetAttribute((const void *)hr_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                            3 * IMG * (int)sizeof(double));
  (void)hipFuncSetAttribute((const void *)house_small_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (SMALL_ROWS * LD + 2 * IMG) * (int)sizeof(double));
  attr = true;
}

// factorisation of the panel at column c0 of A: V into the update's operand images at Vd1 and (if not null) Vd2
// (leading dimension mpad), T into Tp, the reflectors into Vall, tau into tau1, issued on stream st
void panel_chain(hipStream_t st, const ChainBufs &b, double *A, int lda, double *Vall, int ldv, double *tau1, int *d_flag,
                 int c0, double *Vd1, double *Vd2, double *Tp) {
  const int n = b.n, ldi = b.mpad;
  const int r0 = c0 + SB, m = n - r0;
  double *Ap = A + (size_t)r0 + (size_t)c0 * lda;
  double *Vp = Vall + (size_t)r0 + (size_t)c0 * ldv;
  const int nch = ceil_div(m, CH);
  if (m <= SMALL_MAX) {
    SmallArgs sa{m, Ap, lda, Vp, ldv, Vd1, Vd2, ldi, Tp, tau1 + c0};
    hipLaunchKernelGGL(house_small_kernel, dim3(1), dim3(256), (SMALL_ROWS * LD + 2 * IMG) * sizeof(double), st, sa);
    return;
  }
  PanelArgs pa{};
  pa.m = m; pa.src = Ap; pa.lds_ = lda; pa.Gpart = b.Gpart2; pa.nz = b.nzrows ? b.nzrows + c0 / SB : nullptr;
  hipLaunchKernelGGL(panel_kernel<0>, dim3(nch), dim3(256), 0, st, pa);
  hipLaunchKernelGGL(reduce_parts_kernel, dim3(128), dim3(256), 0, st, nch, b.Gpart2, b.Gred2);
  hipLaunchKernelGGL(chol_kernel, dim3(1), dim3(256), 0, st, b.Gred2, b.R1, b.R1inv, b.pflag, pa.nz);
  pa.M = b.R1; pa.dst = b.Qt; pa.ldd = b.mpad;           // (Q1 = A R1^-1 by substitution against R1 itself)
  hipLaunchKernelGGL(panel_kernel<1>, dim3(nch), dim3(256), 0, st, pa);
  hipLaunchKernelGGL(reduce_parts_kernel, dim3(128), dim3(256), 0, st, nch, b.Gpart2, b.Gred2);
  HrArgs ha{b.Gred2, b.Qt, b.mpad, b.R1, b.M2, Tp, b.L1, b.Rband, tau1 + c0, b.pflag, b.prof};
  hipLaunchKernelGGL(hr_kernel, dim3(1), dim3(256), 3 * IMG * sizeof(double), st, ha);
  // the rescue of a panel CholeskyQR2 could not factor (one workgroup; it leaves at once otherwise), then the final pass
  hipLaunchKernelGGL(house_tall_kernel, dim3(1), dim3(HT), 0, st, m, Ap, lda, tau1 + c0, b.pflag, d_flag,
                     b.nzrows ? b.nzrows + c0 / SB : nullptr);
  PanelArgs pf{};
  pf.m = m; pf.src = b.Qt; pf.lds_ = b.mpad; pf.M = b.M2; pf.L1 = b.L1; pf.Rband = b.Rband;
  pf.Vall = Vp; pf.ldv = ldv; pf.Apanel = Ap; pf.lda = lda; pf.Vd1 = Vd1; pf.Vd2 = Vd2; pf.ldi = ldi; pf.pflag = b.pflag;
  pf.Gpart = b.Gpart2; pf.tau = tau1 + c0; pf.Tout = Tp; pf.arrived = (unsigned *)(b.pflag + 16);
  hipLaunchKernelGGL(panel_kernel<2>, dim3(nch), dim3(256), 0, st, pf);
}

// team form: a panel's message [V (m x 64, ld ldy) | T (64 x 64) | tau (64)] in ONE launch each way (round 5; until
// round 4 three launches on the owner and four on every other member, 6 - 9 us each: 9 ms per rank at N = 16384).
// Grid: ceil(m / 256) x 64 for V plus one row of blocks (blockIdx.x == gridDim.x - 1) for T and tau.
struct PanelMsgArgs {
  int m, ldy;
  double *msg;               // the message
  double *V1; int ld1;       // V in the operand image
  double *V2; int ld2;       // unpack only: V in the reflector matrix (may be null)
  double *T; double *tau;
};
template <bool PACK>
__global__ __launch_bounds__(256) void panel_msg_kernel(PanelMsgArgs p) {
  const int t = threadIdx.x;
  if ((int)blockIdx.x == (int)gridDim.x - 1) {          // T and tau: 4096 + 64 doubles over the 64 blocks of this row
    const size_t vcount = (size_t)p.ldy * SB;
    const int i = blockIdx.y * 256 + t;                 // 0 .. 16383: the first 4160 are used
    if (i < SB * SB) { if (PACK) p.msg[vcount + i] = p.T[i]; else p.T[i] = p.msg[vcount + i]; }
    else if (i < SB * SB + SB) { if (PACK) p.msg[vcount + i] = p.tau[i - SB * SB]; else p.tau[i - SB * SB] = p.msg[vcount + i]; }
    return;
  }
  const int r = blockIdx.x * 256 + t, c = blockIdx.y;
  if (r >= p.m) return;
  if (PACK) p.msg[(size_t)r + (size_t)c * p.ldy] = p.V1[(size_t)r + (size_t)c * p.ld1];
  else {
    const double v = p.msg[(size_t)r + (size_t)c * p.ldy];
    p.V1[(size_t)r + (size_t)c * p.ld1] = v;
    if (p.V2) p.V2[(size_t)r + (size_t)c * p.ld2] = v;
  }
}

// team form: the strips a member updates after panel p (batched GEMM: one problem per owned strip that still has
// columns at or beyond cs = 64 (p + 1) + skip).  Entry (panel, k): element offsets {A, B, C} and dims {M, N, K}; the
// operands are the rows of the image [W | V | W] from the strip's first column on.  skip = 64: the table of the member
// that owns the NEXT panel when the look-ahead has already updated that panel's 64 columns on their own.
__global__ void strip_table_kernel(int n, int lda, int P, int rank, int maxb, long long *offs, int *dims, int skip) {
  const int panel = blockIdx.x, k = threadIdx.x;
  if (k >= maxb) return;
  const int r0 = (panel + 1) * SB, cs = r0 + skip;
  const int Sf = cs / 128;
  const int Sfl = Sf + ((rank - Sf) % P + P) % P;
  const int S = Sfl + k * P;
  long long c0 = (long long)S * 128;
  if (c0 < cs) c0 = cs;
  int M = 0, N = 0;
  if (c0 < n) {
    M = n - (int)c0;
    long long cend = (long long)(S + 1) * 128;
    if (cend > n) cend = n;
    N = (int)(cend - c0);
  }
  const size_t e = (size_t)panel * maxb + k;
  offs[3 * e] = c0 - r0; offs[3 * e + 1] = c0 - r0; offs[3 * e + 2] = c0 * ((long long)lda + 1);
  dims[3 * e] = M; dims[3 * e + 1] = N; dims[3 * e + 2] = 2 * SB;
}

// optional per-panel timing of the team form (tools/team_timing.py): HIP events around every panel chain and around
// every "rest of the trailing update" section, on the streams they run on
struct DistProf {
  bool on = false;
  std::vector<hipEvent_t> ev;      // pairs (begin, end)
  std::vector<int> kind;           // per pair: 0 = panel chain (+ packing), 1 = rest of the update
  hipEvent_t mark(hipStream_t st) {
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    (void)hipEventRecord(e, st);
    ev.push_back(e);
    return e;
  }
};
DistProf g_dprof;
int g_dist_la_min = -1;            // rows from which the team form looks ahead (0: never); -1: environment / default

}  // namespace

size_t sy2sb_work_bytes(int n) { return Layout(n).total; }

// Two panels per trailing update (round 5).  The rank-128 update of the trailing matrix reads and writes 16 bytes of it per
// 128 flops and runs at 37 TFLOP/s; with K = 256 the same kernel does 55 (tools/gemm_shapes.py).  So panels go in PAIRS
// while the trailing matrix is tall enough for that to matter (pair_min rows):
//   first panel p : Y = A22 V as before, W; ONLY the next panel's 64 columns are updated (K = 128);
//   second panel  : factored at once (its chain is exposed: nothing else can run), Y2 = A_stale V2 - W1 (V1^T V2) -
//                   V1 (W1^T V2) -- the SYMM on the matrix the first panel has NOT updated plus two 64-wide corrections
//                   (DLATRD's rule; corr_gram_kernel on the second stream beside the SYMM, applied in yred_q_kernel) --, W2;
//   then ONE update A22' -= [W1 V1 W2 V2] [V1 W1 V2 W2]^T with K = 256, the panel after the pair factored beside it
//   on the second stream (look-ahead) as before.
// The operands of the update live in two images per pair (A-operand [W1 | V1 | W2 | V2], B-operand [V1 | W1 | V2 | W2],
// rows relative to the first panel's trailing matrix, the second panel's blocks 64 rows down), double-buffered over pairs.
// A panel without a partner (short trailing matrices, the last panel) uses the first two blocks with K = 128: the flow of
// rounds 2 - 4.
void sy2sb_lower(hipStream_t s, hipStream_t s2, int n, double *A, int lda, double *Vall, int ldv, double *tau1,
                 int *d_flag, void *work) {
  if (n <= 2) return;
  ensure_attrs();
  static bool evs = false;
  static hipEvent_t evA[2], evB[2], evC, evD;
  if (!evs) {
    for (int q = 0; q < 2; ++q) {
      (void)hipEventCreateWithFlags(&evA[q], hipEventDisableTiming);
      (void)hipEventCreateWithFlags(&evB[q], hipEventDisableTiming);
    }
    (void)hipEventCreateWithFlags(&evC, hipEventDisableTiming);
    (void)hipEventCreateWithFlags(&evD, hipEventDisableTiming);
    evs = true;
  }
  const Layout L(n);
  char *w = (char *)work;
  double *opA[2] = {(double *)(w + L.off_opa[0]), (double *)(w + L.off_opa[1])};
  double *opB[2] = {(double *)(w + L.off_opb[0]), (double *)(w + L.off_opb[1])};
  double *Qt = (double *)(w + L.off_qt), *Y = (double *)(w + L.off_y);
  double *Ypart = (double *)(w + L.off_ypart), *Gpart = (double *)(w + L.off_gpart);
  double *Cg1 = (double *)(w + L.off_cg1), *Cg2 = (double *)(w + L.off_cg2);
  double *sm = (double *)(w + L.off_small);
  double *Gred = sm, *R1 = sm + 4096, *R1inv = sm + 2 * 4096, *M2 = sm + 3 * 4096, *L1 = sm + 5 * 4096, *Rband = sm + 6 * 4096;
  double *Tm[2][2] = {{sm + 4 * 4096, sm + 7 * 4096}, {sm + 9 * 4096, sm + 8 * 4096}};     // [pair buffer][panel of the pair]
  double *G1 = sm + 14 * 4096, *G2 = sm + 15 * 4096;
  // panel-chain scratch of the look-ahead stream (its Gram partials must not meet those of yred)
  double *Gpart2 = (double *)(w + L.off_gpart2), *Gred2 = sm + 11 * 4096;
  const int ldi = L.mpad;
  const bool prof = getenv("EK_SY2SB_PROF") != nullptr;
  if (prof) (void)hipMemsetAsync(sm + 10 * 4096, 0, 128, s);
  // rest of the trailing update: below 8192 rows the staged rank-k kernel (one workgroup per CU: the panel chain
  // of the look-ahead finds room beside it), above the plain 8-wave GEMM (two per CU, 33 against 27 - 32 TFLOP/s
  // alone; the chain is short against the update there).  N = 16384: 0.2025 -> 0.1949 s, N = 32768: 1.364 -> 1.278 s.
  static int staged_max = -1;
  if (staged_max < 0) { const char *e = getenv("EK_SY2SB_STAGED_MAX"); staged_max = e ? atoi(e) : 8192; }
  static int la_min = -1;
  if (la_min < 0) { const char *e = getenv("EK_SY2SB_LOOKAHEAD_MIN"); la_min = e ? atoi(e) : 5120; }
  constexpr int small_max = 4096;  // (N = 4096: stage 14.6 -> 13.9 ms)
  int pair_min = 5120;             // rows of the first panel's trailing matrix from which panels go in pairs (0: never)
  { const char *e = getenv("EK_SY2SB_PAIR_MIN"); if (e) pair_min = atoi(e); }      // (read per call: the tests force pairs at small orders)

  int *nzrows = (int *)(sm + 13 * 4096);               // one word per panel (room for 8192)
  (void)hipMemsetAsync(nzrows, 0, (size_t)ceil_div(n, SB) * sizeof(int), s);
  (void)hipMemsetAsync(sm + 12 * 4096, 0, 128, s);      // the panel's flag and the arrival counter of a rescued panel's outputs
  const ChainBufs cb{n, L.mpad, Qt, Gpart2, Gred2, R1, R1inv, M2, L1, Rband, prof ? (long long *)(sm + 10 * 4096) : nullptr,
                     (int *)(sm + 12 * 4096), nzrows};
  // block j (0..3) of an operand image, from image row `row` on
  auto blk = [&](double *img, int j, int row) -> double * { return img + (size_t)row + (size_t)j * SB * ldi; };
  // chain of the panel at column c0 as panel q (0 / 1) of the pair held in buffer pb
  auto chain = [&](hipStream_t st, int c0, int pb, int q) {
    ek::panel_chain(st, cb, A, lda, Vall, ldv, tau1, d_flag, c0, blk(opA[pb], 2 * q + 1, SB * q), blk(opB[pb], 2 * q, SB * q),
                    Tm[pb][q]);
  };
  // Y = A22 V for the trailing matrix at r0 (m rows), V = panel q of buffer pb; then W into the images.  corr: the
  // matrix is stale by the pair's first panel (q == 1)
  auto symm_and_w = [&](int p, int r0, int m, int pb, int q, bool corr) {
    const int nch = ceil_div(m, CH);
    double *A22 = A + (size_t)r0 + (size_t)r0 * lda;
    const double *V = blk(opA[pb], 2 * q + 1, SB * q);
    const int T = ceil_div(m, 128);
    int nsplit, tps;
    symm_split(T, L.maxsplit, &nsplit, &tps);
    SymmArgs sy{m, A22, lda, V, ldi, Ypart, L.mpad, (long long)L.mpad * SB, T, tps, 1, 0, 0};
    const bool timed = kprof_enabled() && (p % 8 == 0);          // a uniform sample of the panels
    if (timed) kprof_begin(s, kProfSymm);
    hipLaunchKernelGGL(symm_lower_kernel<false>, dim3(T, nsplit), dim3(256), 0, s, sy);
    if (timed) kprof_end(s, kProfSymm);
    YredArgs ya{m, nsplit, Ypart, L.mpad, (long long)L.mpad * SB, Y, V, ldi, Gpart};
    if (corr) {
      ya.cW = blk(opA[pb], 0, SB); ya.cV = blk(opA[pb], 1, SB); ya.ldc = ldi; ya.cG1 = G1; ya.cG2 = G2;
      (void)hipStreamWaitEvent(s, evD, 0);
    }
    hipLaunchKernelGGL(yred_q_kernel, dim3(nch, 4), dim3(256), 0, s, ya);
    hipLaunchKernelGGL(reduce_parts_kernel, dim3(128), dim3(256), 0, s, nch, Gpart, Gred);
    WArgs wa{m, Y, L.mpad, V, ldi, Gred, Tm[pb][q], blk(opA[pb], 2 * q, SB * q), blk(opB[pb], 2 * q + 1, SB * q), ldi};
    hipLaunchKernelGGL(w_kernel, dim3(nch), dim3(256), 0, s, wa);
  };
  // A(r0.., r0..) -= opA(rows off.., K columns) opB(rows off.., K columns)^T on the lower triangle, with the look-ahead:
  // the next panel's 64 columns first, then its chain on the second stream beside the rest
  bool waited = true;        // whether stream s already follows the chain of the panel it is about to use
  auto update = [&](int r0, int m, int pb, int off, int K, int next_pb) {
    double *A22 = A + (size_t)r0 + (size_t)r0 * lda;
    const double *P1 = opA[pb] + off, *P2 = opB[pb] + off;
    const bool has_next = m - SB >= 2;
    if (has_next && m >= la_min) {
      gemm(s, false, true, m, SB, K, -1.0, P1, ldi, P2, ldi, 1.0, A22, lda, true, false, /*small_tiles=*/true);
      (void)hipEventRecord(evA[pb], s);
      // (host order: the rest of the update first, then the launches of the chain -- submitted behind them the
      // update would start ~70 us late on every panel)
      gemm(s, false, true, m - SB, m - SB, K, -1.0, P1 + SB, ldi, P2 + SB, ldi, 1.0,
           A22 + (size_t)SB + (size_t)SB * lda, lda, true, /*staged_rank_k=*/m < staged_max);
      (void)hipStreamWaitEvent(s2, evA[pb], 0);
      chain(s2, r0, next_pb, 0);
      (void)hipEventRecord(evB[next_pb], s2);
      waited = false;
    } else {
      // (short trailing matrices: the 64 x 64 tiling -- a launch of the 128 x 128 one lasts one tile time, 44 us, however
      // few tiles there are)
      gemm(s, false, true, m, m, K, -1.0, P1, ldi, P2, ldi, 1.0, A22, lda, true, false, /*small_tiles=*/m <= small_max);
      if (has_next) chain(s, r0, next_pb, 0);
      waited = true;
    }
  };

  chain(s, 0, 0, 0);
  int p = 0, pb = 0;
  for (int c0 = 0; ; ) {
    const int r0 = c0 + SB, m = n - r0;
    if (m < 2) break;
    if (!waited) (void)hipStreamWaitEvent(s, evB[pb], 0);
    symm_and_w(p, r0, m, pb, 0, false);
    const int m2 = m - SB;                        // the second panel's trailing matrix
    if (pair_min > 0 && m >= pair_min && m2 >= 2) {
      // the second panel's 64 columns alone, its chain, the products of the correction beside its SYMM
      double *A22 = A + (size_t)r0 + (size_t)r0 * lda;
      gemm(s, false, true, m, SB, 2 * SB, -1.0, opA[pb], ldi, opB[pb], ldi, 1.0, A22, lda, true, false, /*small_tiles=*/true);
      chain(s, r0, pb, 1);
      (void)hipEventRecord(evC, s);
      (void)hipStreamWaitEvent(s2, evC, 0);
      const int nch2 = ceil_div(m2, CH);
      CorrGramArgs cg{m2, blk(opA[pb], 1, SB), blk(opA[pb], 0, SB), blk(opA[pb], 3, SB), ldi, Cg1, Cg2};
      hipLaunchKernelGGL(corr_gram_kernel, dim3(nch2), dim3(256), 0, s2, cg);
      hipLaunchKernelGGL(reduce_parts_kernel, dim3(128), dim3(256), 0, s2, nch2, Cg1, G1);
      hipLaunchKernelGGL(reduce_parts_kernel, dim3(128), dim3(256), 0, s2, nch2, Cg2, G2);
      (void)hipEventRecord(evD, s2);
      symm_and_w(p + 1, r0 + SB, m2, pb, 1, true);
      update(r0 + SB, m2, pb, SB, 4 * SB, pb ^ 1);
      c0 += 2 * SB; p += 2;
    } else {
      update(r0, m, pb, 0, 2 * SB, pb ^ 1);
      c0 += SB; p += 1;
    }
    pb ^= 1;
  }
  if (getenv("EK_SY2SB_PROF")) {
    long long h[10];
    (void)hipMemcpyAsync(h, sm + 10 * 4096, sizeof(h), hipMemcpyDeviceToHost, s);
    (void)hipStreamSynchronize(s);
    if (h[9] > 0)
      fprintf(stderr, "[hr_kernel prof] calls %lld; cycles: load %.0f chol %.0f inv %.0f qtop-mm %.0f lu %.0f tsolve+uinv %.0f "
              "r1-load %.0f mm2 %.0f\n", h[9], (double)h[0] / h[9], (double)h[1] / h[9], (double)h[2] / h[9], (double)h[3] / h[9],
              (double)h[4] / h[9], (double)h[5] / h[9], (double)h[6] / h[9], (double)h[7] / h[9]);
  }
}


size_t sy2sb_dist_work_bytes(int n, int nranks) { return Layout(n, nranks > 0 ? nranks : 1).total; }

// Team form of the dense -> band stage (the reference's PDSYTRD is distributed over its grid,
// solver_scalapack_all.f90:59, processes.f90:17-36).  1 x P team, 128-wide column strips of the matrix, strip S on
// rank S mod P -- the layout the distributed reduction to standard form leaves the matrix in, so nothing is
// gathered in front of this stage.  Per panel (64 columns, all inside one strip):
//   * the strip's owner factors the panel (the same CholeskyQR2 + reconstruction chain as on one GPU) and
//     broadcasts [V | T | tau]: ONE message of 64 m + 4160 doubles;
//   * every member forms its part of Y = A22 V from the entries of the lower triangle whose column it owns
//     (symm_lower_kernel<DIST>: 1/P of the flops and of the matrix traffic each), ONE all-reduce of Y (64 m doubles);
//   * W (replicated, small), then the rank-128 update of the member's own strips only (one batched GEMM).
// Two bandwidth-bound exchanges per panel, n/64 panels; a member reads and writes only columns it owns.  V, T,
// tau, and therefore the reflectors for the back-transformation, end up complete and identical on every member;
// the band stays distributed by strips (the caller gathers its 65 diagonals: 8 n^2 / 128 bytes... 65 n doubles).
void sy2sb_dist_set_lookahead(int min_rows) { g_dist_la_min = min_rows; }
void sy2sb_dist_profile(bool on) {
  for (auto &e : g_dprof.ev) (void)hipEventDestroy(e);
  g_dprof.ev.clear(); g_dprof.kind.clear();
  g_dprof.on = on;
}
// after the streams have been synchronised: seconds[0] = all panel chains, seconds[1] = all "rest of the update" sections,
// seconds[2] = the first chain + the sum over the panels of max(chain of panel p + 1, (update of panel p) / P): what the
// two cost a rank of a real team of P when the chain (one rank, the others wait for its broadcast) runs beside the
// update (every rank its 1 / P) -- meaningful for a rehearsal WITHOUT look-ahead, where the sections do not overlap on
// the one GPU
void sy2sb_dist_profile_collect(double *seconds, int P) {
  seconds[0] = seconds[1] = seconds[2] = 0.0;
  double last_update = -1.0;
  for (size_t q = 0; q < g_dprof.kind.size(); ++q) {
    float ms = 0.f;
    if (!(g_dprof.ev[2 * q] && g_dprof.ev[2 * q + 1] && hipEventElapsedTime(&ms, g_dprof.ev[2 * q], g_dprof.ev[2 * q + 1]) == hipSuccess)) continue;
    const double t = ms * 1e-3;
    seconds[g_dprof.kind[q]] += t;
    if (g_dprof.kind[q] == 1) {
      if (last_update >= 0.0) seconds[2] += last_update / (P > 0 ? P : 1);    // (an update no chain followed)
      last_update = t;
    } else {
      const double u = last_update >= 0.0 ? last_update / (P > 0 ? P : 1) : 0.0;
      seconds[2] += (t > u) ? t : u;
      last_update = -1.0;
    }
  }
  if (last_update >= 0.0) seconds[2] += last_update / (P > 0 ? P : 1);
  sy2sb_dist_profile(g_dprof.on);
}

// Look-ahead (round 4; the single-GPU form has had it since round 2): as soon as W of panel p exists, the member that
// owns panel p + 1 updates that panel's 64 columns alone; then the chain of panel p + 1, the broadcast of its
// [V | T | tau] and the unpacking on the other members run on the second stream s2 while every member applies the rest
// of update p to its strips on s.  The images [W | V | W] and T are double-buffered for that.  On a real team the
// owner's chain (about 0.25 ms, alone on its GPU) and the broadcast are then hidden behind the update of the other
// strips as far as that lasts; the all-reduce of Y stays on s behind the SYMM it sums.  Same arithmetic on every
// element as without (EK_SY2SB_DIST_LOOKAHEAD_MIN=0): same bits.
void sy2sb_lower_dist(hipStream_t s, hipStream_t s2, int n, int nmem, const Sy2sbMember *mem, const SytrdExchange &x) {
  if (n <= 2 || nmem <= 0 || nmem > kMaxTeam) return;
  ensure_attrs();
  static bool evs = false;
  static hipEvent_t evA[2], evB[2];
  if (!evs) {
    for (int q = 0; q < 2; ++q) {
      (void)hipEventCreateWithFlags(&evA[q], hipEventDisableTiming);
      (void)hipEventCreateWithFlags(&evB[q], hipEventDisableTiming);
    }
    evs = true;
  }
  int la_env = -1;      // (read per call: the multi-process test compares both forms inside one process group)
  { const char *e = getenv("EK_SY2SB_DIST_LOOKAHEAD_MIN"); if (e) la_env = atoi(e); }
  const int la_min = g_dist_la_min >= 0 ? g_dist_la_min : (la_env >= 0 ? la_env : 1024);
  const bool la_on = s2 != nullptr && la_min > 0;
  const int P = x.nranks;
  const Layout L(n, P);
  const int ldi = L.mpad;
  struct St { double *img[2], *Tm[2], *Qt, *Y, *Ypart, *Gpart, *sm, *msg; long long *offs, *offs2; int *dims, *dims2; ChainBufs cb; };
  St st[kMaxTeam];
  double *msgs[kMaxTeam], *ys[kMaxTeam];
  for (int q = 0; q < nmem; ++q) {
    char *w = (char *)mem[q].work;
    St &m = st[q];
    m.img[0] = (double *)(w + L.off_img); m.img[1] = (double *)(w + L.off_img2);
    m.Qt = (double *)(w + L.off_qt); m.Y = (double *)(w + L.off_y);
    m.Ypart = (double *)(w + L.off_ypart); m.Gpart = (double *)(w + L.off_gpart); m.sm = (double *)(w + L.off_small);
    m.msg = (double *)(w + L.off_msg); m.offs = (long long *)(w + L.off_offs); m.dims = (int *)(w + L.off_dims);
    m.offs2 = (long long *)(w + L.off_offs2); m.dims2 = (int *)(w + L.off_dims2);
    double *sm = m.sm;
    m.Tm[0] = sm + 4 * 4096; m.Tm[1] = sm + 9 * 4096;
    m.cb = ChainBufs{n, L.mpad, m.Qt, (double *)(w + L.off_gpart2), sm + 11 * 4096, sm + 4096, sm + 2 * 4096, sm + 3 * 4096,
                     sm + 5 * 4096, sm + 6 * 4096, nullptr, (int *)(sm + 12 * 4096), (int *)(sm + 13 * 4096)};
    (void)hipMemsetAsync(sm + 13 * 4096, 0, (size_t)ceil_div(n, SB) * sizeof(int), s);
    (void)hipMemsetAsync(sm + 12 * 4096, 0, 128, s);
    msgs[q] = m.msg; ys[q] = m.Y;
    hipLaunchKernelGGL(strip_table_kernel, dim3(L.npanels), dim3(round_up(L.maxb, 64)), 0, s, n, mem[q].lda, P, mem[q].rank,
                       L.maxb, m.offs, m.dims, 0);
    if (la_on)
      hipLaunchKernelGGL(strip_table_kernel, dim3(L.npanels), dim3(round_up(L.maxb, 64)), 0, s, n, mem[q].lda, P, mem[q].rank,
                         L.maxb, m.offs2, m.dims2, SB);
  }
  const int NRB = ceil_div(n, 128);
  // panel p on stream sp into image / T buffer `buf`: the owner factors it and packs [V | T | tau]; one broadcast; everybody
  // else unpacks V into its image and into the reflector matrix, T, tau
  auto issue_panel = [&](hipStream_t sp, int p, int buf) {
    const int c0 = p * SB, r0 = c0 + SB, m = n - r0;
    const int owner = (c0 / 128) % P;
    const int ldy = round_up(m, 2);
    const size_t vcount = (size_t)ldy * SB;
    for (int q = 0; q < nmem; ++q) {
      if (mem[q].rank != owner) continue;
      St &M = st[q];
      hipEvent_t e0 = g_dprof.on ? g_dprof.mark(sp) : nullptr;
      panel_chain(sp, M.cb, mem[q].A, mem[q].lda, mem[q].Vall, mem[q].ldv, mem[q].tau1, mem[q].d_flag, c0,
                  M.img[buf] + (size_t)SB * ldi, nullptr, M.Tm[buf]);
      if (P > 1) {                      // (only the m rows of the panel travel: V is packed with leading dimension ldy)
        PanelMsgArgs pm{m, ldy, M.msg, M.img[buf] + (size_t)SB * ldi, ldi, nullptr, 0, M.Tm[buf], mem[q].tau1 + c0};
        hipLaunchKernelGGL(panel_msg_kernel<true>, dim3(ceil_div(m, 256) + 1, SB), dim3(256), 0, sp, pm);
      }
      if (e0) { g_dprof.mark(sp); g_dprof.kind.push_back(0); }
    }
    if (P > 1) {
      size_t offs[kMaxTeam], counts[kMaxTeam];
      for (int r = 0; r < P; ++r) { offs[r] = 0; counts[r] = (r == owner) ? vcount + SB * SB + SB : 0; }
      x.allgatherv(sp, nmem, mem[0].rank, msgs, offs, counts, P, x.user);
    }
    for (int q = 0; q < nmem; ++q) {
      if (mem[q].rank == owner) continue;
      St &M = st[q];
      PanelMsgArgs pm{m, ldy, M.msg, M.img[buf] + (size_t)SB * ldi, ldi, mem[q].Vall + (size_t)r0 + (size_t)c0 * mem[q].ldv,
                      mem[q].ldv, M.Tm[buf], mem[q].tau1 + c0};
      hipLaunchKernelGGL(panel_msg_kernel<false>, dim3(ceil_div(m, 256) + 1, SB), dim3(256), 0, sp, pm);
    }
  };
  // A(:, own strips) -= [W | V] [V | W]^T rows of the strip: the member's strips from column r0 + skip on
  auto update_strips = [&](int q, int p, int buf, int skip) {
    St &M = st[q];
    const int r0 = (p + 1) * SB, m = n - r0, cs = r0 + skip;
    const int Sf = cs / 128, rank = mem[q].rank;
    const int Sfl = Sf + ((rank - Sf) % P + P) % P;
    if (Sfl >= NRB) return;
    const int nb = ceil_div(NRB - Sfl, P);
    GemmDesc g{};
    g.M = m; g.N = 128; g.K = 2 * SB; g.transA = false; g.transB = true; g.alpha = -1.0; g.beta = 1.0;
    g.A = M.img[buf]; g.lda = ldi; g.strideA = 0; g.B = M.img[buf] + (size_t)SB * ldi; g.ldb = ldi; g.strideB = 0;
    g.C = mem[q].A; g.ldc = mem[q].lda; g.strideC = 0; g.batch = nb; g.lower_only = true;
    g.d_offs = (skip ? M.offs2 : M.offs) + (size_t)p * L.maxb * 3; g.d_dims = (skip ? M.dims2 : M.dims) + (size_t)p * L.maxb * 3;
    g.even_offs = true;          // (operand offsets c0 - r0: strips start on multiples of 128, panels on multiples of 64)
    gemm(s, g);
  };

  issue_panel(s, 0, 0);
  bool waited = true;        // whether stream s already follows the chain / broadcast of the current panel
  int p = 0;
  for (int c0 = 0; ; c0 += SB, ++p) {
    const int r0 = c0 + SB, m = n - r0;
    if (m < 2) break;
    const int cur = p & 1;
    if (!waited) (void)hipStreamWaitEvent(s, evB[cur], 0);
    const int ldy = round_up(m, 2);
    // ---- Y = A22 V: every member its own entries, then the sum over the team
    const int nch = ceil_div(m, CH);
    const int T = ceil_div(m, 128);
    // about three tiles per workgroup (a member's T^2 / P tile products spread over the chip's ~512 workgroup slots; a
    // tile takes a workgroup ~14 us, and the launch lasts as long as its longest workgroup): SD chunks of a block row's
    // owned direct slabs (every P-th strip of its up to T tiles), ST chunks of the up to T transposed tiles of an
    // owning block row
    int SD = ceil_div(T, 3 * P), ST = ceil_div(T, 3);      // (2 ties, 1 / 4 / 6 lose 2 - 7 % of a rank's stage: round 5)
    if (SD < 1) SD = 1; if (SD > 8) SD = 8;
    if (ST < 1) ST = 1; if (ST > 48) ST = 48;
    for (int q = 0; q < nmem; ++q) {
      St &M = st[q];
      double *A22 = mem[q].A + (size_t)r0 + (size_t)r0 * mem[q].lda;
      const double *V = M.img[cur] + (size_t)SB * ldi;
      SymmArgs sy{m, A22, mem[q].lda, V, ldi, M.Ypart, L.mpad, (long long)L.mpad * SB, T, 0, P, mem[q].rank, r0, SD, ST};
      // block rows with owned rows: one per owned strip when A22 starts on a strip boundary, two when it starts in the
      // middle of one (a team of one: all of them)
      const int nown = ceil_div(T + 1, P) + 1;
      const int NH = (P == 1) ? T : (((r0 & 127) == 0) ? nown : 2 * nown);
      hipLaunchKernelGGL(symm_lower_kernel<true>, dim3(T * SD + NH * ST), dim3(256), 0, s, sy);
      YredDistArgs ya{m, T, SD, ST, P, mem[q].rank, r0, M.Ypart, L.mpad, (long long)L.mpad * SB, M.Y, ldy, V, ldi, M.Gpart};
      hipLaunchKernelGGL(yred_dist_kernel, dim3(nch, 4), dim3(256), 0, s, ya);
    }
    // G = V^T Y travels WITH Y (round 5): every member's V^T (its part of Y) -- yred_dist_kernel's partial products, summed
    // here -- rides behind the 64 m doubles of its Y in the same all-reduce, 4096 doubles more; until round 4 every
    // member formed G from the summed Y again (yred_kernel: one more launch per panel on every rank)
    for (int q = 0; q < nmem; ++q)
      hipLaunchKernelGGL(reduce_parts_kernel, dim3(128), dim3(256), 0, s, nch, st[q].Gpart, st[q].Y + (size_t)ldy * SB);
    if (P > 1) x.allreduce(s, nmem, ys, (size_t)ldy * SB + SB * SB, x.user);
    for (int q = 0; q < nmem; ++q) {
      St &M = st[q];
      const double *V = M.img[cur] + (size_t)SB * ldi;
      WArgs wa{m, M.Y, ldy, V, ldi, M.Y + (size_t)ldy * SB, M.Tm[cur], M.img[cur], M.img[cur] + (size_t)2 * SB * ldi, ldi};
      hipLaunchKernelGGL(w_kernel, dim3(nch), dim3(256), 0, s, wa);
    }
    const bool has_next = m - SB >= 2;
    if (la_on && has_next && m >= la_min) {
      const int owner_next = (r0 / 128) % P;
      for (int q = 0; q < nmem; ++q) {       // the next panel's 64 columns first, on their owner
        if (mem[q].rank != owner_next) continue;
        St &M = st[q];
        gemm(s, false, true, m, SB, 2 * SB, -1.0, M.img[cur], ldi, M.img[cur] + (size_t)SB * ldi, ldi, 1.0,
             mem[q].A + (size_t)r0 + (size_t)r0 * mem[q].lda, mem[q].lda, true, false, /*small_tiles=*/true);
      }
      (void)hipEventRecord(evA[cur], s);
      // (the update is handed to its stream FIRST: the launches of a chain and of its broadcast take the host ~0.1 ms, and
      // an update submitted behind them would start that much later -- the trace of round 4's first version showed the
      // two streams' kernels strictly one after the other for exactly that reason)
      hipEvent_t e0 = g_dprof.on ? g_dprof.mark(s) : nullptr;
      for (int q = 0; q < nmem; ++q) update_strips(q, p, cur, mem[q].rank == owner_next ? SB : 0);
      if (e0) { g_dprof.mark(s); g_dprof.kind.push_back(1); }
      (void)hipStreamWaitEvent(s2, evA[cur], 0);
      issue_panel(s2, p + 1, cur ^ 1);
      (void)hipEventRecord(evB[cur ^ 1], s2);
      waited = false;
    } else {
      hipEvent_t e0 = g_dprof.on ? g_dprof.mark(s) : nullptr;
      for (int q = 0; q < nmem; ++q) update_strips(q, p, cur, 0);
      if (e0) { g_dprof.mark(s); g_dprof.kind.push_back(1); }
      if (has_next) issue_panel(s, p + 1, cur ^ 1);
      waited = true;
    }
  }
}

}  // namespace ek
