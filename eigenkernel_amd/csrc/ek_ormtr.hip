// ek_ormtr.hip -- back-transformation Z <- Q Z, Q = H(0) H(1) ... H(n-2).
// Replaces PDORMTR('L','L','N') at solver_scalapack_all.f90:115 (1x1 grid).
//
// Compact-WY on the matrix cores: reflectors are grouped KB = 512 at a time,
// Q_b = I - V_b T_b V_b^T, and each group costs three GEMMs
//     W1 = V_b^T Z      W2 = T_b W1      Z -= V_b W2
// applied from the last group to the first.  All Gram matrices V_b^T V_b are formed by ONE
// batched GEMM; the triangular factors T_b = [T11, -T11 G12 T22; 0, T22] by ONE batched
// launch for the 128x128 diagonal parts (an LDS image per workgroup) plus two batched 128^3
// GEMMs for the coupling, all before the sweep, so the sweep itself is nothing but large
// GEMMs (wide blocks cut the C traffic per flop of the rank-k update and give the V^T Z product
// 512 output tiles = two resident workgroups per CU).
// V is the explicit unit-lower-trapezoidal reflector matrix the tridiagonalisation writes
// (zeros above the unit diagonal), so no masking is needed inside the GEMMs.
#include "ek_common.h"

namespace ek {
namespace {

constexpr int KB = 512;    // reflectors per compact-WY block
constexpr int SB = 128;    // sub-block factored by one workgroup (KB/SB per block)

// T of one 128-wide sub-block from its Gram matrix and tau (forward, columnwise: DLARFT):
//   T(i,i) = tau_i,  T(0:i, i) = -tau_i * T(0:i,0:i) * G(0:i, i)
// Sub-block sb lives on the diagonal of block sb/(KB/SB): offset (sb%(KB/SB))*128 in G and T (ld = KB).
__global__ __launch_bounds__(128) void larft_kernel(int nrefl, const double *__restrict__ G,
                                                    const double *__restrict__ tau,
                                                    double *__restrict__ T) {
  extern __shared__ double s[];          // SB x SB image of T + one column of G
  double *sT = s, *sg = s + SB * SB;
  const int sb = blockIdx.x, t = threadIdx.x;
  const int c0 = sb * SB;
  const int kb = (nrefl - c0 < SB) ? nrefl - c0 : SB;
  if (kb <= 0) return;
  constexpr int NSUB = KB / SB;
  const size_t off = (size_t)(sb / NSUB) * KB * KB + (size_t)(sb % NSUB) * SB * (KB + 1);
  const double *Gb = G + off;
  double *Tb = T + off;
  for (int idx = t; idx < SB * SB; idx += 128) sT[idx] = 0.0;
  __syncthreads();
  for (int i = 0; i < kb; ++i) {
    const double ti = tau[c0 + i];
    if (t < i) sg[t] = Gb[(size_t)t + (size_t)i * KB];
    __syncthreads();
    if (t < i) {
      double acc = 0.0;
      for (int c = t; c < i; ++c) acc += sT[t + SB * c] * sg[c];
      sT[t + SB * i] = -ti * acc;
    } else if (t == i) {
      sT[t + SB * i] = ti;
    }
    __syncthreads();
  }
  for (int idx = t; idx < SB * SB; idx += 128) Tb[(size_t)(idx & (SB - 1)) + (size_t)(idx >> 7) * KB] = sT[idx];
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

// Gram matrices of the blocks: reflector j is zero above row j + 1, so block b's product starts at row b * KB (a multiple of
// the GEMM's K step: the steps that remain are the same instructions on the same operands -- same bits, half the flops)
__global__ void gram_table_kernel(int nblk, int n, int ldv, long long *offs, int *dims) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nblk) return;
  const long long c0 = (long long)b * KB;
  offs[3 * b] = c0 + c0 * ldv; offs[3 * b + 1] = offs[3 * b]; offs[3 * b + 2] = (long long)b * KB * KB;
  dims[3 * b] = KB; dims[3 * b + 1] = KB; dims[3 * b + 2] = n - (int)c0;
}

}  // namespace

void build_explicit_v(hipStream_t s, int n, const double *A, int lda, double *V, int ldv) {
  if (n <= 0) return;
  hipLaunchKernelGGL(build_v_kernel, dim3(ceil_div(n, 256), n < 4096 ? n : 4096), dim3(256), 0, s,
                     n, A, lda, V, ldv);
}

// W1 = V_b^T Z has KB x ncols outputs and an inner dimension of up to n: with few columns (a selected part of
// the spectrum) that is a few dozen tiles with a very long K.  Then K is cut into kSplitMax slices at most, one
// batch entry per slice into its own partial, and the partials are summed in a fixed order.  The slicing is a
// function of the GLOBAL number of vectors, not of the columns this call holds: a grid cell's piece of the
// eigenvectors stays bit-identical to the 1 x 1 result.
constexpr int kSplitMax = 16;
static int ormtr_split(int ncols_global) {
  const int tiles = (KB / 128) * ceil_div(ncols_global > 0 ? ncols_global : 1, 128);
  int S = 512 / tiles;
  if (S > kSplitMax) S = kSplitMax;
  return S >= 2 ? S : 1;
}

size_t ormtr_work_bytes(int n, int ncols, int ncols_global) {
  const int nblk = ceil_div(n > 1 ? n - 1 : 1, KB);
  const int S = ormtr_split(ncols_global > 0 ? ncols_global : ncols);
  return 2 * al256((size_t)nblk * KB * KB * 8) + al256((size_t)nblk * (KB / 2) * (KB / 2) * 8) +
         2 * al256((size_t)KB * (ncols > 0 ? ncols : 1) * 8) +
         (S > 1 ? al256((size_t)S * KB * (ncols > 0 ? ncols : 1) * 8) : 0);
}

__global__ void sum_partials_kernel(size_t count, int S, const double *__restrict__ P, double *__restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  double a = P[i];
  for (int c = 1; c < S; ++c) a += P[(size_t)c * count + i];
  out[i] = a;
}

// The block reflectors' T factors do not depend on the vectors they are applied to: ormtr_prepare forms them
// (Gram matrices, 128x128 diagonal parts, couplings) into `prep` (>= ormtr_prep_bytes(n)), ormtr_apply then
// needs three GEMMs per block of KB reflectors.  The whole-path call runs the preparation on its second
// stream beside the bulge chasing, which leaves half of the chip idle.
size_t ormtr_prep_bytes(int n) {
  const int nblk = ceil_div(n > 1 ? n - 1 : 1, KB);
  return 2 * al256((size_t)nblk * KB * KB * 8) + al256((size_t)nblk * (KB / 2) * (KB / 2) * 8);
}

void ormtr_prepare(hipStream_t s, int n, const double *V, int ldv, const double *tau, void *prep) {
  const int nrefl = n - 1;
  if (nrefl <= 0) return;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void *)larft_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (SB * SB + SB) * sizeof(double));
    attr = true;
  }
  const int nblk = ceil_div(nrefl, KB);
  char *w = (char *)prep;
  double *T = (double *)w; w += al256((size_t)nblk * KB * KB * 8);
  double *G = (double *)w; w += al256((size_t)nblk * KB * KB * 8);
  double *Tmp = (double *)w;

  // all Gram matrices G_b = V_b^T V_b in one batched GEMM (rows above a block's reflectors
  // are zero in V, so the full column height can be used for every block); a short last
  // block is zero padded
  (void)hipMemsetAsync(G, 0, (size_t)nblk * KB * KB * 8, s);
  (void)hipMemsetAsync(T, 0, (size_t)nblk * KB * KB * 8, s);
  const int full = nrefl / KB;            // blocks with all KB reflectors
  if (full > 0) {
    // (the table of per-block offsets and inner dimensions lives where the couplings' scratch will be: spent by then)
    long long *offs = (long long *)Tmp;
    int *dims = (int *)(offs + 3 * (size_t)full);
    hipLaunchKernelGGL(gram_table_kernel, dim3(ceil_div(full, 256)), dim3(256), 0, s, full, n, ldv, offs, dims);
    GemmDesc g{};
    g.M = KB; g.N = KB; g.K = n; g.transA = true; g.transB = false; g.alpha = 1.0; g.beta = 0.0;
    g.A = V; g.lda = ldv; g.strideA = 0;
    g.B = V; g.ldb = ldv; g.strideB = 0;
    g.C = G; g.ldc = KB; g.strideC = 0;
    g.batch = full; g.lower_only = false;
    g.d_offs = offs; g.d_dims = dims; g.even_offs = (ldv & 1) == 0;
    gemm(s, g);
  }
  if (full < nblk) {
    const int c0 = full * KB, kb = nrefl - c0;
    gemm(s, true, false, kb, kb, n - c0, 1.0, V + (size_t)c0 + (size_t)c0 * ldv, ldv, V + (size_t)c0 + (size_t)c0 * ldv, ldv, 0.0,
         G + (size_t)full * KB * KB, KB);
  }
  // T_b: the 128x128 diagonal parts by one batched LDS kernel, then pairs are coupled bottom-up,
  //   T = [T11, -T11 G12 T22; 0, T22]   (128 -> 256 -> ... -> KB), two batched GEMMs per step
  hipLaunchKernelGGL(larft_kernel, dim3(nblk * (KB / SB)), dim3(128), (SB * SB + SB) * sizeof(double), s,
                     nrefl, G, tau, T);
  for (int sz = SB; sz < KB; sz *= 2) {
    for (int o = 0; o < KB; o += 2 * sz) {
      GemmDesc g{};
      g.M = sz; g.N = sz; g.K = sz; g.transA = false; g.transB = false; g.alpha = 1.0; g.beta = 0.0;
      g.A = G + (size_t)o + (size_t)(o + sz) * KB; g.lda = KB; g.strideA = (long long)KB * KB;         // G12
      g.B = T + (size_t)(o + sz) * (KB + 1); g.ldb = KB; g.strideB = (long long)KB * KB;               // T22
      g.C = Tmp; g.ldc = sz; g.strideC = (long long)sz * sz;
      g.batch = nblk; g.lower_only = false;
      gemm(s, g);
      g.alpha = -1.0;
      g.A = T + (size_t)o * (KB + 1); g.lda = KB; g.strideA = (long long)KB * KB;                      // T11
      g.B = Tmp; g.ldb = sz; g.strideB = (long long)sz * sz;
      g.C = T + (size_t)o + (size_t)(o + sz) * KB; g.ldc = KB; g.strideC = (long long)KB * KB;         // T12
      gemm(s, g);
    }
  }
}

// prep: what ormtr_prepare left (its T factors come first); work: >= 2 * KB * ncols doubles
void ormtr_apply(hipStream_t s, int n, int ncols, const double *V, int ldv, const void *prep, double *Z, int ldz,
                 void *work, int ncols_global) {
  const int nrefl = n - 1;
  if (nrefl <= 0 || ncols <= 0) return;
  const int nblk = ceil_div(nrefl, KB);
  const double *T = (const double *)prep;
  char *w = (char *)work;
  double *W1 = (double *)w; w += al256((size_t)KB * ncols * 8);
  double *W2 = (double *)w; w += al256((size_t)KB * ncols * 8);
  double *Wp = (double *)w;                        // partials of the split form
  const int Smax = ormtr_split(ncols_global > 0 ? ncols_global : ncols);
  for (int b = nblk - 1; b >= 0; --b) {
    const int c0 = b * KB;
    const int kb = (nrefl - c0 < KB) ? nrefl - c0 : KB;
    // rows from c0 on: row c0 of these reflectors is zero (reflector j starts below row j), and starting on an
    // even row keeps the operands 16-byte aligned for the GEMM's paired loads
    const int row0 = c0, m = n - row0;
    const double *Vb = V + (size_t)row0 + (size_t)c0 * ldv;
    double *Zb = Z + row0;
    // slices start on even rows (16-byte operand loads).  Equal slices where m allows it (every block of an order that
    // is a multiple of 512): ONE launch of S x tiles workgroups; with a shorter last slice that slice is a launch of
    // its own, a few dozen workgroups that last as long as the whole batch before them did (until round 4 the slice
    // length was rounded up to 64, which left such a launch behind every other block at N = 16384).
    const int kc = round_up(ceil_div(m, Smax), 2);
    const int S = (Smax > 1 && m >= 4096) ? ceil_div(m, kc) : 1;
    if (S > 1) {
      const size_t cnt = (size_t)KB * ncols;
      const int klast = m - (S - 1) * kc;
      GemmDesc g{};
      g.M = kb; g.N = ncols; g.K = kc; g.transA = true; g.transB = false; g.alpha = 1.0; g.beta = 0.0;
      g.A = Vb; g.lda = ldv; g.strideA = kc; g.B = Zb; g.ldb = ldz; g.strideB = kc;
      g.C = Wp; g.ldc = KB; g.strideC = (long long)cnt; g.batch = (klast == kc) ? S : S - 1;
      gemm(s, g);
      if (klast != kc)
        gemm(s, true, false, kb, ncols, klast, 1.0, Vb + (size_t)(S - 1) * kc, ldv, Zb + (size_t)(S - 1) * kc, ldz, 0.0,
             Wp + (size_t)(S - 1) * cnt, KB);
      hipLaunchKernelGGL(sum_partials_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, s, cnt, S, Wp, W1);
    } else
    gemm(s, true, false, kb, ncols, m, 1.0, Vb, ldv, Zb, ldz, 0.0, W1, KB);
    gemm(s, false, false, kb, ncols, kb, 1.0, T + (size_t)b * KB * KB, KB, W1, KB, 0.0, W2, KB);
    gemm(s, false, false, m, ncols, kb, -1.0, Vb, ldv, W2, KB, 1.0, Zb, ldz);
  }
}

void ormtr_lower(hipStream_t s, int n, int ncols, const double *V, int ldv, const double *tau,
                 double *Z, int ldz, void *work, int ncols_global) {
  if (n - 1 <= 0 || ncols <= 0) return;
  char *prep = (char *)work;
  ormtr_prepare(s, n, V, ldv, tau, prep);
  ormtr_apply(s, n, ncols, V, ldv, prep, Z, ldz, prep + ormtr_prep_bytes(n), ncols_global);
}

}  // namespace ek
