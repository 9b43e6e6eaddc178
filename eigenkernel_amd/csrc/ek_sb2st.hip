// ek_sb2st.hip -- stage 2 of the two-stage tridiagonalisation: symmetric band (half bandwidth
// kBandW = 64) -> tridiagonal by bulge chasing, Bd = Q2 T Q2^T, and the application of Q2.
//
// (With ek_sy2sb.hip this stands where the whole-path call used the one-stage PDSYTRD of
// solver_scalapack_all.f90:59 and PDORMTR of :115; see the header of ek_sy2sb.hip.)
//
// Bulge chasing (Schwarz / Lang's column-wise scheme, the one LAPACK's *SBTRD descendants and the
// two-stage solvers use): sweep s annihilates column s below the sub-diagonal with a reflector on
// rows s+1 .. s+64, whose two-sided application fills a 64x64 bulge one block further down; the
// first column of the bulge is annihilated by the next reflector of the sweep, and so on to the end
// of the band.  Task (s, k): reflector k of sweep s on I_k = [s+1+64k, s+64(k+1)], applied from the
// left to B_{k-1} = A(I_k, I_{k-1}), from both sides to D_k = A(I_k, I_k), from the right to
// B_k = A(I_{k+1}, I_k).  Task (s+1, j) may run once task (s, j+2) is done: the sweeps form a
// pipeline.
//
// MI355X shape: ONE persistent launch; a workgroup takes sweeps from a ticket counter (in order, so
// a workgroup only ever waits for a sweep whose owner is already running) and walks down the band,
// carrying B_{k-1} in registers from task to task (row per lane, 16 columns per wave).  Sweeps
// synchronise through one progress word per sweep.  The band lives in L2 / Infinity Cache (16 MB at
// n = 16384) and is only ever touched with agent-scope (sc1) loads and stores, so no cache
// maintenance is needed: writer = sc1 stores, s_waitcnt vmcnt(0) in every wave, workgroup barrier,
// sc1 store of the progress word; reader = sc1 poll by one lane, workgroup barrier, sc1 loads
// (MI355X_MICROARCH.md, "Valid forms").  Every spin is bounded; a workgroup that gives up raises an
// abort word that ends all others.
//
// Q2 = prod_s prod_k H(s,k) is applied to the eigenvectors of T in blocks of G = 32 consecutive
// sweeps at equal k (a 95 x 32 parallelogram of reflectors = one compact-WY factor): for a block of
// sweeps the factors are applied with k ascending, blocks of sweeps descending; reflectors of
// different blocks commute unless their row ranges overlap, which this order respects.
#include "ek_common.h"

#include <cstdlib>
#include <vector>

namespace ek {
namespace {

typedef double double4_t __attribute__((ext_vector_type(4)));

constexpr int SB = kBandW;        // 64
constexpr int LDAB = 2 * SB;      // rows of the band array: sub-diagonals 0 .. 127 (band + bulge)
constexpr int DLD = SB + 1;       // LDS image of a diagonal block

__device__ __forceinline__ double ld_sc1(const double *p) {
  return __longlong_as_double((long long)__hip_atomic_load((const unsigned long long *)p, __ATOMIC_RELAXED,
                                                           __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void st_sc1(double *p, double v) {
  __hip_atomic_store((unsigned long long *)p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_wave_barrier();
}

// band <- lower band of A, zero bulge area
__global__ void pack_band_kernel(int n, const double *__restrict__ A, int lda, double *__restrict__ AB) {
  const int c = blockIdx.x;
  for (int d = threadIdx.x; d < LDAB; d += blockDim.x) {
    double v = 0.0;
    if (d <= SB && c + d < n) v = A[(size_t)(c + d) + (size_t)c * lda];
    AB[(size_t)d + (size_t)c * LDAB] = v;
  }
}
__global__ void unpack_de_kernel(int n, const double *__restrict__ AB, double *__restrict__ d, double *__restrict__ e) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n) return;
  d[c] = AB[(size_t)c * LDAB];
  if (c < n - 1) e[c] = AB[1 + (size_t)c * LDAB];
}

struct ChaseArgs {
  int n, nsweeps;
  double *AB;
  double *V2; int ldv2;
  double *tau2; int ldt;
  unsigned *prog;        // [nsweeps] tasks completed per sweep
  unsigned *ctl;         // [0] ticket, [1] abort
};

constexpr unsigned kSpinLimit = 1u << 22;

__global__ __launch_bounds__(256) void chase_kernel(ChaseArgs p) {
  __shared__ double s_v[SB], s_w[SB], s_z[SB];
  __shared__ double s_p[4][SB];
  __shared__ double s_t[4][16 * 65];
  __shared__ double s_D[SB * DLD];
  __shared__ double s_tau;
  __shared__ int s_sweep, s_ok;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int n = p.n;
  double *AB = p.AB;
  while (true) {
    __syncthreads();
    if (t == 0) s_sweep = (int)atomicAdd(&p.ctl[0], 1u);
    __syncthreads();
    const int s = s_sweep;
    if (s >= p.nsweeps) break;
    const int K = (n - 3 - s) / SB + 1;
    const int Kprev = (s > 0) ? (n - 2 - s) / SB + 1 : 0;
    double bp[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) bp[j] = 0.0;
    for (int k = 0; k < K; ++k) {
      // ---- wait until sweep s-1 is far enough ahead
      if (t == 0) {
        int ok = 1;
        if (s > 0) {
          const unsigned need = (unsigned)((k + 3 < Kprev) ? k + 3 : Kprev);
          unsigned spins = 0;
          while (__hip_atomic_load(&p.prog[s - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 63u) == 0u &&
                (spins > kSpinLimit || __hip_atomic_load(&p.ctl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
              ok = 0; break;
            }
          }
        }
        if (!ok) __hip_atomic_store(&p.ctl[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_ok = ok;
      }
      __syncthreads();
      if (!s_ok) return;
      const int i0 = s + 1 + k * SB;                       // first index of I_k
      const int L = (n - i0 < SB) ? n - i0 : SB;           // its length (>= 2)
      const int i1 = i0 + SB;                              // first index of I_{k+1}
      int L1 = n - i1; if (L1 > SB) L1 = SB; if (L1 < 0) L1 = 0;
      const int c0w = 16 * wave;                           // this wave's columns of a block
      // ---- prefetch D_k (lower part of row `lane`) and B_k (row `lane`)
      double dl[16], bk[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int c = c0w + j;
        dl[j] = (lane < L && c <= lane) ? ld_sc1(AB + (size_t)(lane - c) + (size_t)(i0 + c) * LDAB) : 0.0;
        bk[j] = (lane < L1 && c < L) ? ld_sc1(AB + (size_t)(SB + lane - c) + (size_t)(i0 + c) * LDAB) : 0.0;
      }
      // ---- (a) the reflector: x = A(I_k, s) for k = 0, else the first column of B_{k-1}
      if (wave == 0) {
        double x = 0.0;
        if (k == 0) { if (lane < L) x = ld_sc1(AB + (size_t)(1 + lane) + (size_t)s * LDAB); }
        else x = bp[0];
        double ssq = (lane >= 1) ? x * x : 0.0;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) ssq += __shfl_xor(ssq, o, 64);
        const double alpha0 = __shfl(x, 0, 64);
        double beta = alpha0, tau = 0.0, scale = 0.0;
        if (ssq != 0.0) {
          beta = -copysign(hypot(alpha0, sqrt(ssq)), alpha0);
          tau = (beta - alpha0) / beta;
          scale = 1.0 / (alpha0 - beta);
        }
        const double v = (lane == 0) ? 1.0 : x * scale;    // rows >= L carry x = 0
        s_v[lane] = v;
        if (lane == 0) { s_tau = tau; p.tau2[(size_t)k + (size_t)s * p.ldt] = tau; }
        if (lane < L) p.V2[(size_t)(i0 + lane) + (size_t)s * p.ldv2] = v;
        const double xnew = (lane == 0) ? beta : 0.0;
        if (k == 0) { if (lane < L) st_sc1(AB + (size_t)(1 + lane) + (size_t)s * LDAB, xnew); }
        else bp[0] = xnew;
      }
      __syncthreads();
      const double tau = s_tau;
      const double v_r = s_v[lane];
      double vc[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) vc[j] = s_v[c0w + j];
      // ---- (b) B_{k-1} <- H B_{k-1} (rows I_k, columns I_{k-1}), then it is final for this sweep
      if (k > 0) {
        double *st = s_t[wave];
#pragma unroll
        for (int j = 0; j < 16; ++j) st[j * 65 + lane] = v_r * bp[j];
        wave_sync();
        {
          const int j = lane >> 2, q = lane & 3;
          const double *src = st + j * 65 + 16 * q;
          double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
          for (int l = 0; l < 16; l += 4) { a0 += src[l]; a1 += src[l + 1]; a2 += src[l + 2]; a3 += src[l + 3]; }
          double tot = (a0 + a1) + (a2 + a3);
          tot += __shfl_xor(tot, 1, 64);
          tot += __shfl_xor(tot, 2, 64);
          if (q == 0) s_z[c0w + j] = tot;
        }
        wave_sync();
        const int ip = i0 - SB;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const int c = c0w + j;
          if (c > 0) bp[j] -= tau * v_r * s_z[c];          // column 0 is (beta, 0, ..., 0) already
          if (lane < L) st_sc1(AB + (size_t)(SB + lane - c) + (size_t)(ip + c) * LDAB, bp[j]);
        }
      }
      // ---- (c) D_k <- H D_k H
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int c = c0w + j;
        if (c <= lane) { s_D[lane * DLD + c] = dl[j]; s_D[c * DLD + lane] = dl[j]; }
      }
      __syncthreads();
      double dd[16];
      double part = 0.0;
#pragma unroll
      for (int j = 0; j < 16; ++j) { dd[j] = s_D[lane * DLD + c0w + j]; part += dd[j] * vc[j]; }
      s_p[wave][lane] = part;
      __syncthreads();
      const double p_r = tau * ((s_p[0][lane] + s_p[1][lane]) + (s_p[2][lane] + s_p[3][lane]));
      double dot = p_r * v_r;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) dot += __shfl_xor(dot, o, 64);
      const double w_r = p_r - 0.5 * tau * dot * v_r;
      if (wave == 0) s_w[lane] = w_r;
      __syncthreads();
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int c = c0w + j;
        dd[j] -= v_r * s_w[c] + w_r * vc[j];
        if (c <= lane && lane < L) st_sc1(AB + (size_t)(lane - c) + (size_t)(i0 + c) * LDAB, dd[j]);
      }
      // ---- (d) B_k <- B_k H (rows I_{k+1}, columns I_k); carried to the next task in registers
      if (L1 > 0) {
        double q = 0.0;
#pragma unroll
        for (int j = 0; j < 16; ++j) q += bk[j] * vc[j];
        __syncthreads();                                   // s_p is reused
        s_p[wave][lane] = q;
        __syncthreads();
        const double q_r = tau * ((s_p[0][lane] + s_p[1][lane]) + (s_p[2][lane] + s_p[3][lane]));
#pragma unroll
        for (int j = 0; j < 16; ++j) bp[j] = bk[j] - q_r * vc[j];
        if (k == K - 1) {                                  // no further task in this sweep: store it now
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            const int c = c0w + j;
            if (lane < L1 && c < L) st_sc1(AB + (size_t)(SB + lane - c) + (size_t)(i0 + c) * LDAB, bp[j]);
          }
        }
      }
      // ---- publish: all stores of the task have completed before the progress word moves
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (t == 0) __hip_atomic_store(&p.prog[s], (unsigned)(k + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// ------------------------------------------------------------------------ Q2: T factors
constexpr int QG = 32;                 // sweeps per compact-WY block
constexpr int QR = QG + SB;            // rows of a block's window (95 used, 96 with padding)
constexpr int QVLD = QG + 1;           // LDS leading dimension of the V image (row-major)

struct Q2Geom {
  int n, nsweeps, nS, kmax;            // kmax: groups per block of sweeps (uniform index S * kmax + k)
};
__host__ __device__ inline int q2_groups_of_block(int n, int S) {   // number of k for which any reflector exists
  const int s0 = S * QG;
  return (s0 <= n - 3) ? (n - 3 - s0) / SB + 1 : 0;
}

// V image of group (S, k): rows o .. o + QR - 1 (o = S*QG + 1 + k*SB), column i = sweep S*QG + i,
// non-zero for i <= row - o < i + SB where the reflector (s, k) exists.
__device__ __forceinline__ double q2_v_entry(const Q2Geom &g, const double *__restrict__ V2, int ldv2, int S, int k,
                                             int rr, int i) {
  const int s = S * QG + i, o = S * QG + 1 + k * SB, row = o + rr;
  if (s >= g.nsweeps || rr < i || rr >= i + SB || row >= g.n) return 0.0;
  if (s + 1 + k * SB > g.n - 2) return 0.0;                 // task (s, k) does not exist
  return V2[(size_t)row + (size_t)s * ldv2];
}

__global__ __launch_bounds__(64) void q2_tfactor_kernel(Q2Geom g, const double *__restrict__ V2, int ldv2,
                                                        const double *__restrict__ tau2, int ldt,
                                                        double *__restrict__ Tall) {
  __shared__ double sV[QR * QVLD];
  __shared__ double sT[QG * QVLD];
  __shared__ double s_g[QG];
  const int S = blockIdx.y, k = blockIdx.x, lane = threadIdx.x;
  if (k >= q2_groups_of_block(g.n, S)) return;
  for (int idx = lane; idx < QR * QG; idx += 64) {
    const int rr = idx % QR, i = idx / QR;
    sV[rr * QVLD + i] = q2_v_entry(g, V2, ldv2, S, k, rr, i);
  }
  for (int idx = lane; idx < QG * QVLD; idx += 64) sT[idx] = 0.0;
  wave_sync();
  for (int i = 0; i < QG; ++i) {
    const int s = S * QG + i;
    const bool exists = s < g.nsweeps && s + 1 + k * SB <= g.n - 2;
    const double ti = exists ? tau2[(size_t)k + (size_t)s * ldt] : 0.0;
    if (lane < i) {                              // g_a = v_a^T v_i over the common rows [i, a + SB)
      double acc = 0.0;
      for (int rr = i; rr < lane + SB; ++rr) acc += sV[rr * QVLD + lane] * sV[rr * QVLD + i];
      s_g[lane] = acc;
    }
    wave_sync();
    if (lane < i) {
      double a = 0.0;
      for (int l = lane; l < i; ++l) a += sT[lane * QVLD + l] * s_g[l];
      sT[lane * QVLD + i] = -ti * a;
    } else if (lane == i) sT[i * QVLD + i] = ti;
    wave_sync();
  }
  double *T = Tall + ((size_t)S * g.kmax + k) * QG * QG;
  for (int idx = lane; idx < QG * QG; idx += 64) T[idx] = sT[(idx % QG) * QVLD + idx / QG];   // column-major
}

// ------------------------------------------------------------------------ Q2: application
// Workgroup = QNC columns of Z; per block of sweeps S (descending) and k (ascending):
//   W1 = V^T Zw,  W2 = T W1,  Zw -= V W2      on the window Zw = Z(o : o+96, columns)
constexpr int QNC = 32;
constexpr int QZLD = QNC + 2;          // LDS window, row-major
constexpr int QWLD = QNC + 2;

struct Q2ApplyArgs {
  Q2Geom g;
  const double *V2; int ldv2;
  const double *Tall;
  double *Z; int ldz; int ncols;
};

__global__ __launch_bounds__(256, 2) void q2_apply_kernel(Q2ApplyArgs p) {
  __shared__ double sZ[QR * QZLD];
  __shared__ double sV[QR * QVLD];
  __shared__ double sT[QG * QVLD];
  __shared__ double sW1[QG * QWLD], sW2[QG * QWLD];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l15 = lane & 15, l4 = lane >> 4;
  const int col0 = blockIdx.x * QNC;
  const int n = p.g.n;
  double vreg[12], treg[4];
  auto fetch = [&](int S, int k) {       // operands of group (S, k) into registers
#pragma unroll
    for (int q = 0; q < 12; ++q) {
      const int idx = t + 256 * q, rr = idx % QR, i = idx / QR;
      vreg[q] = q2_v_entry(p.g, p.V2, p.ldv2, S, k, rr, i);
    }
    const double *T = p.Tall + ((size_t)S * p.g.kmax + k) * QG * QG;
#pragma unroll
    for (int q = 0; q < 4; ++q) treg[q] = T[t + 256 * q];
  };
  for (int S = p.g.nS - 1; S >= 0; --S) {
    const int KS = q2_groups_of_block(n, S);
    if (KS > 0) fetch(S, 0);
    for (int k = 0; k < KS; ++k) {
      const int o = S * QG + 1 + k * SB;
      __syncthreads();
      // operands -> LDS; window of Z -> LDS
#pragma unroll
      for (int q = 0; q < 12; ++q) { const int idx = t + 256 * q, rr = idx % QR, i = idx / QR; sV[rr * QVLD + i] = vreg[q]; }
#pragma unroll
      for (int q = 0; q < 4; ++q) { const int idx = t + 256 * q; sT[(idx % QG) * QVLD + idx / QG] = treg[q]; }
      for (int idx = t; idx < QR * QNC; idx += 256) {
        const int rr = idx % QR, c = idx / QR, row = o + rr, col = col0 + c;
        sZ[rr * QZLD + c] = (row < n && col < p.ncols) ? p.Z[(size_t)row + (size_t)col * p.ldz] : 0.0;
      }
      __syncthreads();
      if (k + 1 < KS) fetch(S, k + 1);
      // W1 (32 x 32) = V^T Zw: wave -> one 16x16 tile
      {
        const int it = wave & 1, jt = wave >> 1;
        double4_t acc = (double4_t){0.0, 0.0, 0.0, 0.0};
        for (int kk = 0; kk < QR; kk += 4)
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sV[(kk + l4) * QVLD + 16 * it + l15],
                                                     sZ[(kk + l4) * QZLD + 16 * jt + l15], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) sW1[(16 * it + l4 + 4 * r) * QWLD + 16 * jt + l15] = acc[r];
      }
      __syncthreads();
      // W2 = T W1
      {
        const int it = wave & 1, jt = wave >> 1;
        double4_t acc = (double4_t){0.0, 0.0, 0.0, 0.0};
        for (int kk = 0; kk < QG; kk += 4)
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sT[(16 * it + l15) * QVLD + kk + l4],
                                                     sW1[(kk + l4) * QWLD + 16 * jt + l15], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) sW2[(16 * it + l4 + 4 * r) * QWLD + 16 * jt + l15] = acc[r];
      }
      __syncthreads();
      // Zw -= V W2: 6 x 2 tiles, 3 per wave
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const int tile = wave * 3 + q, it = tile % 6, jt = tile / 6;
        double4_t acc = (double4_t){0.0, 0.0, 0.0, 0.0};
        for (int kk = 0; kk < QG; kk += 4)
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sV[(16 * it + l15) * QVLD + kk + l4],
                                                     sW2[(kk + l4) * QWLD + 16 * jt + l15], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) sZ[(16 * it + l4 + 4 * r) * QZLD + 16 * jt + l15] -= acc[r];
      }
      __syncthreads();
      for (int idx = t; idx < QR * QNC; idx += 256) {
        const int rr = idx % QR, c = idx / QR, row = o + rr, col = col0 + c;
        if (row < n && col < p.ncols) p.Z[(size_t)row + (size_t)col * p.ldz] = sZ[rr * QZLD + c];
      }
    }
  }
}

__global__ void forward_abort_kernel(const unsigned *ctl, int *flag) { if (ctl[1]) atomicOr(flag, 4); }

inline size_t al256(size_t b) { return (b + 255) & ~(size_t)255; }

struct Layout {
  int nsweeps, nS, kmax, ldt;
  size_t off_ab, off_tau, off_prog, off_ctl, off_T, total;
  explicit Layout(int n) {
    nsweeps = n > 2 ? n - 2 : 0;
    nS = ceil_div(nsweeps > 0 ? nsweeps : 1, QG);
    kmax = q2_groups_of_block(n, 0); if (kmax < 1) kmax = 1;
    ldt = kmax + 1;
    size_t o = 0;
    off_ab = o; o += al256((size_t)LDAB * (n + 1) * 8);
    off_tau = o; o += al256((size_t)ldt * (nsweeps + 1) * 8);
    off_prog = o; o += al256((size_t)(nsweeps + 1) * 4);
    off_ctl = o; o += 256;
    off_T = o; o += al256((size_t)nS * kmax * QG * QG * 8);
    total = o;
  }
};

}  // namespace

size_t sb2st_work_bytes(int n) { return Layout(n).total; }

// Band (lower band of A, half bandwidth 64) -> d, e; the reflectors go to V2 (n x n, ldv2, zero on
// entry; column s = the reflectors of sweep s stacked) and into the workspace (tau).  *d_flag |= 4
// if the persistent kernel had to be abandoned (a bounded spin ran out).
void sb2st_lower(hipStream_t s, int n, const double *A, int lda, double *d, double *e, double *V2, int ldv2,
                 int *d_flag, void *work) {
  if (n <= 0) return;
  const Layout L(n);
  char *w = (char *)work;
  double *AB = (double *)(w + L.off_ab), *tau2 = (double *)(w + L.off_tau);
  unsigned *prog = (unsigned *)(w + L.off_prog), *ctl = (unsigned *)(w + L.off_ctl);
  hipLaunchKernelGGL(pack_band_kernel, dim3(n), dim3(LDAB), 0, s, n, A, lda, AB);
  (void)hipMemsetAsync(tau2, 0, (size_t)L.ldt * (L.nsweeps + 1) * 8, s);
  (void)hipMemsetAsync(prog, 0, (size_t)(L.nsweeps + 1) * 4 + 0, s);
  (void)hipMemsetAsync(ctl, 0, 256, s);
  if (L.nsweeps > 0) {
    ChaseArgs c{n, L.nsweeps, AB, V2, ldv2, tau2, L.ldt, prog, ctl};
    // enough workgroups for the pipeline (a sweep can start three tasks behind its predecessor)
    int nwg = n / (3 * SB) + 8;
    if (nwg > 256) nwg = 256;
    if (nwg > L.nsweeps) nwg = L.nsweeps;
    if (const char *ev = getenv("EK_SB2ST_WGS")) { const int v = atoi(ev); if (v > 0) nwg = v; }
    hipLaunchKernelGGL(chase_kernel, dim3(nwg), dim3(256), 0, s, c);
  }
  hipLaunchKernelGGL(unpack_de_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, s, n, AB, d, e);
  hipLaunchKernelGGL(forward_abort_kernel, dim3(1), dim3(1), 0, s, ctl, d_flag);   // abort word -> caller's flag
}

// Z(:, 0:ncols) <- Q2 Z with the reflectors left by sb2st_lower (same V2, same workspace)
void sb2st_apply_q2(hipStream_t s, int n, int ncols, const double *V2, int ldv2, double *Z, int ldz, void *work) {
  if (n <= 2 || ncols <= 0) return;
  const Layout L(n);
  char *w = (char *)work;
  const double *tau2 = (const double *)(w + L.off_tau);
  double *Tall = (double *)(w + L.off_T);
  Q2Geom g{n, L.nsweeps, L.nS, L.kmax};
  hipLaunchKernelGGL(q2_tfactor_kernel, dim3(L.kmax, L.nS), dim3(64), 0, s, g, V2, ldv2, tau2, L.ldt, Tall);
  Q2ApplyArgs a{g, V2, ldv2, Tall, Z, ldz, ncols};
  hipLaunchKernelGGL(q2_apply_kernel, dim3(ceil_div(ncols, QNC)), dim3(256), 0, s, a);
}

}  // namespace ek
