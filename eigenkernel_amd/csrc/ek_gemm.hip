// ek_gemm.hip -- fp64 GEMM on the CDNA4 matrix cores (v_mfma_f64_16x16x4_f64).
//
// Used by every MFMA-bound stage of the path: the trailing SYRK of the Cholesky
// factorisation (reference: generalized_to_standard.f90:24, PDPOTRF), the triangular
// solves of the reduction/recovery (:37 PDSYGST, :103 PDTRTRS), the SYR2K trailing update
// of the tridiagonalisation (solver_scalapack_all.f90:59, PDSYTRD), the merge products of
// the divide & conquer (:96 PDSTEDC) and the block-reflector products of the
// back-transformation (:115 PDORMTR).
//
// Tiling: one 256-thread workgroup (4 waves, 2x2) owns a 128x128 tile of C; each wave a
// 64x64 sub-tile = 4x4 MFMA 16x16 accumulators (64 fp64 = 128 VGPRs per lane).  K is
// walked in steps of 16 through LDS.  The MFMA "A" operand is taken from op(B) and the "B"
// operand from op(A), i.e. the instruction computes the transposed tile: its lane&15 index
// then runs along M, which is the contiguous direction of column-major C, so every store
// instruction writes four 128-byte row segments instead of 64 scattered doubles.
//
// LDS images depend on which direction is contiguous in global memory:
//   MC (M- or N-contiguous operand): s[k][128 + 16]  -- global reads and LDS writes are
//       contiguous along the 128-direction; fragment reads (16 consecutive m per k) are
//       conflict-free because the 1152-byte row stride shifts each k-row by 32 banks.
//   KC (K-contiguous operand):       s[128][16 + 1]  -- global reads run along k, and the
//       17-double row stride keeps both the writes and the fragment reads conflict-free.
#include "ek_common.h"

#include <cstdlib>

namespace ek {
namespace {

typedef double double4_t __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, BK = 16;
constexpr int MC_LD = BM + 16;   // doubles per k-row of an MC image
constexpr int KC_LD = BK + 1;    // doubles per m-row of a KC image
constexpr int TILE_DOUBLES = (BK * MC_LD > BM * KC_LD) ? BK * MC_LD : BM * KC_LD;

struct GemmArgs {
  int M, N, K;
  double alpha, beta;
  const double *A; int lda; long long sA;
  const double *B; int ldb; long long sB;
  double *C; int ldc; long long sC;
  int lower_only;
  int tiles_m, tiles_n;
  const long long *offs;   // optional per-batch element offsets {A, B, C}
  const int *dims;         // optional per-batch {M, N, K} (device memory), <= the host M, N, K
};

// Loads the 128 x 16 slab of an operand into registers (8 doubles per thread).
// KCONTIG = false: element (x, k) at P[x + k*ld]; true: at P[k + x*ld].
template <bool KCONTIG>
__device__ __forceinline__ void load_slab(double (&r)[8], const double *__restrict__ P, int ld,
                                          int x0, int X, int k0, int K, int t) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int idx = t + 256 * i;
    int x, k;
    if (KCONTIG) { k = idx & 15; x = idx >> 4; }
    else         { x = idx & 127; k = idx >> 7; }
    const int gx = x0 + x, gk = k0 + k;
    double v = 0.0;
    if (gx < X && gk < K)
      v = KCONTIG ? P[(size_t)gk + (size_t)gx * ld] : P[(size_t)gx + (size_t)gk * ld];
    r[i] = v;
  }
}

template <bool KCONTIG>
__device__ __forceinline__ void store_slab(const double (&r)[8], double *s, int t) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int idx = t + 256 * i;
    if (KCONTIG) { const int k = idx & 15, x = idx >> 4; s[x * KC_LD + k] = r[i]; }
    else         { const int x = idx & 127, k = idx >> 7; s[k * MC_LD + x] = r[i]; }
  }
}

// The same slab in PAIRS (the VEC variant of gemm_kernel: operands with even leading dimension and 16-byte
// aligned base): MC: pair = rows (2 xp, 2 xp + 1) at one k, thread -> xp = idx & 63, k = idx >> 6;
// KC: pair = k indices (2 kp, 2 kp + 1) of one row, thread -> kp = idx & 7, x = idx >> 3 (idx = t + 256 i).
// Interior slabs are fetched as 16-byte loads without predicates, edge slabs element by element into the
// same registers.
typedef double double2g_t __attribute__((ext_vector_type(2)));
template <bool KCONTIG>
__device__ __forceinline__ void load_slab_v(double (&r)[8], const double *__restrict__ P, int ld, int x0, int X,
                                            int k0, int K, int t) {
  if (x0 + BM <= X && k0 + BK <= K) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = t + 256 * i;
      double2g_t v;
      if (KCONTIG) v = *reinterpret_cast<const double2g_t *>(P + (size_t)(k0 + 2 * (idx & 7)) + (size_t)(x0 + (idx >> 3)) * ld);
      else         v = *reinterpret_cast<const double2g_t *>(P + (size_t)(x0 + 2 * (idx & 63)) + (size_t)(k0 + (idx >> 6)) * ld);
      r[2 * i] = v.x; r[2 * i + 1] = v.y;
    }
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = t + 256 * i;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int gx = x0 + (KCONTIG ? (idx >> 3) : 2 * (idx & 63) + h);
        const int gk = k0 + (KCONTIG ? 2 * (idx & 7) + h : (idx >> 6));
        double v = 0.0;
        if (gx < X && gk < K) v = KCONTIG ? P[(size_t)gk + (size_t)gx * ld] : P[(size_t)gx + (size_t)gk * ld];
        r[2 * i + h] = v;
      }
    }
  }
}
template <bool KCONTIG>
__device__ __forceinline__ void store_slab_v(const double (&r)[8], double *s, int t) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = t + 256 * i;
    if (KCONTIG) { const int k = 2 * (idx & 7), x = idx >> 3; s[x * KC_LD + k] = r[2 * i]; s[x * KC_LD + k + 1] = r[2 * i + 1]; }
    else { const int x = 2 * (idx & 63), k = idx >> 6; *reinterpret_cast<double2g_t *>(&s[k * MC_LD + x]) = (double2g_t){r[2 * i], r[2 * i + 1]}; }
  }
}

template <bool KCONTIG>
__device__ __forceinline__ double frag(const double *s, int x, int k) {
  return KCONTIG ? s[x * KC_LD + k] : s[k * MC_LD + x];
}

// Workgroup -> tile.  Plain: consecutive workgroups walk down M first (they share the B slab in L2); lower_only == 1: the
// same grid, workgroups of tiles above the diagonal leave at once; lower_only == 2 (square tilings, one product, set by
// gemm()): the grid holds the tiles on and below the diagonal only, column by column -- a trailing update of 120 x 120
// tiles no longer dispatches 7140 workgroups that have nothing to do.
__device__ __forceinline__ bool tile_of(const GemmArgs &p, int tile, int &tm, int &tn) {
  if (p.lower_only == 2) {
    const long long T = p.tiles_m, c = tile;
    int j = (int)(((double)(2 * T + 1) - sqrt((double)((2 * T + 1) * (2 * T + 1) - 8 * c))) * 0.5);
    if (j < 0) j = 0;
    if (j > p.tiles_n - 1) j = p.tiles_n - 1;
    while (j + 1 < p.tiles_n && (long long)(j + 1) * T - (long long)(j + 1) * j / 2 <= c) ++j;
    while (j > 0 && (long long)j * T - (long long)j * (j - 1) / 2 > c) --j;
    tn = j; tm = j + (int)(c - ((long long)j * T - (long long)j * (j - 1) / 2));
    return tm < p.tiles_m;
  }
  tm = tile % p.tiles_m; tn = tile / p.tiles_m;
  return true;
}

template <bool TA, bool TB, bool VEC>
__global__ __launch_bounds__(256, 2) void gemm_kernel(GemmArgs p) {
  __shared__ __attribute__((aligned(16))) double smem[2 * TILE_DOUBLES];
  double *sA = smem, *sB = smem + TILE_DOUBLES;

  // tile index: consecutive workgroups walk down M first (they share the B slab in L2)
  int tm, tn;
  if (!tile_of(p, blockIdx.x, tm, tn)) return;
  const int m0 = tm * BM, n0 = tn * BN;
  if (p.lower_only && n0 > m0 + BM - 1) return;
  if (p.dims) {
    p.M = p.dims[3 * blockIdx.y]; p.N = p.dims[3 * blockIdx.y + 1]; p.K = p.dims[3 * blockIdx.y + 2];
    if (m0 >= p.M || n0 >= p.N) return;
  }

  const double *__restrict__ A = p.A + (size_t)blockIdx.y * p.sA;
  const double *__restrict__ B = p.B + (size_t)blockIdx.y * p.sB;
  double *__restrict__ C = p.C + (size_t)blockIdx.y * p.sC;
  if (p.offs) {
    A += p.offs[3 * blockIdx.y]; B += p.offs[3 * blockIdx.y + 1]; C += p.offs[3 * blockIdx.y + 2];
  }

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = (wave & 1) * 64, wn = (wave >> 1) * 64;
  const int l15 = lane & 15, l4 = lane >> 4;

  double4_t acc[4][4];   // [ni][mi]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (double4_t){0.0, 0.0, 0.0, 0.0};

  double ra[8], rb[8];
  // op(A) is M x K: not transposed -> M-contiguous (MC); transposed -> K-contiguous (KC)
  // op(B) is K x N: not transposed -> K-contiguous (KC); transposed -> N-contiguous (MC)
  if (VEC) {
    load_slab_v<TA>(ra, A, p.lda, m0, p.M, 0, p.K, t);
    load_slab_v<!TB>(rb, B, p.ldb, n0, p.N, 0, p.K, t);
  } else {
    load_slab<TA>(ra, A, p.lda, m0, p.M, 0, p.K, t);
    load_slab<!TB>(rb, B, p.ldb, n0, p.N, 0, p.K, t);
  }

  for (int k0 = 0; k0 < p.K; k0 += BK) {
    __syncthreads();
    if (VEC) { store_slab_v<TA>(ra, sA, t); store_slab_v<!TB>(rb, sB, t); }
    else { store_slab<TA>(ra, sA, t); store_slab<!TB>(rb, sB, t); }
    __syncthreads();
    if (k0 + BK < p.K) {
      if (VEC) {
        load_slab_v<TA>(ra, A, p.lda, m0, p.M, k0 + BK, p.K, t);
        load_slab_v<!TB>(rb, B, p.ldb, n0, p.N, k0 + BK, p.K, t);
      } else {
        load_slab<TA>(ra, A, p.lda, m0, p.M, k0 + BK, p.K, t);
        load_slab<!TB>(rb, B, p.ldb, n0, p.N, k0 + BK, p.K, t);
      }
    }
#pragma unroll
    for (int kk = 0; kk < BK; kk += 4) {
      double fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        fa[i] = frag<TA>(sA, wm + i * 16 + l15, kk + l4);
        fb[i] = frag<!TB>(sB, wn + i * 16 + l15, kk + l4);
      }
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
          acc[ni][mi] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[ni], fa[mi], acc[ni][mi], 0, 0, 0);
    }
  }

  // D[i][j]: i = n index = (lane>>4) + 4*reg, j = m index = lane&15
  const double alpha = p.alpha, beta = p.beta;
#pragma unroll
  for (int ni = 0; ni < 4; ++ni)
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      const int m = m0 + wm + mi * 16 + l15;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + wn + ni * 16 + l4 + 4 * r;
        if (m < p.M && n < p.N) {
          double *c = C + (size_t)m + (size_t)n * p.ldc;
          double v = alpha * acc[ni][mi][r];
          if (beta != 0.0) v += beta * *c;
          *c = v;
        }
      }
    }
}

// (Measured and withdrawn, round 4: the same kernel with K walked in steps of 32 -- half the barrier pairs, 2 x 36 KB of
// LDS, sixteen doubles of prefetch per operand: 256 VGPRs with 11 - 30 spilled, 54 TFLOP/s against 68 on the (n/2)^3
// shapes.)

// ---------------------------------------------------------------------------------------
// 8-wave variant of the 128x128 tile: 512 threads, each wave a 64x32 slice (4x2 MFMA tiles,
// 64 accumulator VGPRs), so two workgroups = 4 waves per SIMD are resident.
template <bool KCONTIG>
__device__ __forceinline__ void load_slab8(double (&r)[4], const double *__restrict__ P, int ld,
                                           int x0, int X, int k0, int K, int t) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = t + 512 * i;
    int x, k;
    if (KCONTIG) { k = idx & 15; x = idx >> 4; }
    else         { x = idx & 127; k = idx >> 7; }
    const int gx = x0 + x, gk = k0 + k;
    double v = 0.0;
    if (gx < X && gk < K)
      v = KCONTIG ? P[(size_t)gk + (size_t)gx * ld] : P[(size_t)gx + (size_t)gk * ld];
    r[i] = v;
  }
}
template <bool KCONTIG>
__device__ __forceinline__ void store_slab8(const double (&r)[4], double *s, int t) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = t + 512 * i;
    if (KCONTIG) { const int k = idx & 15, x = idx >> 4; s[x * KC_LD + k] = r[i]; }
    else         { const int x = idx & 127, k = idx >> 7; s[k * MC_LD + x] = r[i]; }
  }
}

// paired form for the VEC variant (see load_slab_v): idx = t + 512 i, i < 2
template <bool KCONTIG>
__device__ __forceinline__ void load_slab8_v(double (&r)[4], const double *__restrict__ P, int ld, int x0, int X,
                                             int k0, int K, int t) {
  if (x0 + BM <= X && k0 + BK <= K) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = t + 512 * i;
      double2g_t v;
      if (KCONTIG) v = *reinterpret_cast<const double2g_t *>(P + (size_t)(k0 + 2 * (idx & 7)) + (size_t)(x0 + (idx >> 3)) * ld);
      else         v = *reinterpret_cast<const double2g_t *>(P + (size_t)(x0 + 2 * (idx & 63)) + (size_t)(k0 + (idx >> 6)) * ld);
      r[2 * i] = v.x; r[2 * i + 1] = v.y;
    }
  } else {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = t + 512 * i;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int gx = x0 + (KCONTIG ? (idx >> 3) : 2 * (idx & 63) + h);
        const int gk = k0 + (KCONTIG ? 2 * (idx & 7) + h : (idx >> 6));
        double v = 0.0;
        if (gx < X && gk < K) v = KCONTIG ? P[(size_t)gk + (size_t)gx * ld] : P[(size_t)gx + (size_t)gk * ld];
        r[2 * i + h] = v;
      }
    }
  }
}
template <bool KCONTIG>
__device__ __forceinline__ void store_slab8_v(const double (&r)[4], double *s, int t) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int idx = t + 512 * i;
    if (KCONTIG) { const int k = 2 * (idx & 7), x = idx >> 3; s[x * KC_LD + k] = r[2 * i]; s[x * KC_LD + k + 1] = r[2 * i + 1]; }
    else { const int x = 2 * (idx & 63), k = idx >> 6; *reinterpret_cast<double2g_t *>(&s[k * MC_LD + x]) = (double2g_t){r[2 * i], r[2 * i + 1]}; }
  }
}

template <bool TA, bool TB, bool VEC>
__global__ __launch_bounds__(512, 4) void gemm_kernel_w8(GemmArgs p) {
  __shared__ __attribute__((aligned(16))) double smem[2 * TILE_DOUBLES];
  double *sA = smem, *sB = smem + TILE_DOUBLES;
  int tm, tn;
  if (!tile_of(p, blockIdx.x, tm, tn)) return;
  const int m0 = tm * BM, n0 = tn * BN;
  if (p.lower_only && n0 > m0 + BM - 1) return;
  if (p.dims) {
    p.M = p.dims[3 * blockIdx.y]; p.N = p.dims[3 * blockIdx.y + 1]; p.K = p.dims[3 * blockIdx.y + 2];
    if (m0 >= p.M || n0 >= p.N) return;
  }
  const double *__restrict__ A = p.A + (size_t)blockIdx.y * p.sA;
  const double *__restrict__ B = p.B + (size_t)blockIdx.y * p.sB;
  double *__restrict__ C = p.C + (size_t)blockIdx.y * p.sC;
  if (p.offs) {
    A += p.offs[3 * blockIdx.y]; B += p.offs[3 * blockIdx.y + 1]; C += p.offs[3 * blockIdx.y + 2];
  }
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = (wave & 1) * 64, wn = (wave >> 1) * 32;
  const int l15 = lane & 15, l4 = lane >> 4;
  double4_t acc[2][4];   // [ni][mi]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (double4_t){0.0, 0.0, 0.0, 0.0};
  double ra0[4], rb0[4];
  if (VEC) {
    load_slab8_v<TA>(ra0, A, p.lda, m0, p.M, 0, p.K, t);
    load_slab8_v<!TB>(rb0, B, p.ldb, n0, p.N, 0, p.K, t);
  } else {
    load_slab8<TA>(ra0, A, p.lda, m0, p.M, 0, p.K, t);
    load_slab8<!TB>(rb0, B, p.ldb, n0, p.N, 0, p.K, t);
  }
  for (int k0 = 0; k0 < p.K; k0 += BK) {
    __syncthreads();
    if (VEC) { store_slab8_v<TA>(ra0, sA, t); store_slab8_v<!TB>(rb0, sB, t); }
    else { store_slab8<TA>(ra0, sA, t); store_slab8<!TB>(rb0, sB, t); }
    __syncthreads();
    if (k0 + BK < p.K) {
      if (VEC) {
        load_slab8_v<TA>(ra0, A, p.lda, m0, p.M, k0 + BK, p.K, t);
        load_slab8_v<!TB>(rb0, B, p.ldb, n0, p.N, k0 + BK, p.K, t);
      } else {
        load_slab8<TA>(ra0, A, p.lda, m0, p.M, k0 + BK, p.K, t);
        load_slab8<!TB>(rb0, B, p.ldb, n0, p.N, k0 + BK, p.K, t);
      }
    }
#pragma unroll
    for (int kk = 0; kk < BK; kk += 4) {
      double fa[4], fb[2];
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = frag<TA>(sA, wm + i * 16 + l15, kk + l4);
#pragma unroll
      for (int i = 0; i < 2; ++i) fb[i] = frag<!TB>(sB, wn + i * 16 + l15, kk + l4);
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
          acc[ni][mi] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[ni], fa[mi], acc[ni][mi], 0, 0, 0);
    }
  }
  const double alpha = p.alpha, beta = p.beta;
#pragma unroll
  for (int ni = 0; ni < 2; ++ni)
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      const int m = m0 + wm + mi * 16 + l15;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + wn + ni * 16 + l4 + 4 * r;
        if (m < p.M && n < p.N) {
          double *c = C + (size_t)m + (size_t)n * p.ldc;
          double v = alpha * acc[ni][mi][r];
          if (beta != 0.0) v += beta * *c;
          *c = v;
        }
      }
    }
}

// ---------------------------------------------------------------------------------------
// Small-grid variant: 64x64 tile, BK = 32, same 256 threads (2x2 waves of 32x32 = 2x2 MFMA
// tiles).  Used when the 128x128 tiling would leave most CUs idle (the leaves of the
// recursive Cholesky / triangular solves, low D&C levels): 4x the workgroups and half the
// K iterations, so these latency-bound launches finish in roughly a third of the time.
constexpr int SM = 64, SN = 64, SK = 32;
constexpr int SMC_LD = SM + 16;   // 640-byte row stride: k-rows alternate bank halves
constexpr int SKC_LD = 49;        // 98 dwords = 34 mod 64: conflict-free fragment reads
constexpr int STILE_DOUBLES = (SK * SMC_LD > SM * SKC_LD) ? SK * SMC_LD : SM * SKC_LD;

template <bool KCONTIG>
__device__ __forceinline__ void sload_slab(double (&r)[8], const double *__restrict__ P, int ld,
                                           int x0, int X, int k0, int K, int t) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int idx = t + 256 * i;
    int x, k;
    if (KCONTIG) { k = idx & 31; x = idx >> 5; }
    else         { x = idx & 63; k = idx >> 6; }
    const int gx = x0 + x, gk = k0 + k;
    double v = 0.0;
    if (gx < X && gk < K)
      v = KCONTIG ? P[(size_t)gk + (size_t)gx * ld] : P[(size_t)gx + (size_t)gk * ld];
    r[i] = v;
  }
}
template <bool KCONTIG>
__device__ __forceinline__ void sstore_slab(const double (&r)[8], double *s, int t) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int idx = t + 256 * i;
    if (KCONTIG) { const int k = idx & 31, x = idx >> 5; s[x * SKC_LD + k] = r[i]; }
    else         { const int x = idx & 63, k = idx >> 6; s[k * SMC_LD + x] = r[i]; }
  }
}
// the same slab as 16-byte pairs (interior slabs of operands with even leading dimension and aligned base: the VEC
// conditions of gemm()): MC: pair = rows (2 xp, 2 xp + 1) at one k, xp = idx & 31, k = idx >> 5; KC: pair = k indices
// (2 kp, 2 kp + 1) of one row, kp = idx & 15, x = idx >> 4 (idx = t + 256 i, i < 4).  Same LDS images.
template <bool KCONTIG>
__device__ __forceinline__ void sload_slab_v(double (&r)[8], const double *__restrict__ P, int ld, int x0, int X,
                                             int k0, int K, int t) {
  if (x0 + SM <= X && k0 + SK <= K) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = t + 256 * i;
      double2g_t v;
      if (KCONTIG) v = *reinterpret_cast<const double2g_t *>(P + (size_t)(k0 + 2 * (idx & 15)) + (size_t)(x0 + (idx >> 4)) * ld);
      else         v = *reinterpret_cast<const double2g_t *>(P + (size_t)(x0 + 2 * (idx & 31)) + (size_t)(k0 + (idx >> 5)) * ld);
      r[2 * i] = v.x; r[2 * i + 1] = v.y;
    }
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = t + 256 * i;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int gx = x0 + (KCONTIG ? (idx >> 4) : 2 * (idx & 31) + h);
        const int gk = k0 + (KCONTIG ? 2 * (idx & 15) + h : (idx >> 5));
        double v = 0.0;
        if (gx < X && gk < K) v = KCONTIG ? P[(size_t)gk + (size_t)gx * ld] : P[(size_t)gx + (size_t)gk * ld];
        r[2 * i + h] = v;
      }
    }
  }
}
template <bool KCONTIG>
__device__ __forceinline__ void sstore_slab_v(const double (&r)[8], double *s, int t) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = t + 256 * i;
    if (KCONTIG) { const int k = 2 * (idx & 15), x = idx >> 4; s[x * SKC_LD + k] = r[2 * i]; s[x * SKC_LD + k + 1] = r[2 * i + 1]; }
    else { const int x = 2 * (idx & 31), k = idx >> 5; *reinterpret_cast<double2g_t *>(&s[k * SMC_LD + x]) = (double2g_t){r[2 * i], r[2 * i + 1]}; }
  }
}
template <bool KCONTIG>
__device__ __forceinline__ double sfrag(const double *s, int x, int k) {
  return KCONTIG ? s[x * SKC_LD + k] : s[k * SMC_LD + x];
}

template <bool TA, bool TB, bool VEC>
__global__ __launch_bounds__(256, 2) void gemm_small_kernel(GemmArgs p) {
  __shared__ __attribute__((aligned(16))) double smem[2 * STILE_DOUBLES];
  double *sA = smem, *sB = smem + STILE_DOUBLES;
  int tm, tn;
  if (!tile_of(p, blockIdx.x, tm, tn)) return;
  const int m0 = tm * SM, n0 = tn * SN;
  if (p.lower_only && n0 > m0 + SM - 1) return;
  if (p.dims) {
    p.M = p.dims[3 * blockIdx.y]; p.N = p.dims[3 * blockIdx.y + 1]; p.K = p.dims[3 * blockIdx.y + 2];
    if (m0 >= p.M || n0 >= p.N) return;
  }
  const double *__restrict__ A = p.A + (size_t)blockIdx.y * p.sA;
  const double *__restrict__ B = p.B + (size_t)blockIdx.y * p.sB;
  double *__restrict__ C = p.C + (size_t)blockIdx.y * p.sC;
  if (p.offs) {
    A += p.offs[3 * blockIdx.y]; B += p.offs[3 * blockIdx.y + 1]; C += p.offs[3 * blockIdx.y + 2];
  }
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = (wave & 1) * 32, wn = (wave >> 1) * 32;
  const int l15 = lane & 15, l4 = lane >> 4;
  double4_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (double4_t){0.0, 0.0, 0.0, 0.0};
  double ra[8], rb[8];
  if (VEC) { sload_slab_v<TA>(ra, A, p.lda, m0, p.M, 0, p.K, t); sload_slab_v<!TB>(rb, B, p.ldb, n0, p.N, 0, p.K, t); }
  else { sload_slab<TA>(ra, A, p.lda, m0, p.M, 0, p.K, t); sload_slab<!TB>(rb, B, p.ldb, n0, p.N, 0, p.K, t); }
  for (int k0 = 0; k0 < p.K; k0 += SK) {
    __syncthreads();
    if (VEC) { sstore_slab_v<TA>(ra, sA, t); sstore_slab_v<!TB>(rb, sB, t); }
    else { sstore_slab<TA>(ra, sA, t); sstore_slab<!TB>(rb, sB, t); }
    __syncthreads();
    if (k0 + SK < p.K) {
      if (VEC) { sload_slab_v<TA>(ra, A, p.lda, m0, p.M, k0 + SK, p.K, t); sload_slab_v<!TB>(rb, B, p.ldb, n0, p.N, k0 + SK, p.K, t); }
      else { sload_slab<TA>(ra, A, p.lda, m0, p.M, k0 + SK, p.K, t); sload_slab<!TB>(rb, B, p.ldb, n0, p.N, k0 + SK, p.K, t); }
    }
#pragma unroll
    for (int kk = 0; kk < SK; kk += 4) {
      double fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        fa[i] = sfrag<TA>(sA, wm + i * 16 + l15, kk + l4);
        fb[i] = sfrag<!TB>(sB, wn + i * 16 + l15, kk + l4);
      }
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
          acc[ni][mi] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[ni], fa[mi], acc[ni][mi], 0, 0, 0);
    }
  }
  const double alpha = p.alpha, beta = p.beta;
#pragma unroll
  for (int ni = 0; ni < 2; ++ni)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const int m = m0 + wm + mi * 16 + l15;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + wn + ni * 16 + l4 + 4 * r;
        if (m < p.M && n < p.N) {
          double *c = C + (size_t)m + (size_t)n * p.ldc;
          double v = alpha * acc[ni][mi][r];
          if (beta != 0.0) v += beta * *c;
          *c = v;
        }
      }
    }
}

// ---------------------------------------------------------------------------------------
// Rank-k update C += alpha A B^T with short K for callers that overlap it with a second stream (the SYR2K
// of the dense -> band stage, GemmDesc::staged_rank_k).  A workgroup of 8 waves owns a 128x128 tile and
// takes K in stages of 64: both operand slabs of a stage (2 x 128 x 64) are in LDS at once (one barrier
// pair per 64 of K), fetched as 16-byte vectors, the next stage and the C tile are in flight in registers
// while the matrix cores work on the current one.  One workgroup per CU: ALONE it is slower than the
// 8-wave kernel above (K = 128, n = 16384 lower: 26 vs 32 TFLOP/s -- prologue and epilogue of a tile are not
// covered by a second workgroup), but the dense -> band stage as a whole is faster with it (0.288 -> 0.268 s
// at N = 16384): the panel chain of the look-ahead stream gets its workgroups dispatched sooner.
constexpr int RK = 32;                    // K per stage (64: 0.2335 s, 32: 0.2276 s for sy2sb at N = 16384)
constexpr int RK_LD = BM + 16;            // doubles per k-row of a slab image (conflict-free fragment reads)
typedef double double2_t __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(512) void gemm_rankk_kernel(GemmArgs p) {
  extern __shared__ double rk_smem[];
  double *sA = rk_smem, *sB = rk_smem + RK * RK_LD;
  int tm, tn;
  if (!tile_of(p, blockIdx.x, tm, tn)) return;
  const int m0 = tm * BM, n0 = tn * BN;
  if (p.lower_only && n0 > m0 + BM - 1) return;
  if (p.dims) {
    p.M = p.dims[3 * blockIdx.y]; p.N = p.dims[3 * blockIdx.y + 1]; p.K = p.dims[3 * blockIdx.y + 2];
    if (m0 >= p.M || n0 >= p.N) return;
  }
  const double *__restrict__ A = p.A + (size_t)blockIdx.y * p.sA;
  const double *__restrict__ B = p.B + (size_t)blockIdx.y * p.sB;
  double *__restrict__ C = p.C + (size_t)blockIdx.y * p.sC;
  if (p.offs) { A += p.offs[3 * blockIdx.y]; B += p.offs[3 * blockIdx.y + 1]; C += p.offs[3 * blockIdx.y + 2]; }
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = (wave & 1) * 64, wn = (wave >> 1) * 32;
  const int l15 = lane & 15, l4 = lane >> 4;
  // 16-byte operand fetch: thread -> (row pair xp = t & 63, k = t >> 6 + 8 i), RK / 8 per operand and stage;
  // legal when the leading dimension and the base are even (the library's work arrays are), else scalars
  const bool vec = ((p.lda | p.ldb) & 1) == 0 && ((((size_t)A | (size_t)B) & 15) == 0);
  const int xp = 2 * (t & 63), kq = t >> 6;
  constexpr int NF = RK / 8;            // row pairs per thread, operand and stage
  double2_t ra[NF], rb[NF];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int i = 0; i < NF; ++i) {
      const int k = k0 + kq + 8 * i;
      double2_t va = (double2_t){0.0, 0.0}, vb = (double2_t){0.0, 0.0};
      if (k < p.K) {
        const double *pa = A + (size_t)(m0 + xp) + (size_t)k * p.lda;
        const double *pb = B + (size_t)(n0 + xp) + (size_t)k * p.ldb;
        if (vec && m0 + xp + 1 < p.M) va = *reinterpret_cast<const double2_t *>(pa);
        else { if (m0 + xp < p.M) va.x = pa[0]; if (m0 + xp + 1 < p.M) va.y = pa[1]; }
        if (vec && n0 + xp + 1 < p.N) vb = *reinterpret_cast<const double2_t *>(pb);
        else { if (n0 + xp < p.N) vb.x = pb[0]; if (n0 + xp + 1 < p.N) vb.y = pb[1]; }
      }
      ra[i] = va; rb[i] = vb;
    }
  };
  auto put = [&]() {
#pragma unroll
    for (int i = 0; i < NF; ++i) {
      const int k = kq + 8 * i;
      *reinterpret_cast<double2_t *>(&sA[k * RK_LD + xp]) = ra[i];
      *reinterpret_cast<double2_t *>(&sB[k * RK_LD + xp]) = rb[i];
    }
  };
  double4_t acc[2][4];   // [ni][mi]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (double4_t){0.0, 0.0, 0.0, 0.0};
  fetch(0);
  // the C tile travels beside the first stage
  double cval[2][4][4];
  const double beta = p.beta;
  if (beta != 0.0) {
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        const int m = m0 + wm + mi * 16 + l15;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int n = n0 + wn + ni * 16 + l4 + 4 * r;
          cval[ni][mi][r] = (m < p.M && n < p.N) ? C[(size_t)m + (size_t)n * p.ldc] : 0.0;
        }
      }
  }
  for (int k0 = 0; k0 < p.K; k0 += RK) {
    __syncthreads();
    put();
    __syncthreads();
    if (k0 + RK < p.K) fetch(k0 + RK);
    const int kend = (p.K - k0 < RK) ? p.K - k0 : RK;
    for (int kk = 0; kk < kend; kk += 4) {
      double fa[4], fb[2];
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = sA[(kk + l4) * RK_LD + wm + i * 16 + l15];
#pragma unroll
      for (int i = 0; i < 2; ++i) fb[i] = sB[(kk + l4) * RK_LD + wn + i * 16 + l15];
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
          acc[ni][mi] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[ni], fa[mi], acc[ni][mi], 0, 0, 0);
    }
  }
  const double alpha = p.alpha;
#pragma unroll
  for (int ni = 0; ni < 2; ++ni)
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      const int m = m0 + wm + mi * 16 + l15;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + wn + ni * 16 + l4 + 4 * r;
        if (m < p.M && n < p.N) {
          double v = alpha * acc[ni][mi][r];
          if (beta != 0.0) v += beta * cval[ni][mi][r];
          C[(size_t)m + (size_t)n * p.ldc] = v;
        }
      }
    }
}

}  // namespace

void gemm(hipStream_t s, const GemmDesc &g) {
  if (g.M <= 0 || g.N <= 0 || g.batch <= 0) return;
  GemmArgs p;
  p.M = g.M; p.N = g.N; p.K = g.K > 0 ? g.K : 0;
  p.alpha = g.alpha; p.beta = g.beta;
  p.A = g.A; p.lda = g.lda; p.sA = g.strideA;
  p.B = g.B; p.ldb = g.ldb; p.sB = g.strideB;
  p.C = g.C; p.ldc = g.ldc; p.sC = g.strideC;
  p.lower_only = g.lower_only ? 1 : 0;
  p.offs = g.d_offs; p.dims = g.d_dims;
  // lower_only is defined on the 128x128 tiling (callers rely on whole diagonal tiles being
  // written), so the small-grid variant is used for plain products only
  const long long big_tiles = (long long)ceil_div(g.M, BM) * ceil_div(g.N, BN) * g.batch;
  // lower-only products of one batch entry: a grid of the tiles on and below the diagonal only
  const bool compact = g.lower_only && g.batch == 1 && !g.d_dims && !g.d_offs;
  // 16-byte operand fetch: every slab start must be 16-byte aligned -- even leading dimensions, aligned bases,
  // even batch strides, no per-batch offsets from device memory (EK_GEMM_VEC=0 turns the variants off)
  static int vec_env = -1;
  if (vec_env < 0) { const char *e = getenv("EK_GEMM_VEC"); vec_env = e ? atoi(e) : 1; }
  const bool vec = vec_env && (!g.d_offs || g.even_offs) && ((g.lda | g.ldb) & 1) == 0 &&
                   ((((size_t)g.A | (size_t)g.B) & 15) == 0) && (g.batch == 1 || ((g.strideA | g.strideB) & 1) == 0);
  if ((big_tiles < 256 && !g.lower_only) || (g.lower_only && g.small_tiles)) {
    p.tiles_m = ceil_div(g.M, SM); p.tiles_n = ceil_div(g.N, SN);
    dim3 sgrid(p.tiles_m * p.tiles_n, g.batch), sblock(256);
    if (compact && p.tiles_n <= p.tiles_m) {
      p.lower_only = 2;
      sgrid.x = (unsigned)((long long)p.tiles_n * p.tiles_m - (long long)p.tiles_n * (p.tiles_n - 1) / 2);
    }
    if (vec) {
      if (!g.transA && !g.transB) hipLaunchKernelGGL((gemm_small_kernel<false, false, true>), sgrid, sblock, 0, s, p);
      else if (!g.transA && g.transB) hipLaunchKernelGGL((gemm_small_kernel<false, true, true>), sgrid, sblock, 0, s, p);
      else if (g.transA && !g.transB) hipLaunchKernelGGL((gemm_small_kernel<true, false, true>), sgrid, sblock, 0, s, p);
      else hipLaunchKernelGGL((gemm_small_kernel<true, true, true>), sgrid, sblock, 0, s, p);
      return;
    }
    if (!g.transA && !g.transB) hipLaunchKernelGGL((gemm_small_kernel<false, false, false>), sgrid, sblock, 0, s, p);
    else if (!g.transA && g.transB) hipLaunchKernelGGL((gemm_small_kernel<false, true, false>), sgrid, sblock, 0, s, p);
    else if (g.transA && !g.transB) hipLaunchKernelGGL((gemm_small_kernel<true, false, false>), sgrid, sblock, 0, s, p);
    else hipLaunchKernelGGL((gemm_small_kernel<true, true, false>), sgrid, sblock, 0, s, p);
    return;
  }
  p.tiles_m = ceil_div(g.M, BM); p.tiles_n = ceil_div(g.N, BN);
  dim3 grid(p.tiles_m * p.tiles_n, g.batch), block(256);
  if (compact && p.tiles_n <= p.tiles_m) {
    p.lower_only = 2;
    grid.x = (unsigned)((long long)p.tiles_n * p.tiles_m - (long long)p.tiles_n * (p.tiles_n - 1) / 2);
  }
  // Rank-k updates (short K, C read-modify-write) run on the 8-wave variant: twice the resident
  // waves hide the operand and C latency that the 4-wave kernel exposes when there are only a few
  // k-steps per tile (measured: SYR2K of the tridiagonalisation -25 %); long-K products stay on
  // the 4-wave kernel (measured 3 % faster there).  EK_GEMM_W8=0/1 forces one of them.
  static int w8 = -2;
  if (w8 == -2) { const char *e = getenv("EK_GEMM_W8"); w8 = e ? atoi(e) : -1; }
  // rank-k updates with short K: the staged 8-wave kernel (EK_GEMM_RANKK=0 turns it off)
  static int rankk = -1;
  if (rankk < 0) {
    const char *e = getenv("EK_GEMM_RANKK"); rankk = e ? atoi(e) : 1;
    (void)hipFuncSetAttribute((const void *)gemm_rankk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                              2 * RK * RK_LD * (int)sizeof(double));
  }
  if (rankk && g.staged_rank_k && !g.transA && g.transB && g.K <= 256 && g.K >= 32 && w8 < 0) {
    hipLaunchKernelGGL(gemm_rankk_kernel, grid, dim3(512), 2 * RK * RK_LD * sizeof(double), s, p);
    return;
  }
  const bool use_w8 = (w8 >= 0) ? (w8 != 0) : (g.K <= 512 && g.beta != 0.0);
  if (use_w8) {
    dim3 b8(512);
    if (vec) {
      if (!g.transA && !g.transB) hipLaunchKernelGGL((gemm_kernel_w8<false, false, true>), grid, b8, 0, s, p);
      else if (!g.transA && g.transB) hipLaunchKernelGGL((gemm_kernel_w8<false, true, true>), grid, b8, 0, s, p);
      else if (g.transA && !g.transB) hipLaunchKernelGGL((gemm_kernel_w8<true, false, true>), grid, b8, 0, s, p);
      else hipLaunchKernelGGL((gemm_kernel_w8<true, true, true>), grid, b8, 0, s, p);
      return;
    }
    if (!g.transA && !g.transB) hipLaunchKernelGGL((gemm_kernel_w8<false, false, false>), grid, b8, 0, s, p);
    else if (!g.transA && g.transB) hipLaunchKernelGGL((gemm_kernel_w8<false, true, false>), grid, b8, 0, s, p);
    else if (g.transA && !g.transB) hipLaunchKernelGGL((gemm_kernel_w8<true, false, false>), grid, b8, 0, s, p);
    else hipLaunchKernelGGL((gemm_kernel_w8<true, true, false>), grid, b8, 0, s, p);
    return;
  }
  if (vec) {
    if (!g.transA && !g.transB) hipLaunchKernelGGL((gemm_kernel<false, false, true>), grid, block, 0, s, p);
    else if (!g.transA && g.transB) hipLaunchKernelGGL((gemm_kernel<false, true, true>), grid, block, 0, s, p);
    else if (g.transA && !g.transB) hipLaunchKernelGGL((gemm_kernel<true, false, true>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((gemm_kernel<true, true, true>), grid, block, 0, s, p);
    return;
  }
  if (!g.transA && !g.transB) hipLaunchKernelGGL((gemm_kernel<false, false, false>), grid, block, 0, s, p);
  else if (!g.transA && g.transB) hipLaunchKernelGGL((gemm_kernel<false, true, false>), grid, block, 0, s, p);
  else if (g.transA && !g.transB) hipLaunchKernelGGL((gemm_kernel<true, false, false>), grid, block, 0, s, p);
  else hipLaunchKernelGGL((gemm_kernel<true, true, false>), grid, block, 0, s, p);
}

}  // namespace ek
