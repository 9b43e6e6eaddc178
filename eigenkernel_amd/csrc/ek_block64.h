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
constexpr int kScratch = 16 * LD;   // doubles of LDS scratch the factorisations below ask for

// C = op(A) op(B), all 64x64 LDS images, on the matrix cores, by the 4 waves of the workgroup:
// wave w owns rows 16 w .. 16 w + 15 of C; acc[jt][r] is C(16 w + l4 + 4 r, 16 jt + l15).
__device__ __forceinline__ void mm64_acc(const double *sA, bool ta, const double *sB, bool tb, double4_t (&acc)[4]) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l15 = lane & 15, l4 = lane >> 4;
  const int i0 = 16 * wave;
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
}
__device__ __forceinline__ void mm64_store(const double4_t (&acc)[4], double *sC) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l15 = lane & 15, l4 = lane >> 4;
#pragma unroll
  for (int jt = 0; jt < 4; ++jt)
#pragma unroll
    for (int r = 0; r < 4; ++r) sC[(16 * wave + l4 + 4 * r) * LD + 16 * jt + l15] = acc[jt][r];
}
// Callers synchronise before and after.
// out_g != nullptr: C goes to global memory (column-major, ld 64), row i scaled by rs[i] if rs.
__device__ __forceinline__ void mm64(const double *sA, bool ta, const double *sB, bool tb, double *sC,
                                     double *out_g = nullptr, const double *rs = nullptr) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l15 = lane & 15, l4 = lane >> 4;
  const int i0 = 16 * wave;
  double4_t acc[4];
  mm64_acc(sA, ta, sB, tb, acc);
  if (!out_g) { mm64_store(acc, sC); return; }
#pragma unroll
  for (int jt = 0; jt < 4; ++jt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = i0 + l4 + 4 * r, j = 16 * jt + l15;
      out_g[i + SB * j] = rs ? rs[i] * acc[jt][r] : acc[jt][r];
    }
}

// The factorisations below are the serial part of a panel.  They work on 64x64 LDS images in row blocks
// of 16.  The serial part of a block runs INSIDE ONE WAVE with the 16 x 64 row block in registers
// (lane c holds column c, 16 registers): a step takes its pivot and its multipliers from other lanes with
// v_readlane (they are wave-uniform, so they sit in scalar registers and feed the FMAs directly) -- no LDS
// traffic and no barrier inside a block.  The rest of the image is then updated with the block on the matrix
// cores by all four waves (16x16x4 f64 tiles), two barriers per block, eight per factorisation.
// The earlier forms (whole matrix in the registers of 256 threads, one row and one barrier per step) took
// 630 - 1260 cycles per step: 50 000 cycles for the Cholesky factor, 81 000 for the LU.
__device__ __forceinline__ double fast_rcp(double p) {     // |p| well inside the normal range
  double r = __builtin_amdgcn_rcp(p);
  double e = __builtin_fma(-p, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-p, r, 1.0);
  return __builtin_fma(r, e, r);
}

// 1/sqrt(d) and sqrt(d) for d > 0 well inside the normal range: v_rsq_f64 and two coupled Newton steps
__device__ __forceinline__ void fast_rsqrt(double d, double &rinv, double &root) {
  const double r = __builtin_amdgcn_rsq(d);
  double g = d * r, h = 0.5 * r;
  double e = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, e, g); h = __builtin_fma(h, e, h);
  e = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, e, g); h = __builtin_fma(h, e, h);
  e = __builtin_fma(-g, g, d);                 // one correction of the root itself
  root = __builtin_fma(e, h, g);
  rinv = h + h;
}

__device__ __forceinline__ double lane_get(double v, int lane) {     // lane: wave-uniform
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

// C(ci.., cj..) -= A(ci.., ka..) B(kb.., cj..) for one 16x16 tile of an image, inner dimension 16, by one wave.
// A(i, k) = TA ? sA[(ka + k) LD + i] : sA[i LD + ka + k];  B(k, j) = sB[(kb + k) LD + j].
template <bool TA>
__device__ __forceinline__ void tile_sub16(double *sC, int ci, int cj, const double *sA, int ka, const double *sB, int kb) {
  const int lane = threadIdx.x & 63, l15 = lane & 15, l4 = lane >> 4;
  double4_t acc;
#pragma unroll
  for (int r = 0; r < 4; ++r) acc[r] = sC[(ci + l4 + 4 * r) * LD + cj + l15];
#pragma unroll
  for (int kk = 0; kk < 16; kk += 4) {
    const double x = TA ? sA[(ka + kk + l4) * LD + ci + l15] : sA[(ci + l15) * LD + ka + kk + l4];
    const double y = sB[(kb + kk + l4) * LD + cj + l15];
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-x, y, acc, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) sC[(ci + l4 + 4 * r) * LD + cj + l15] = acc[r];
}

// G = R^T R, in place: image -> upper factor (zeros below; only the upper triangle of G is used).
// Returns (in all threads) the index of the first pivot that is not positive (or outside 1e-290 .. 1e290, or
// not a number), -1 if there is none.  scr: kScratch doubles.
// A step of the in-wave part: the current row goes to an LDS line, its pivot and the 15 - j multipliers come
// back as broadcast reads (a v_readlane pair per multiplier costs about 40 cycles), the row is updated with
// multipliers  g_i / d; the 16 rows are scaled by 1 / sqrt(d) after the 16 steps (one rsqrt per lane).
__device__ __forceinline__ int chol64_upper_wg(double *sG, double *scr) {
  const int t = threadIdx.x, c = t & 63, w = t >> 6;
  int fail = -1;
#pragma unroll 1
  for (int jb = 0; jb < 4; ++jb) {
    const int j0 = 16 * jb;
    if (w == 0) {
      double a[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) a[i] = sG[(j0 + i) * LD + c];
      double mydiag = 1.0;                        // the pivot of this lane's column
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int jg = j0 + j;
        double *line = scr + (j & 1) * 64;
        const double g = a[j];
        line[c] = g;
        double d = lane_get(g, jg);               // two v_readlane: sooner here than the LDS round trip
        double m[16];
#pragma unroll
        for (int i = j + 1; i < 16; ++i) m[i] = line[j0 + i];
        if (!(d > 1e-290) || !(d < 1e290)) { fail = (fail < 0) ? jg : fail; d = 1.0; }
        mydiag = (c == jg) ? d : mydiag;
        const double wv = g * fast_rcp(d);
#pragma unroll
        for (int i = j + 1; i < 16; ++i) a[i] -= m[i] * wv;
      }
      // rows by 1 / sqrt(pivot): one rsqrt per lane, the 16 factors back through the line
      double rinv, root;
      fast_rsqrt(mydiag, rinv, root);
      scr[c] = rinv;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const double ri = scr[j0 + i];
        a[i] = (c > j0 + i) ? a[i] * ri : (c == j0 + i ? root : 0.0);
        sG[(j0 + i) * LD + c] = a[i];
      }
    }
    __syncthreads();
    const int nt = 3 - jb, r0 = j0 + 16;          // tiles on and above the diagonal of the rest
    int idx = 0;
    for (int I = 0; I < nt; ++I)
      for (int J = I; J < nt; ++J, ++idx)
        if ((idx & 3) == w) tile_sub16<true>(sG, r0 + 16 * I, r0 + 16 * J, sG, j0, sG, j0);
    __syncthreads();
  }
  int *s_int = (int *)(scr + 128);
  if (t == 0) *s_int = fail;
  __syncthreads();
  fail = *s_int;
  __syncthreads();
  return fail;
}

// T Y = C in place (sY: C on entry, Y on exit) for a triangular 64x64 image sT: UPPER (only its upper
// triangle is read) or lower (only the strictly lower part is read when UNIT).  Stage 1: wave w inverts the
// diagonal 16x16 block w inside its registers (a column per lane) into sInv (16 x LD doubles of scratch).
// Stage 2: wave w owns the 16 columns 16 w .. of Y as four accumulator tiles and runs the block substitution
// on the matrix cores without meeting the other waves: an accumulator tile is fed back as the B operand with
// the inner index permuted to the accumulator's row order (k = l4 + 4 r), so Y never leaves the registers.
// The caller's writes to sY need no barrier of their own before the call (the one after stage 1 covers them).
template <bool UPPER, bool UNIT>
__device__ __forceinline__ void trisolve64_wg(const double *sT, double *sY, double *sInv) {
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, l15 = lane & 15, l4 = lane >> 4;
  {
    const int j0 = 16 * w;
    double x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = (i == l15) ? 1.0 : 0.0;
    double rd = 1.0;
    if (!UNIT) rd = fast_rcp(sT[(j0 + l15) * LD + j0 + l15]);
#pragma unroll
    for (int st = 0; st < 16; ++st) {
      const int k = UPPER ? 15 - st : st;
      if (!UNIT) x[k] *= lane_get(rd, k);
#pragma unroll
      for (int i = 0; i < 16; ++i)
        if (UPPER ? i < k : i > k) x[i] -= sT[(j0 + i) * LD + j0 + k] * x[k];      // a broadcast read
    }
    if (lane < 16) {
#pragma unroll
      for (int i = 0; i < 16; ++i) sInv[i * LD + j0 + l15] = x[i];
    }
  }
  double xt[4][4][4];               // the off-diagonal blocks of T as A operands: they do not wait for stage 1
#pragma unroll
  for (int jb = 0; jb < 4; ++jb)
#pragma unroll
    for (int I = 0; I < 4; ++I)
      if (UPPER ? I < jb : I > jb) {
#pragma unroll
        for (int r = 0; r < 4; ++r) xt[jb][I][r] = -sT[(16 * I + l15) * LD + 16 * jb + l4 + 4 * r];
      }
  __syncthreads();
  double4_t Y[4];
  double xi[4][4];
  const int cj = 16 * w + l15;
#pragma unroll
  for (int I = 0; I < 4; ++I)
#pragma unroll
    for (int r = 0; r < 4; ++r) { Y[I][r] = sY[(16 * I + l4 + 4 * r) * LD + cj]; xi[I][r] = sInv[l15 * LD + 16 * I + l4 + 4 * r]; }
#pragma unroll
  for (int bi = 0; bi < 4; ++bi) {
    const int jb = UPPER ? 3 - bi : bi;
    double4_t acc = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xi[jb][r], Y[jb][r], acc, 0, 0, 0);
    Y[jb] = acc;
#pragma unroll
    for (int I = 0; I < 4; ++I)
      if (UPPER ? I < jb : I > jb) {
#pragma unroll
        for (int r = 0; r < 4; ++r) Y[I] = __builtin_amdgcn_mfma_f64_16x16x4f64(xt[jb][I][r], Y[jb][r], Y[I], 0, 0, 0);
      }
  }
#pragma unroll
  for (int I = 0; I < 4; ++I)
#pragma unroll
    for (int r = 0; r < 4; ++r) sY[(16 * I + l4 + 4 * r) * LD + cj] = Y[I][r];
  __syncthreads();
}

// X = R^-1 for upper triangular R (only its upper triangle is read); every entry of sX is written.
__device__ __forceinline__ void triinv64_upper_wg(const double *sR, double *sX, double *sInv) {
  const int t = threadIdx.x;
  for (int idx = t; idx < SB * SB; idx += 256) sX[(idx >> 6) * LD + (idx & 63)] = ((idx >> 6) == (idx & 63)) ? 1.0 : 0.0;
  trisolve64_wg<true, false>(sR, sX, sInv);
}

// LU of (A - S) without pivoting, S(j,j) = -sign(pivot) so that |pivot| >= 1; in place (strictly
// lower = L, upper = U), signs to s_sign.  scr: kScratch doubles.  Wave 0 carries the 16 x 64 row block
// (a: lane = column) AND the 64 x 16 column block (b: lane = row) of the current block through the same 16
// steps; a step sends row j and column j through two LDS lines and reads the pivot and the 2 (15 - j)
// multipliers back as broadcasts.
__device__ __forceinline__ void lu64_signed_wg(double *sD, double *s_sign, double *scr) {
  const int t = threadIdx.x, c = t & 63, w = t >> 6;
#pragma unroll 1
  for (int jb = 0; jb < 4; ++jb) {
    const int j0 = 16 * jb;
    if (w == 0) {
      double a[16], b[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) { a[i] = sD[(j0 + i) * LD + c]; b[i] = sD[c * LD + j0 + i]; }
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int jg = j0 + j;
        double *rowl = scr + (j & 1) * 128, *coll = rowl + 64;
        rowl[c] = a[j];                          // A(jg, c)
        coll[c] = b[j];                          // A(c, jg), not yet divided by the pivot
        double piv = lane_get(a[j], jg);
        double mu[16], ml[16];
#pragma unroll
        for (int i = j + 1; i < 16; ++i) { mu[i] = rowl[j0 + i]; ml[i] = coll[j0 + i]; }
        const double sj = (piv >= 0.0) ? -1.0 : 1.0;
        piv -= sj;
        const double pinv = fast_rcp(piv);
        if (c == jg) { a[j] = piv; s_sign[jg] = sj; }
        const double us = (c > jg) ? a[j] * pinv : 0.0;        // U(jg, c) / pivot; rows of lanes <= jg are final
        const double bs = (c > jg) ? b[j] * pinv : 0.0;        // L(c, jg)
        b[j] = bs;
#pragma unroll
        for (int i = j + 1; i < 16; ++i) { a[i] -= ml[i] * us; b[i] -= bs * mu[i]; }
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (c >= j0 + i) sD[(j0 + i) * LD + c] = a[i];          // U
        if (c > j0 + i) sD[c * LD + j0 + i] = b[i];             // L, rows inside the block and below it
      }
    }
    __syncthreads();
    const int nt = 3 - jb, r0 = j0 + 16;
    for (int idx = w; idx < nt * nt; idx += 4)
      tile_sub16<false>(sD, r0 + 16 * (idx / nt), r0 + 16 * (idx % nt), sD, j0, sD, j0);
    __syncthreads();
  }
}

// Y = L^-1 C for the UNIT lower triangular L held strictly below the diagonal of sL, with
// C(r, c) = -U(c, r) sign(r) for c <= r (U = upper part of sL): Y(j, i) = T(i, j) of T = -U S L^-T.
__device__ __forceinline__ void tsolve64_wg(const double *sL, const double *s_sign, double *sY, double *sInv) {
  const int t = threadIdx.x;
  for (int idx = t; idx < SB * SB; idx += 256) {
    const int r = idx >> 6, c = idx & 63;
    sY[r * LD + c] = (c <= r) ? -sL[c * LD + r] * s_sign[r] : 0.0;
  }
  trisolve64_wg<false, true>(sL, sY, sInv);
}

}  // namespace b64
}  // namespace ek
