// ek_sytrd.hip -- Householder tridiagonalisation A = Q T Q^T (lower), the largest stage
// of the path.  Replaces PDSYTRD('L') at solver_scalapack_all.f90:59 (one GPU = 1x1 grid).
//
// Algorithm: the LAPACK/ScaLAPACK blocked scheme (panel of NBP columns with deferred
// rank-2 updates, then one SYR2K trailing update on the matrix cores), restated for
// MI355X as a chain of exactly TWO kernels per column and one GEMM per panel:
//
//   colupd  (HBM-light)  finishes w of the previous column from the partial sums below,
//                        applies the deferred panel updates to the next column, and
//                        produces its Householder data (alpha, partial sum of squares);
//   symv    (HBM-bound)  y = A22 v with the lower triangle read exactly ONCE: the 128x128
//                        tiles are dealt strip-major in equal runs to ~2 workgroups per CU
//                        (every workgroup streams the same bytes); inside a tile lane l holds
//                        rows 2l,2l+1 of the block (16-byte coalesced loads down the
//                        column), accumulates the "row part" A v in registers and the
//                        "column part" A^T v in 32 per-lane accumulators that are reduced
//                        across the wave once per unit, not once per tile.  The reflector
//                        is formed on the fly from the unscaled column (v = x * scale), so
//                        no kernel boundary is spent on the scaling.  The same launch
//                        carries the small panel products V^T v, W^T v.
//
// All cross-workgroup sums go through partial buffers that are reduced in a fixed order
// (no atomics): results are bit-reproducible run to run.
//
// Algorithmic HBM traffic of symv: 8 bytes x (lower triangle of the active matrix) per
// column = 4 N^3 / 3 bytes in total; partial sums add ~2/128 of that.
#include "ek_common.h"

#include <cstdlib>
#include <vector>

namespace ek {
namespace {

constexpr int NBP = 64;      // panel width (128 measured equal: shorter SYR2K, longer column updates)
constexpr int TS = 128;      // symv strip width / row-block height
#ifndef EK_NDOT
#define EK_NDOT 8
#endif
constexpr int NDOT = EK_NDOT;   // reducer workgroups of the panel products inside a symv launch

struct SytrdBufs {
  double *xbuf;      // npad     unscaled current column (0 above the active part)
  double *P;         // npad x 3*NBP  panel image [V | W | V] (so [V|W] and [W|V] are both slices)
  double *ypart;     // NRB x npad   row-part partial sums per strip
  double *tpart;     // NRB x NRB x 128  column-part partial sums per (strip, piece)
  double *vavpart;   // one v^T A v partial sum per symv workgroup
  double *normpart;  // 2 x nch: one partial sum of squares per colupd workgroup, double-buffered
                     // by column parity (a launch reads the previous column's while writing its own)
  int nch;           // stride between the two normpart buffers
  double *dotpart;   // (colupd workgroups) x 2*NBP  partial sums of V^T x, W^T x
  double *dottot;    // 2*NBP  totals V^T v, W^T v of the current column (written by symv)
  double *scal;      // [c & 1] = alpha0 (first entry of the unscaled column) of column c
};

struct Refl { double beta, tau, scale; };

__device__ __forceinline__ double block_sum(double v, double *red /* >= 4 */) {
  // 256 threads = 4 waves; fixed order => deterministic
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// N sums at once (one barrier pair); red holds 4*N doubles
template <int N>
__device__ __forceinline__ void block_sum_n(double (&v)[N], double *red) {
#pragma unroll
  for (int i = 0; i < N; ++i)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v[i] += __shfl_down(v[i], o, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0)
#pragma unroll
    for (int i = 0; i < N; ++i) red[4 * i + (threadIdx.x >> 6)] = v[i];
  __syncthreads();
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = (red[4 * i] + red[4 * i + 1]) + (red[4 * i + 2] + red[4 * i + 3]);
}

// Householder data (DLARFG) from alpha = x_1 and the sum of squares of x_2..x_m.
__device__ __forceinline__ Refl reflector(double ssq, double alpha0) {
  Refl r;
  if (ssq == 0.0) { r.beta = alpha0; r.tau = 0.0; r.scale = 0.0; return r; }
  const double xnorm = sqrt(ssq);
  r.beta = -copysign(hypot(alpha0, xnorm), alpha0);
  r.tau = (r.beta - alpha0) / r.beta;
  r.scale = 1.0 / (alpha0 - r.beta);
  return r;
}

// ---- tile bookkeeping shared by symv (producer) and colupd (consumer of its partials) ----
// The active lower triangle is T strips of 128 columns; strip s (relative to the first active
// strip) owns the 128x128 tiles rb = s .. T-1.  Tiles are numbered strip-major and dealt to
// the workgroups in equal contiguous runs of q tiles, so every workgroup streams the same
// number of bytes; a strip therefore reaches colupd as one partial per workgroup ("piece").
// With P > 1 (distributed run, one member of a 1 x P column-block-cyclic layout with 128-wide
// blocks) a member owns every P-th strip: its k-th active strip has T - k*P tiles, T being the
// length of its first active strip, and the same formulas hold with the stride P.
__device__ __host__ __forceinline__ int tile_start(int s, int T, int P = 1) { return s * T - P * ((s * (s - 1)) / 2); }
__device__ __forceinline__ int strip_of_tile(int L, int T, int P = 1) {
  const double b = 2.0 * T + (double)P;
  const double disc = b * b - 8.0 * (double)P * (double)L;
  int s = (int)((b - sqrt(disc > 0.0 ? disc : 0.0)) / (2.0 * P));
  const int smax = (T + P - 1) / P - 1;
  if (s < 0) s = 0;
  if (s > smax) s = smax;
  while (s + 1 <= smax && tile_start(s + 1, T, P) <= L) ++s;
  while (s > 0 && tile_start(s, T, P) > L) --s;
  return s;
}
__device__ __forceinline__ int first_piece(int s, int T, int q, int P = 1) { return tile_start(s, T, P) / q; }
__device__ __forceinline__ int num_pieces(int s, int T, int q, int P = 1) {
  return (tile_start(s + 1, T, P) - 1) / q - tile_start(s, T, P) / q + 1;
}

// The exchange window of a distributed run: one local array that the team's all-reduce sums in place (the same sum on
// every rank, so everything computed from it is bit-identical across the team).  (Until round 4 a second, "peer" form
// stored every rank's contribution straight into every other rank's HBM; it never ran on two devices and went with the
// one-stage form's retirement from the whole-path call.)
struct XWin {
  const double *slot[1];          // the summed window
  int nslots;
  __device__ __forceinline__ double get(size_t i) const { return slot[0][i]; }
};
struct XDst {
  double *slot[1];                // this rank's window
  int nslots;
  __device__ __forceinline__ void put(size_t i, double v) const { slot[0][i] = v; }
  __device__ __forceinline__ void announce() const {}
};

struct ColupdArgs {
  int n, npad, lda, ldv;
  double *A, *V;          // V: explicit reflector matrix (may be null)
  double *d, *e, *tau;
  SytrdBufs b;
  int r0;                 // first row handled by workgroup 0 (multiple of CR)
  // finalize part (previous column)
  int finalize;           // 0/1
  int jp, ip;             // global / in-panel index of the column whose w is finished
  int S0p, NRB;           // symv launch geometry of that column
  int qp;                 // tiles per symv workgroup
  int nwg_p;              // symv workgroups (vavpart entries)
  int nchunks_p;          // colupd workgroups that produced normpart for column jp
  // update part (next column)
  int update;             // 0/1
  int j;                  // global index of the column to update
  int i_new;              // finished panel columns once this launch is done (dots needed for them)
  // distributed run: the all-reduced exchange window of this column (rows r0 .. npad-1):
  // [0, xcnt) raw entries of column j, [xcnt, 2 xcnt) y = A22 x, [2 xcnt] x^T A x
  XWin xw;
  int xcnt;
};

constexpr int CR = 32;    // rows per colupd workgroup
constexpr int NSL = 256 / CR;   // slices per row (threads sharing one row's sums)

// DIST: the symv partial sums and the raw column come from the exchange window (every member of
// the team runs this kernel redundantly on identical data), not from this member's own buffers.
template <bool DIST>
__global__ __launch_bounds__(256) void colupd_kernel(ColupdArgs p) {
  __shared__ double s_pvw[2 * NBP];          // [0,NBP): V^T v totals, [NBP,2NBP): W^T v totals
  __shared__ double s_Vj[NBP], s_Wj[NBP];
  __shared__ double s_red[8];
  __shared__ double s_acc[3][NSL][CR];
  __shared__ double s_x[CR], s_vn[CR], s_wn[CR];   // new column x, newest panel column (v, w)
  const int t = threadIdx.x, lane = t % CR, q = t / CR;
  const int r = p.r0 + blockIdx.x * CR + lane;
  const int ldp = p.npad;
  double *__restrict__ Pv = p.b.P;                          // V block
  double *__restrict__ Pw = p.b.P + (size_t)NBP * ldp;      // W block
  double *__restrict__ Pv2 = p.b.P + (size_t)2 * NBP * ldp; // V copy
  const double *s_pv = s_pvw, *s_pw = s_pvw + NBP;

  double accB = 0.0, a_old = 0.0;
  if (p.finalize) {
    // symv ran on the UNSCALED column x (x_j = alpha0 included): with v = scale*x + corr*e_j,
    // corr = 1 - scale*alpha0, everything it produced is corrected here by linearity:
    //   A v = scale*(A x) + corr*A(:,j),  V^T v = scale*(V^T x) + corr*V(j,:),
    //   v^T A v = scale^2 x^T A x + 2 scale corr (A x)_j + corr^2 A(j,j).
    const int ip = p.ip, jp = p.jp, j = jp + 1;
    const int T = p.NRB - p.S0p;
    const int rbj = j / TS;
    double red3[3] = {0.0, 0.0, 0.0};   // ssq, x^T A x, (A x)_j
    const double *normp = p.b.normpart + (size_t)(jp & 1) * p.b.nch;
    for (int c = t; c < p.nchunks_p; c += 256) red3[0] += normp[c];
    if (DIST) {
      if (t == 0) { red3[1] = p.xw.get(2 * (size_t)p.xcnt); red3[2] = p.xw.get((size_t)p.xcnt + (j - p.r0)); }
    } else {
      for (int u = t; u < p.nwg_p; u += 256) red3[1] += p.b.vavpart[u];
      const int nS = rbj - p.S0p + 1;
      for (int idx = t; idx < nS; idx += 256) red3[2] += p.b.ypart[(size_t)(p.S0p + idx) * p.npad + j];
      const int np = num_pieces(rbj - p.S0p, T, p.qp);
      for (int idx = t; idx < np; idx += 256)
        red3[2] += p.b.tpart[((size_t)rbj * p.NRB + idx) * TS + (j % TS)];
    }
    double vjk = 0.0, wjk = 0.0, dtot = 0.0;
    if (t < 2 * NBP && (t & (NBP - 1)) < ip) {
      const int k = t & (NBP - 1);
      vjk = Pv[(size_t)j + (size_t)k * ldp]; wjk = Pw[(size_t)j + (size_t)k * ldp];
      dtot = p.b.dottot[t];
    }
    const double alpha0 = p.b.scal[jp & 1];
    // still the panel-start value: A(:,j) is not written until column j is finalized
    const double ajj = DIST ? p.xw.get(j - p.r0) : p.A[(size_t)j + (size_t)j * p.lda];
    if (r >= j && r < p.n) a_old = DIST ? p.xw.get(r - p.r0) : p.A[(size_t)r + (size_t)j * p.lda];
    block_sum_n<3>(red3, s_red);
    const Refl rf = reflector(red3[0], alpha0);
    const double corr = 1.0 - rf.scale * alpha0;
    if (t < 2 * NBP) {
      const int k = t & (NBP - 1);
      double pvw = 0.0;
      if (k < ip) pvw = rf.scale * dtot + corr * (t < NBP ? vjk : wjk);
      s_pvw[t] = pvw;
      if (t < NBP) { s_Vj[k] = vjk; } else { s_Wj[k] = wjk; }
    }
    __syncthreads();
    double red2[2] = {0.0, 0.0};   // sum_k pv*pw,  sum_k (Vj pw + Wj pv)
    if (t < ip) { red2[0] = s_pv[t] * s_pw[t]; red2[1] = s_Vj[t] * s_pw[t] + s_Wj[t] * s_pv[t]; }
    block_sum_n<2>(red2, s_red);
    const double vav = rf.scale * rf.scale * red3[1] + 2.0 * rf.scale * corr * red3[2] + corr * corr * ajj;
    const double wv = rf.tau * (vav - 2.0 * red2[0]);
    const double alpha = -0.5 * rf.tau * wv;
    const double yj = rf.scale * red3[2] + corr * ajj;
    const double wj = rf.tau * (yj - red2[1]) + alpha;   // v_j = 1
    // per-row sums, 4 slices per row for memory-level parallelism
    double y = 0.0, accA = 0.0, aB = 0.0;
    if (r >= j && r < p.npad) {
      const int rb = r / TS;
      if (DIST) {
        if (q == 0) y = p.xw.get((size_t)p.xcnt + (r - p.r0));
      } else {
        double y0 = 0.0, y1 = 0.0, y2 = 0.0, y3 = 0.0;
        int S = p.S0p + q;
        const double *yp = p.b.ypart + r;
        const size_t stn = (size_t)NSL * p.npad;
        for (; S + 3 * NSL <= rb; S += 4 * NSL) {   // 4 loads in flight per thread
          const double *b0 = yp + (size_t)S * p.npad;
          const double u0 = b0[0], u1 = b0[stn], u2 = b0[2 * stn], u3 = b0[3 * stn];
          y0 += u0; y1 += u1; y2 += u2; y3 += u3;
        }
        for (; S <= rb; S += NSL) y0 += yp[(size_t)S * p.npad];
        const int np = num_pieces(rb - p.S0p, T, p.qp);
        for (int pc = q; pc < np; pc += NSL) y1 += p.b.tpart[((size_t)rb * p.NRB + pc) * TS + (r % TS)];
        y = (y0 + y1) + (y2 + y3);
      }
      {
        double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
        int k = q;
        const double *pvr = Pv + r, *pwr = Pw + r;
        for (; k + 3 * NSL < ip; k += 4 * NSL) {     // 8 loads in flight per thread
          const double v0 = pvr[(size_t)k * ldp], w0 = pwr[(size_t)k * ldp];
          const double v1 = pvr[(size_t)(k + NSL) * ldp], w1 = pwr[(size_t)(k + NSL) * ldp];
          const double v2 = pvr[(size_t)(k + 2 * NSL) * ldp], w2 = pwr[(size_t)(k + 2 * NSL) * ldp];
          const double v3 = pvr[(size_t)(k + 3 * NSL) * ldp], w3 = pwr[(size_t)(k + 3 * NSL) * ldp];
          a0 += v0 * s_pw[k] + w0 * s_pv[k];                         b0 += v0 * s_Wj[k] + w0 * s_Vj[k];
          a1 += v1 * s_pw[k + NSL] + w1 * s_pv[k + NSL];             b1 += v1 * s_Wj[k + NSL] + w1 * s_Vj[k + NSL];
          a0 += v2 * s_pw[k + 2 * NSL] + w2 * s_pv[k + 2 * NSL];     b0 += v2 * s_Wj[k + 2 * NSL] + w2 * s_Vj[k + 2 * NSL];
          a1 += v3 * s_pw[k + 3 * NSL] + w3 * s_pv[k + 3 * NSL];     b1 += v3 * s_Wj[k + 3 * NSL] + w3 * s_Vj[k + 3 * NSL];
        }
        for (; k < ip; k += NSL) {
          const double v0 = pvr[(size_t)k * ldp], w0 = pwr[(size_t)k * ldp];
          a0 += v0 * s_pw[k] + w0 * s_pv[k]; b0 += v0 * s_Wj[k] + w0 * s_Vj[k];
        }
        accA = a0 + a1; aB = b0 + b1;
      }
    }
    s_acc[0][q][lane] = y; s_acc[1][q][lane] = accA; s_acc[2][q][lane] = aB;
    __syncthreads();
    if (q == 0) {
      if (r >= j && r < p.npad) {
        y = 0.0; accA = 0.0; accB = 0.0;
#pragma unroll
        for (int u = 0; u < NSL; ++u) { y += s_acc[0][u][lane]; accA += s_acc[1][u][lane]; accB += s_acc[2][u][lane]; }
        double v_r = (r == j) ? 1.0 : p.b.xbuf[r] * rf.scale;
        if (r >= p.n) v_r = 0.0;
        const double y_r = rf.scale * y + corr * a_old;
        double w_r = (r == j) ? wj : rf.tau * (y_r - accA) + alpha * v_r;
        if (r >= p.n) w_r = 0.0;
        Pv[(size_t)r + (size_t)ip * ldp] = v_r;
        Pv2[(size_t)r + (size_t)ip * ldp] = v_r;
        Pw[(size_t)r + (size_t)ip * ldp] = w_r;
        if (r < p.n) {
          p.A[(size_t)r + (size_t)jp * p.lda] = (r == j) ? rf.beta : v_r;
          if (p.V) p.V[(size_t)r + (size_t)jp * p.ldv] = v_r;
        }
        if (r == j) { p.e[jp] = rf.beta; p.tau[jp] = rf.tau; }
        accB += v_r * wj + w_r;   // k = ip term: V(j,ip) = 1, W(j,ip) = wj
        s_vn[lane] = v_r; s_wn[lane] = w_r;
      } else if (r < j && r < p.npad) {
        Pv[(size_t)r + (size_t)ip * ldp] = 0.0;
        Pv2[(size_t)r + (size_t)ip * ldp] = 0.0;
        Pw[(size_t)r + (size_t)ip * ldp] = 0.0;
      }
    }
  }
  if (!p.update) return;

  const int j = p.j;
  double sq = 0.0, xr = 0.0;
  if (q == 0 && r >= j && r < p.n) {
    if (!p.finalize) a_old = DIST ? p.xw.get(r - p.r0) : p.A[(size_t)r + (size_t)j * p.lda];
    const double a = a_old - accB;   // the updated column lives in xbuf / d only
    if (r == j) { p.d[j] = a; p.b.xbuf[r] = 0.0; }
    else {
      p.b.xbuf[r] = a;
      xr = a;
      if (r == j + 1) p.b.scal[j & 1] = a;
      else sq = a * a;
    }
  }
  if (q == 0) {
    s_x[lane] = xr;
    if (!(p.finalize && r >= j && r < p.npad)) { s_vn[lane] = 0.0; s_wn[lane] = 0.0; }
  }
  const double tot = block_sum(sq, s_red);   // its barriers also publish s_x / s_vn / s_wn
  if (t == 0) p.b.normpart[(size_t)(j & 1) * p.b.nch + blockIdx.x] = tot;
  // partial panel products V^T x and W^T x over this workgroup's 64 rows: thread <-> panel
  // column, no cross-lane reduction; the newest column comes from LDS (not yet visible in P)
  if (t < 2 * NBP) {
    const int k = t & (NBP - 1), inew = p.i_new;
    double acc = 0.0;
    if (k < inew) {
      const int rbase = p.r0 + blockIdx.x * CR;
      if (p.finalize && k == p.ip) {
        const double *src = (t < NBP) ? s_vn : s_wn;
        for (int l = 0; l < CR; ++l) acc += src[l] * s_x[l];
      } else {
        // rbase is a multiple of 64 and npad of 128: the 64 rows are in bounds and 16-B aligned
        const double2 *col = reinterpret_cast<const double2 *>(((t < NBP) ? Pv : Pw) + (size_t)k * ldp + rbase);
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
        for (int l = 0; l < CR / 2; l += 4) {
          const double2 c0 = col[l], c1 = col[l + 1], c2 = col[l + 2], c3 = col[l + 3];
          a0 += c0.x * s_x[2 * l] + c0.y * s_x[2 * l + 1];
          a1 += c1.x * s_x[2 * l + 2] + c1.y * s_x[2 * l + 3];
          a2 += c2.x * s_x[2 * l + 4] + c2.y * s_x[2 * l + 5];
          a3 += c3.x * s_x[2 * l + 6] + c3.y * s_x[2 * l + 7];
        }
        acc = (a0 + a1) + (a2 + a3);
      }
    }
    p.b.dotpart[(size_t)blockIdx.x * 2 * NBP + t] = acc;
  }
}

// ------------------------------------------------------------------------------ symv
struct SymvArgs {
  int n, npad, lda;
  const double *A;
  SytrdBufs b;
  int j;            // column whose reflector is applied; active rows/cols > j
  int i;            // in-panel index (number of finished panel columns)
  int S0, NRB;      // first active strip (of this member), total row blocks
  int P;            // strip stride: 1, or the team size of a distributed run (strips S0, S0+P, ...)
  int q, nwg;       // tiles per workgroup, symv workgroups
  int ntiles;
  int ndot;         // reducer workgroups (blockIdx < ndot) that total the panel products: NDOT or 0
  int nchunks;      // colupd workgroups that produced normpart / dotpart
};

typedef double d2_t __attribute__((ext_vector_type(2)));

// 8 columns x (2 rows per lane) of a tile: the unit of the software pipeline (a quarter of a
// wave's 32-column share of a tile; two such buffers are alive at a time)
constexpr int QC = 8;
__device__ __forceinline__ void part_load(d2_t (&a)[QC], const double *__restrict__ Acol, int lda) {
#pragma unroll
  for (int cc = 0; cc < QC; ++cc)
    a[cc] = __builtin_nontemporal_load(reinterpret_cast<const d2_t *>(Acol + (size_t)cc * lda));
}

template <bool DIAG>
__device__ __forceinline__ void part_fma(const d2_t (&a)[QC], int h, int rloc0, int cloc0,
                                         const double *__restrict__ svc, double vr0, double vr1,
                                         double &y0, double &y1, double (&tc)[32]) {
#pragma unroll
  for (int cc = 0; cc < QC; ++cc) {
    const int c = h * QC + cc;
    const double ax = a[cc].x, ay = a[cc].y;
    const double vc = svc[c];
    if (DIAG) {
      const int cl = cloc0 + c;   // row part uses r >= c, column part r > c
      const double rx = (rloc0 >= cl) ? ax : 0.0, ry = (rloc0 + 1 >= cl) ? ay : 0.0;
      const double cx = (rloc0 > cl) ? ax : 0.0, cy = (rloc0 + 1 > cl) ? ay : 0.0;
      y0 += rx * vc; y1 += ry * vc;
      tc[c] += cx * vr0 + cy * vr1;
    } else {
      y0 += ax * vc; y1 += ay * vc;
      tc[c] += ax * vr0 + ay * vr1;
    }
  }
}

template <bool DIST>
__global__ __launch_bounds__(256, 2) void symv_kernel(SymvArgs p) {
  __shared__ double s_vc[TS];            // v on the strip's columns
  __shared__ double s_y[2][4][TS];       // per-wave row-part partials, double buffered
  __shared__ double s_red[8];
  __shared__ double s_dot[2][2 * NBP];
  __shared__ double s_t[4][16 * 65];     // per-wave transpose buffer for the column-part flush
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const double *__restrict__ xbuf = p.b.xbuf;

  if ((int)blockIdx.x < p.ndot) {
    // ---- reducers: totals of the raw panel products V^T x, W^T x (colupd applies the scaling).
    // NDOT workgroups, each owning 2*NBP/NDOT of the 2*NBP columns (so no cross-workgroup sum):
    // thread = (column, one of GRP chunk groups), 4 loads in flight, fixed-order LDS finish.
    constexpr int CPW = 2 * NBP / NDOT;              // columns per reducer workgroup
    constexpr int GRP = 256 / CPW;                   // chunk groups
    const int col = t % CPW, grp = t / CPW;
    const int k2 = (int)blockIdx.x * CPW + col;
    const double *dp = p.b.dotpart + k2;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    int c = grp;
    for (; c + 3 * GRP < p.nchunks; c += 4 * GRP) {
      a0 += dp[(size_t)c * 2 * NBP];
      a1 += dp[(size_t)(c + GRP) * 2 * NBP];
      a2 += dp[(size_t)(c + 2 * GRP) * 2 * NBP];
      a3 += dp[(size_t)(c + 3 * GRP) * 2 * NBP];
    }
    for (; c < p.nchunks; c += GRP) a0 += dp[(size_t)c * 2 * NBP];
    double *sd = &s_dot[0][0];                       // 2 * 2 * NBP >= 256 doubles
    sd[grp * CPW + col] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (t < CPW) {
      double tot = 0.0;
#pragma unroll
      for (int g = 0; g < GRP; ++g) tot += sd[g * CPW + t];
      const int kk = (int)blockIdx.x * CPW + t;
      p.b.dottot[kk] = ((kk & (NBP - 1)) < p.i) ? tot : 0.0;
    }
    return;
  }

  // ---- symv workgroup: a run of q tiles in strip-major order, software-pipelined by halves
  const int w = blockIdx.x - p.ndot;
  const int T = p.NRB - p.S0;
  const int P = DIST ? p.P : 1;
  const int L0 = w * p.q;
  int L1 = L0 + p.q; if (L1 > p.ntiles) L1 = p.ntiles;
  double vav = 0.0;
  if (L0 >= L1) {   // (cannot happen with nwg = ceil(ntiles / q); kept for safety)
    if (t == 0) p.b.vavpart[w] = 0.0;
    return;
  }
  int s = strip_of_tile(L0, T, P);
  int S = p.S0 + s * P;                          // absolute strip, absolute row block
  int rb = S + (L0 - tile_start(s, T, P));
  const size_t lda = (size_t)p.lda;
  auto tile_ptr = [&](int SS, int RB) {
    return p.A + (size_t)(RB * TS + 2 * lane) + (size_t)(SS * TS + wave * 32) * lda;
  };
  d2_t bufA[QC], bufB[QC];
  const double *cur = tile_ptr(S, rb);
  part_load(bufA, cur, p.lda);
  double tc[32];
  int buf = 0;
  bool new_strip = true;
  const double *svc = s_vc + wave * 32;
  for (int L = L0; L < L1; ++L) {
    part_load(bufB, cur + QC * lda, p.lda);
    if (new_strip) {
      if (t < TS) s_vc[t] = xbuf[S * TS + t];   // x is 0 outside the active range j1 .. n-1
#pragma unroll
      for (int c = 0; c < 32; ++c) tc[c] = 0.0;
      __syncthreads();
      new_strip = false;
    }
    const int row = rb * TS + 2 * lane;
    const double2 x = *reinterpret_cast<const double2 *>(xbuf + row);
    const double vr0 = x.x, vr1 = x.y;
    double y0 = 0.0, y1 = 0.0;
    const bool diag = (rb == S);
    if (diag) part_fma<true>(bufA, 0, 2 * lane, wave * 32, svc, vr0, vr1, y0, y1, tc);
    else part_fma<false>(bufA, 0, 0, 0, svc, vr0, vr1, y0, y1, tc);
    part_load(bufA, cur + 2 * QC * lda, p.lda);
    if (diag) part_fma<true>(bufB, 1, 2 * lane, wave * 32, svc, vr0, vr1, y0, y1, tc);
    else part_fma<false>(bufB, 1, 0, 0, svc, vr0, vr1, y0, y1, tc);
    part_load(bufB, cur + 3 * QC * lda, p.lda);
    if (diag) part_fma<true>(bufA, 2, 2 * lane, wave * 32, svc, vr0, vr1, y0, y1, tc);
    else part_fma<false>(bufA, 2, 0, 0, svc, vr0, vr1, y0, y1, tc);
    // next tile (possibly in the next strip): issue its first part before finishing this one
    int s2 = s, S2 = S, rb2 = rb + 1;
    if (rb2 == p.NRB) { ++s2; S2 += P; rb2 = S2; }
    const bool strip_ends = (s2 != s) || (L + 1 == L1);
    if (L + 1 < L1) { cur = tile_ptr(S2, rb2); part_load(bufA, cur, p.lda); }
    if (diag) part_fma<true>(bufB, 3, 2 * lane, wave * 32, svc, vr0, vr1, y0, y1, tc);
    else part_fma<false>(bufB, 3, 0, 0, svc, vr0, vr1, y0, y1, tc);
    *reinterpret_cast<double2 *>(&s_y[buf][wave][2 * lane]) = make_double2(y0, y1);
    __syncthreads();
    if (t < TS) {
      const double ys = (s_y[buf][0][t] + s_y[buf][1][t]) + (s_y[buf][2][t] + s_y[buf][3][t]);
      const int rr = rb * TS + t;
      p.b.ypart[(size_t)S * p.npad + rr] = ys;
      vav += xbuf[rr] * ys;
    }
    buf ^= 1;
    if (strip_ends) {
      // flush the strip's column part: transpose through LDS (16 columns per round), then
      // lanes 0..15 each add up one column's 64 lane-partials in a fixed order
      const int piece = w - first_piece(s, T, p.q, P);
      double *tp = p.b.tpart + ((size_t)S * p.NRB + piece) * TS + wave * 32;
      double *st = s_t[wave];
#pragma unroll
      for (int round = 0; round < 2; ++round) {
#pragma unroll
        for (int c = 0; c < 16; ++c) st[c * 65 + lane] = tc[round * 16 + c];
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): this wave's LDS writes have landed
        {
          // all 64 lanes: 4 lanes per column, 16 lane-partials each, then two butterfly steps
          const int cidx = lane >> 2, part = lane & 3;
          const double *src = st + cidx * 65 + part * 16;
          double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
          for (int l = 0; l < 16; l += 4) { a0 += src[l]; a1 += src[l + 1]; a2 += src[l + 2]; a3 += src[l + 3]; }
          double tot = (a0 + a1) + (a2 + a3);
          tot += __shfl_xor(tot, 1, 64);
          tot += __shfl_xor(tot, 2, 64);
          if (part == 0) {
            tp[round * 16 + cidx] = tot;
            vav += svc[round * 16 + cidx] * tot;
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
      new_strip = true;
      __syncthreads();   // s_vc is about to be rewritten
    }
    s = s2; S = S2; rb = rb2;
  }
  const double tot = block_sum(vav, s_red);
  if (t == 0) p.b.vavpart[w] = tot;
}

// ------------------------------------------------------------------------------ distributed run
// One member of a team of P ranks (1 x P process grid, column-block-cyclic with 128-wide blocks:
// member r owns the strips S = r, r+P, ...).  Every member holds a full-size A of which only its
// own strips are kept current by the trailing updates, runs symv over its own strips, and the team
// all-reduces one small window per column:
//   [0, cnt)      raw entries of the next column (contributed by its owner, zeros elsewhere),
//   [cnt, 2 cnt)  this member's part of y = A22 x,
//   [2 cnt]       its part of x^T A x,
// cnt = npad - r0 rows from the colupd row base r0 of the next column.  colupd<true> then runs
// redundantly on identical data on every member, so the panel, d, e, tau and the reflector
// columns stay bit-identical across the team with no further exchange.
struct YredArgs {
  int n, npad, lda, NRB;
  const double *A;
  SytrdBufs b;
  int r0, cnt;
  int jn;            // column whose raw entries travel (the next one to be updated), -1: none
  int own_next;      // this member owns the strip of column jn
  int with_y;        // 0 at a panel start: only the column part is filled
  int S0, P, T, q, nwg;   // the member's symv geometry of this column (nwg = 0: it had no tiles)
  XDst xd;
};

constexpr int YR = 64;   // rows per yreduce workgroup (4 slices per row)

__global__ __launch_bounds__(256) void yreduce_kernel(YredArgs p) {
  __shared__ double s_acc[4][YR];
  __shared__ double s_red[8];
  const int t = threadIdx.x, lane = t & (YR - 1), q4 = t / YR;
  const int r = p.r0 + blockIdx.x * YR + lane;
  if (q4 == 0 && r < p.npad) {
    double a = 0.0;
    if (p.jn >= 0 && p.own_next && r >= p.jn && r < p.n) a = p.A[(size_t)r + (size_t)p.jn * p.lda];
    p.xd.put(r - p.r0, a);
  }
  if (!p.with_y) { p.xd.announce(); return; }
  double y = 0.0;
  if (p.nwg > 0 && r < p.npad && r >= p.jn) {
    const int rb = r / TS;
    if (rb >= p.S0) {
      const int kmax = (rb - p.S0) / p.P;
      double y0 = 0.0, y1 = 0.0;
      int k = q4;
      for (; k + 4 <= kmax; k += 8) {
        y0 += p.b.ypart[(size_t)(p.S0 + k * p.P) * p.npad + r];
        y1 += p.b.ypart[(size_t)(p.S0 + (k + 4) * p.P) * p.npad + r];
      }
      if (k <= kmax) y0 += p.b.ypart[(size_t)(p.S0 + k * p.P) * p.npad + r];
      if ((rb - p.S0) % p.P == 0) {
        const int np = num_pieces(kmax, p.T, p.q, p.P);
        for (int pc = q4; pc < np; pc += 4) y1 += p.b.tpart[((size_t)rb * p.NRB + pc) * TS + (r % TS)];
      }
      y = y0 + y1;
    }
  }
  s_acc[q4][lane] = y;
  __syncthreads();
  if (q4 == 0 && r < p.npad)
    p.xd.put((size_t)p.cnt + (r - p.r0), (s_acc[0][lane] + s_acc[1][lane]) + (s_acc[2][lane] + s_acc[3][lane]));
  if (blockIdx.x == 0) {
    double v = 0.0;
    for (int u = t; u < p.nwg; u += 256) v += p.b.vavpart[u];
    const double tot = block_sum(v, s_red);
    if (t == 0) p.xd.put(2 * (size_t)p.cnt, tot);
  }
  p.xd.announce();
}

// Rehearsal exchange for a team held by ONE process on one GPU: sum of the members' windows in
// member order, written back to all of them (what ncclAllReduce does across processes).
struct TeamBufs { double *buf[kMaxTeam]; int nmem; };
__global__ void team_allreduce_kernel(TeamBufs tb, size_t count) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  double v = 0.0;
  for (int m = 0; m < tb.nmem; ++m) v += tb.buf[m][i];
  for (int m = 0; m < tb.nmem; ++m) tb.buf[m][i] = v;
}

// Operand table of the trailing update of one member: per panel, one GEMM per owned strip that
// still has active columns (A22(:, strip) -= [V|W] [W|V](strip rows)^T, lower part).
// Entry (panel, k): offsets {A, B, C} and dims {M, N, K} for the batched GEMM.
__global__ void syr2k_table_kernel(int n, int npad, int lda, int P, int rank, int maxb,
                                   long long *offs, int *dims) {
  const int panel = blockIdx.x, k = threadIdx.x;
  if (k >= maxb) return;
  const int j0 = panel * NBP;
  const int pw = (n - 1 - j0 < NBP) ? n - 1 - j0 : NBP;
  const int r2 = j0 + pw;
  const int Sf = r2 / TS;
  const int Sfl = Sf + ((rank - Sf) % P + P) % P;
  const int S = Sfl + k * P;
  long long c0 = (long long)S * TS;
  if (c0 < r2) c0 = r2;
  int M = 0, N = 0;
  if (c0 < n) {
    M = n - (int)c0;
    long long cend = (long long)(S + 1) * TS;
    if (cend > n) cend = n;
    N = (int)(cend - c0);
  }
  const size_t e = (size_t)panel * maxb + k;
  offs[3 * e] = c0; offs[3 * e + 1] = c0; offs[3 * e + 2] = c0 * ((long long)lda + 1);
  dims[3 * e] = M; dims[3 * e + 1] = N; dims[3 * e + 2] = (pw == NBP) ? 2 * NBP : pw;
}

// PDSYTRD leaves d on the diagonal and e on the sub-diagonal of A
__global__ void put_diag_kernel(int n, double *A, int lda, const double *__restrict__ d) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n) A[(size_t)j + (size_t)j * lda] = d[j];
}

inline size_t al256(size_t b) { return (b + 255) & ~(size_t)255; }

// Optional instrumentation for bench.py's roofline line: HIP events around every stride-th symv
// launch (by column index, so the sample is uniform over the trailing orders) on the launch
// stream.  Off by default: a pair of event records costs ~5 us of host time per launch.
struct SymvProfile {
  bool enabled = false;
  int stride = 1;
  std::vector<hipEvent_t> ev;
  size_t used = 0;
  double bytes = 0.0;
  long long launches = 0;
  double seconds = 0.0;
} g_prof;

// bring-up / tuning knobs (environment), read once
struct Knobs {
  int max_cols = -1;   // EK_SYTRD_MAXCOLS: stop after this many columns (micro-benchmarks only)
  int G = 0;           // EK_SYMV_G: minimum tiles per symv workgroup (0 = none)
  int wgs = 0;         // EK_SYMV_WGS: symv workgroups to aim for (0 = 512)
  Knobs() {
    if (const char *e = getenv("EK_SYMV_WGS")) wgs = atoi(e);
    if (const char *e = getenv("EK_SYTRD_MAXCOLS")) max_cols = atoi(e);
    if (const char *e = getenv("EK_SYMV_G")) G = atoi(e);
  }
};
Knobs &knobs_rw() { static Knobs k; return k; }
const Knobs &knobs() { return knobs_rw(); }

struct Layout {
  int npad, NRB, nch;
  size_t off_x, off_P, off_y, off_t, off_vav, off_norm, off_dot, off_dtot, off_scal, total;
  explicit Layout(int n) {
    npad = round_up(n > 0 ? n : 1, TS); NRB = npad / TS; nch = npad / CR + 2;
    size_t o = 0;
    off_x = o; o += al256((size_t)npad * 8);
    off_P = o; o += al256((size_t)npad * 3 * NBP * 8);
    off_y = o; o += al256((size_t)NRB * npad * 8);
    off_t = o; o += al256((size_t)NRB * NRB * TS * 8);
    off_vav = o; o += al256((size_t)(NRB * NRB + 4096) * 8);
    off_norm = o; o += al256((size_t)2 * nch * 8);
    off_dot = o; o += al256((size_t)nch * 2 * NBP * 8);
    off_dtot = o; o += al256((size_t)2 * NBP * 8);
    off_scal = o; o += 256;
    total = o;
  }
};

}  // namespace

size_t sytrd_work_bytes(int n) { return Layout(n).total; }

// placement experiments (tools/placement_split.py): sub-buffers selected by mask (1 x, 2 panel,
// 4 row-part sums, 8 column-part sums, 16 the small rest) are taken from `alt` instead of `work`
static void *g_split_alt = nullptr;
static int g_split_mask = 0;
void sytrd_debug_split(void *alt, int mask) { g_split_alt = alt; g_split_mask = mask; }

void sytrd_lower(hipStream_t s, int n, double *A, int lda, double *d, double *e, double *tau,
                 double *V, int ldv, void *work) {
  if (n <= 0) return;
  const Layout L(n);
  char *w = (char *)work;
  SytrdBufs b;
  b.xbuf = (double *)(w + L.off_x); b.P = (double *)(w + L.off_P);
  b.ypart = (double *)(w + L.off_y); b.tpart = (double *)(w + L.off_t);
  b.vavpart = (double *)(w + L.off_vav); b.normpart = (double *)(w + L.off_norm); b.nch = L.nch;
  b.dotpart = (double *)(w + L.off_dot); b.dottot = (double *)(w + L.off_dtot);
  b.scal = (double *)(w + L.off_scal);
  (void)hipMemsetAsync(work, 0, L.total, s);
  if (g_split_alt) {   // placement experiments: selected sub-buffers live in a second scratch block
    char *a = (char *)g_split_alt;
    (void)hipMemsetAsync(a, 0, L.total, s);
    if (g_split_mask & 1) b.xbuf = (double *)(a + L.off_x);
    if (g_split_mask & 2) b.P = (double *)(a + L.off_P);
    if (g_split_mask & 4) b.ypart = (double *)(a + L.off_y);
    if (g_split_mask & 8) b.tpart = (double *)(a + L.off_t);
    if (g_split_mask & 16) {
      b.vavpart = (double *)(a + L.off_vav); b.normpart = (double *)(a + L.off_norm);
      b.dotpart = (double *)(a + L.off_dot); b.dottot = (double *)(a + L.off_dtot); b.scal = (double *)(a + L.off_scal);
    }
  }
  const int npad = L.npad, NRB = L.NRB;

  ColupdArgs c{};
  c.n = n; c.npad = npad; c.lda = lda; c.ldv = ldv; c.A = A; c.V = V; c.d = d; c.e = e; c.tau = tau;
  c.b = b; c.NRB = NRB;
  SymvArgs sv{};
  sv.n = n; sv.npad = npad; sv.lda = lda; sv.A = A; sv.b = b; sv.NRB = NRB; sv.P = 1;

  auto launch_colupd = [&](int row_from) {
    c.r0 = (row_from / CR) * CR;
    const int nblk = ceil_div(npad - c.r0, CR);
    hipLaunchKernelGGL(colupd_kernel<false>, dim3(nblk), dim3(256), 0, s, c);
    return nblk;
  };
  const int wg_target = knobs().wgs > 0 ? knobs().wgs : 512;   // 2 resident workgroups per CU

  int nchunks_cur = 0;
  for (int j0 = 0; j0 < n - 1; j0 += NBP) {
    if (knobs().max_cols >= 0 && j0 >= knobs().max_cols) break;
    const int pw = (n - 1 - j0 < NBP) ? n - 1 - j0 : NBP;   // reflector columns j0 .. j0+pw-1
    // first column of the panel: nothing deferred yet
    c.finalize = 0; c.update = 1; c.j = j0; c.i_new = 0;
    nchunks_cur = launch_colupd(j0);
    for (int i = 0; i < pw; ++i) {
      const int j = j0 + i;
      // y = A22 v and the panel products
      sv.j = j; sv.i = i; sv.S0 = (j + 1) / TS;
      const int T = NRB - sv.S0;
      sv.ntiles = T * (T + 1) / 2;
      // ~2 resident workgroups per CU for large trailing matrices; one per CU once the matrix is
      // small enough that per-workgroup fixed costs dominate (measured: T <= 40 strips)
      const int target = (knobs().wgs > 0) ? wg_target : (T <= 40 ? 256 : wg_target);
      sv.q = ceil_div(sv.ntiles, target);
      if (knobs().G > 0 && sv.q < knobs().G) sv.q = knobs().G;
      sv.nwg = ceil_div(sv.ntiles, sv.q);
      sv.ndot = (i > 0) ? NDOT : 0;
      sv.nchunks = nchunks_cur;
      hipEvent_t e0 = nullptr, e1 = nullptr;
      const bool timed = g_prof.enabled && (j % g_prof.stride == 0);
      if (timed) {
        if (g_prof.used + 2 > g_prof.ev.size()) {
          const size_t old = g_prof.ev.size();
          g_prof.ev.resize(old + 4096);
          for (size_t q = old; q < g_prof.ev.size(); ++q) (void)hipEventCreate(&g_prof.ev[q]);
        }
        e0 = g_prof.ev[g_prof.used++]; e1 = g_prof.ev[g_prof.used++];
        (void)hipEventRecord(e0, s);
      }
      hipLaunchKernelGGL(symv_kernel<false>, dim3(sv.nwg + sv.ndot), dim3(256), 0, s, sv);
      if (timed) {
        (void)hipEventRecord(e1, s);
        const double m = (double)(n - j - 1);
        g_prof.bytes += 8.0 * m * (m + 1.0) * 0.5;   // lower triangle of the active matrix, once
        g_prof.launches += 1;
      }
      // finish w_j, then update column j+1 (unless the panel ends here)
      c.finalize = 1; c.jp = j; c.ip = i; c.S0p = sv.S0; c.qp = sv.q; c.nwg_p = sv.nwg;
      c.nchunks_p = nchunks_cur;
      c.update = (i + 1 < pw) ? 1 : 0; c.j = j + 1; c.i_new = i + 1;
      const int nb = launch_colupd(j + 1);
      if (c.update) nchunks_cur = nb;
    }
    // trailing update A22 -= V W^T + W V^T = [V|W] [W|V]^T, lower triangle, on the matrix cores
    const int r2 = j0 + pw, m2 = n - r2;
    if (m2 > 0) {
      const double *P1 = b.P + r2, *P2 = b.P + (size_t)NBP * npad + r2;
      double *A22 = A + (size_t)r2 + (size_t)r2 * lda;
      if (pw == NBP) {
        gemm(s, false, true, m2, m2, 2 * NBP, -1.0, P1, npad, P2, npad, 1.0, A22, lda, true);
      } else {   // short last panel: columns >= pw of the image are stale
        gemm(s, false, true, m2, m2, pw, -1.0, P1, npad, P2, npad, 1.0, A22, lda, true);
        gemm(s, false, true, m2, m2, pw, -1.0, P2, npad, P1, npad, 1.0, A22, lda, true);
      }
    }
  }
  // last diagonal entry
  c.finalize = 0; c.update = 1; c.j = n - 1; c.i_new = 0;
  launch_colupd(n - 1);
  hipLaunchKernelGGL(put_diag_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, s, n, A, lda, d);
}

// ------------------------------------------------------------------------------ distributed host side
namespace {
struct DistLayout {
  Layout base;
  int maxb, npanels;
  size_t off_xch, off_offs, off_dims, total;
  DistLayout(int n, int P) : base(n) {
    maxb = ceil_div(base.NRB, P) + 1;
    npanels = ceil_div(n > 1 ? n - 1 : 1, NBP);
    size_t o = base.total;
    off_xch = o; o += al256((size_t)(2 * base.npad + 8) * 8);
    off_offs = o; o += al256((size_t)npanels * maxb * 3 * sizeof(long long));
    off_dims = o; o += al256((size_t)npanels * maxb * 3 * sizeof(int));
    total = o;
  }
};
}  // namespace

size_t sytrd_dist_work_bytes(int n, int nranks) { return DistLayout(n, nranks > 0 ? nranks : 1).total; }

void sytrd_team_allreduce(hipStream_t s, int nmem, double *const *bufs, size_t count, void *) {
  if (nmem <= 1 || count == 0) return;
  TeamBufs tb{};
  tb.nmem = nmem;
  for (int m = 0; m < nmem; ++m) tb.buf[m] = bufs[m];
  hipLaunchKernelGGL(team_allreduce_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, tb, count);
}

void team_allgatherv(hipStream_t s, int nmem, int rank0, double *const *bufs, const size_t *offs,
                     const size_t *counts, int nranks, void *) {
  (void)rank0;
  for (int r = 0; r < nranks && r < nmem; ++r)
    for (int m = 0; m < nmem; ++m)
      if (m != r && counts[r] > 0)
        (void)hipMemcpyAsync(bufs[m] + offs[r], bufs[r] + offs[r], counts[r] * sizeof(double),
                             hipMemcpyDeviceToDevice, s);
}

// Same column loop as sytrd_lower; every phase is issued for each member held by this process
// (one in production, the whole team in the single-GPU rehearsal), with one exchange per column.
void sytrd_lower_dist(hipStream_t s, int n, int nmem, const SytrdMember *mem, const SytrdExchange &x) {
  if (n <= 0 || nmem <= 0 || nmem > kMaxTeam) return;
  const int P = x.nranks;
  const DistLayout L(n, P);
  const int npad = L.base.npad, NRB = L.base.NRB;

  struct MemberState {
    SytrdBufs b; double *xch; long long *offs; int *dims;
    ColupdArgs c; SymvArgs sv; YredArgs yr;
  };
  std::vector<MemberState> st(nmem);
  double *xbufs[kMaxTeam];
  for (int m = 0; m < nmem; ++m) {
    char *w = (char *)mem[m].work;
    MemberState &M = st[m];
    SytrdBufs &b = M.b;
    b.xbuf = (double *)(w + L.base.off_x); b.P = (double *)(w + L.base.off_P);
    b.ypart = (double *)(w + L.base.off_y); b.tpart = (double *)(w + L.base.off_t);
    b.vavpart = (double *)(w + L.base.off_vav); b.normpart = (double *)(w + L.base.off_norm); b.nch = L.base.nch;
    b.dotpart = (double *)(w + L.base.off_dot); b.dottot = (double *)(w + L.base.off_dtot);
    b.scal = (double *)(w + L.base.off_scal);
    M.xch = (double *)(w + L.off_xch); M.offs = (long long *)(w + L.off_offs); M.dims = (int *)(w + L.off_dims);
    xbufs[m] = M.xch;
    (void)hipMemsetAsync(w, 0, L.total, s);
    hipLaunchKernelGGL(syr2k_table_kernel, dim3(L.npanels), dim3(round_up(L.maxb, 64)), 0, s, n, npad,
                       mem[m].lda, P, mem[m].rank, L.maxb, M.offs, M.dims);
    ColupdArgs &c = M.c; c = ColupdArgs{};
    c.n = n; c.npad = npad; c.lda = mem[m].lda; c.ldv = mem[m].ldv; c.A = mem[m].A; c.V = mem[m].V;
    c.d = mem[m].d; c.e = mem[m].e; c.tau = mem[m].tau; c.b = b; c.NRB = NRB;
    c.xw.slot[0] = M.xch; c.xw.nslots = 1;
    SymvArgs &sv = M.sv; sv = SymvArgs{};
    sv.n = n; sv.npad = npad; sv.lda = mem[m].lda; sv.A = mem[m].A; sv.b = b; sv.NRB = NRB; sv.P = P;
    YredArgs &yr = M.yr; yr = YredArgs{};
    yr.n = n; yr.npad = npad; yr.lda = mem[m].lda; yr.NRB = NRB; yr.A = mem[m].A; yr.b = b; yr.P = P;
    yr.xd.slot[0] = M.xch; yr.xd.nslots = 1;
  }
  // exchange of the window of column jn (with or without the symv sums), then colupd on every member
  auto exchange = [&](int jn, bool with_y) {
    const int r0 = (jn / CR) * CR, cnt = npad - r0;
    for (int m = 0; m < nmem; ++m) {
      YredArgs &yr = st[m].yr;
      yr.r0 = r0; yr.cnt = cnt; yr.jn = jn; yr.with_y = with_y ? 1 : 0;
      yr.own_next = ((jn / TS) % P == mem[m].rank) ? 1 : 0;
      hipLaunchKernelGGL(yreduce_kernel, dim3(ceil_div(cnt, YR)), dim3(256), 0, s, yr);
    }
    x.allreduce(s, nmem, xbufs, with_y ? 2 * (size_t)cnt + 1 : (size_t)cnt, x.user);
    return cnt;
  };
  auto launch_colupd = [&](int row_from, int cnt) {
    const int r0 = (row_from / CR) * CR;
    const int nblk = ceil_div(npad - r0, CR);
    for (int m = 0; m < nmem; ++m) {
      st[m].c.r0 = r0; st[m].c.xcnt = cnt;
      hipLaunchKernelGGL(colupd_kernel<true>, dim3(nblk), dim3(256), 0, s, st[m].c);
    }
    return nblk;
  };

  int nchunks_cur = 0;
  for (int j0 = 0; j0 < n - 1; j0 += NBP) {
    if (knobs().max_cols >= 0 && j0 >= knobs().max_cols) break;
    const int pw = (n - 1 - j0 < NBP) ? n - 1 - j0 : NBP;
    // first column of the panel: its raw entries from the owner of the strip
    for (int m = 0; m < nmem; ++m) { ColupdArgs &c = st[m].c; c.finalize = 0; c.update = 1; c.j = j0; c.i_new = 0; }
    nchunks_cur = launch_colupd(j0, exchange(j0, false));
    for (int i = 0; i < pw; ++i) {
      const int j = j0 + i;
      const int S0 = (j + 1) / TS;
      for (int m = 0; m < nmem; ++m) {
        SymvArgs &sv = st[m].sv;
        const int rank = mem[m].rank;
        const int S0l = S0 + ((rank - S0) % P + P) % P;      // first active strip of this member
        const int T = NRB - S0l;                              // its length in tiles (<= 0: none)
        sv.j = j; sv.i = i; sv.S0 = S0l;
        sv.ntiles = (T > 0) ? tile_start(ceil_div(T, P), T, P) : 0;
        if (sv.ntiles > 0) {
          // a member streams 1/P of the triangle per launch, so per-workgroup fixed costs weigh
          // more than on one GPU: one workgroup per CU (measured, team of 8, s per rank: N=16384
          // 0.470 / 0.460 / 0.474 / 0.472 at 192 / 256 / 384 / 512; N=32768 1.70 / 1.72 at 256 / 512)
          const int target = (knobs().wgs > 0) ? knobs().wgs : 256;
          sv.q = ceil_div(sv.ntiles, target);
          if (knobs().G > 0 && sv.q < knobs().G) sv.q = knobs().G;
          sv.nwg = ceil_div(sv.ntiles, sv.q);
        } else { sv.q = 1; sv.nwg = 0; }
        sv.ndot = (i > 0) ? NDOT : 0;
        sv.nchunks = nchunks_cur;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        const bool timed = g_prof.enabled && m == 0 && (j % g_prof.stride == 0) && sv.nwg > 0;
        if (timed) {
          if (g_prof.used + 2 > g_prof.ev.size()) {
            const size_t old = g_prof.ev.size();
            g_prof.ev.resize(old + 4096);
            for (size_t q = old; q < g_prof.ev.size(); ++q) (void)hipEventCreate(&g_prof.ev[q]);
          }
          e0 = g_prof.ev[g_prof.used++]; e1 = g_prof.ev[g_prof.used++];
          (void)hipEventRecord(e0, s);
        }
        if (sv.nwg + sv.ndot > 0)
          hipLaunchKernelGGL(symv_kernel<true>, dim3(sv.nwg + sv.ndot), dim3(256), 0, s, sv);
        if (timed) {
          (void)hipEventRecord(e1, s);
          // this member's share of the lower triangle of the active matrix (by tiles)
          const double mm = (double)(n - j - 1);
          const int Tall = NRB - S0;
          g_prof.bytes += 8.0 * mm * (mm + 1.0) * 0.5 * (double)sv.ntiles / (double)(Tall * (Tall + 1) / 2);
          g_prof.launches += 1;
        }
        YredArgs &yr = st[m].yr;
        yr.S0 = S0l; yr.T = T > 0 ? T : 0; yr.q = sv.q; yr.nwg = sv.nwg;
        ColupdArgs &c = st[m].c;
        c.finalize = 1; c.jp = j; c.ip = i; c.S0p = S0l; c.qp = sv.q; c.nwg_p = sv.nwg;
        c.nchunks_p = nchunks_cur;
        c.update = (i + 1 < pw) ? 1 : 0; c.j = j + 1; c.i_new = i + 1;
      }
      const int nb = launch_colupd(j + 1, exchange(j + 1, true));
      if (i + 1 < pw) nchunks_cur = nb;
    }
    // trailing update of the strips each member owns: one batched GEMM (a GEMM per strip)
    const int r2 = j0 + pw;
    if (n - r2 > 0) {
      const int Sf = r2 / TS, panel = j0 / NBP;
      for (int m = 0; m < nmem; ++m) {
        const int Sfl = Sf + ((mem[m].rank - Sf) % P + P) % P;
        if (Sfl >= NRB) continue;
        const int nb = ceil_div(NRB - Sfl, P);
        long long c0 = (long long)Sfl * TS; if (c0 < r2) c0 = r2;
        if (c0 >= n) continue;
        const double *P1 = st[m].b.P, *P2 = st[m].b.P + (size_t)NBP * npad;
        GemmDesc g{};
        g.M = n - (int)c0; g.N = TS; g.K = (pw == NBP) ? 2 * NBP : pw;
        g.transA = false; g.transB = true; g.alpha = -1.0; g.beta = 1.0;
        g.A = P1; g.lda = npad; g.strideA = 0; g.B = P2; g.ldb = npad; g.strideB = 0;
        g.C = mem[m].A; g.ldc = mem[m].lda; g.strideC = 0; g.batch = nb; g.lower_only = true;
        g.d_offs = st[m].offs + (size_t)panel * L.maxb * 3; g.d_dims = st[m].dims + (size_t)panel * L.maxb * 3;
        gemm(s, g);
        if (pw != NBP) { g.A = P2; g.B = P1; gemm(s, g); }   // short last panel: columns >= pw of the image are stale
      }
    }
  }
  // last diagonal entry
  for (int m = 0; m < nmem; ++m) { ColupdArgs &c = st[m].c; c.finalize = 0; c.update = 1; c.j = n - 1; c.i_new = 0; }
  launch_colupd(n - 1, exchange(n - 1, false));
  for (int m = 0; m < nmem; ++m)
    hipLaunchKernelGGL(put_diag_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, s, n, mem[m].A, mem[m].lda, mem[m].d);
}

void sytrd_set_max_cols(int max_cols) { knobs_rw().max_cols = max_cols; }   // tuning hooks only (-1 = all)
int sytrd_get_max_cols() { return knobs().max_cols; }

void symv_profile_enable(int stride) {
  g_prof.enabled = stride > 0;
  g_prof.stride = stride > 0 ? stride : 1;
  g_prof.used = 0; g_prof.bytes = 0.0; g_prof.launches = 0; g_prof.seconds = 0.0;
}

// Call after the stream has been synchronised. Accumulates and resets the event pool.
void symv_profile_collect(double *seconds, long long *launches, double *bytes) {
  for (size_t q = 0; q + 1 < g_prof.used; q += 2) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, g_prof.ev[q], g_prof.ev[q + 1]) == hipSuccess) g_prof.seconds += ms * 1e-3;
  }
  g_prof.used = 0;
  if (seconds) *seconds = g_prof.seconds;
  if (launches) *launches = g_prof.launches;
  if (bytes) *bytes = g_prof.bytes;
}

}  // namespace ek
