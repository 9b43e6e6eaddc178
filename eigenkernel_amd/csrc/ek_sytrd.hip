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
//   symv    (HBM-bound)  y = A22 v with the lower triangle read exactly ONCE: a workgroup
//                        owns a 128-column strip x a run of 128-row blocks; lane l holds
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

#include <vector>

namespace ek {
namespace {

constexpr int NBP = 64;      // panel width
constexpr int CH = 256;      // rows per colupd workgroup
constexpr int TS = 128;      // symv strip width / row-block height

struct SytrdBufs {
  double *xbuf;      // npad     unscaled current column (0 above the active part)
  double *P;         // npad x 3*NBP  panel image [V | W | V] (so [V|W] and [W|V] are both slices)
  double *ypart;     // NRB x npad   row-part partial sums per strip
  double *tpart;     // NRB x NRB x 128  column-part partial sums per (strip, segment)
  double *vavpart;   // NRB*NRB   v^T A v partial sums per unit
  double *normpart;  // npad/CH + 1
  double *dotpart;   // (npad/CH + 1) x 2*NBP  panel-dot partial sums
  double *scal;      // [0] = alpha0 of the current column
};

struct Refl { double beta, tau, scale; };

// Householder data of the current column from its partial sums; every workgroup of both
// kernels evaluates this identically (same inputs, same order).
__device__ __forceinline__ Refl reflector(const double *__restrict__ normpart, int nchunks,
                                          double alpha0) {
  double ssq = 0.0;
  for (int c = 0; c < nchunks; ++c) ssq += normpart[c];
  Refl r;
  if (ssq == 0.0) { r.beta = alpha0; r.tau = 0.0; r.scale = 0.0; return r; }
  const double xnorm = sqrt(ssq);
  r.beta = -copysign(hypot(alpha0, xnorm), alpha0);
  r.tau = (r.beta - alpha0) / r.beta;
  r.scale = 1.0 / (alpha0 - r.beta);
  return r;
}

__device__ __forceinline__ double block_sum(double v, double *red /* >= 4 */) {
  // 256 threads = 4 waves; fixed order => deterministic
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

struct ColupdArgs {
  int n, npad, lda, ldv;
  double *A, *V;          // V: explicit reflector matrix (may be null)
  double *d, *e, *tau;
  SytrdBufs b;
  int r0;                 // first row handled by workgroup 0 (multiple of CH)
  // finalize part (previous column)
  int finalize;           // 0/1
  int jp, ip;             // global / in-panel index of the column whose w is finished
  int S0p, Gp, NRB;       // symv launch geometry of that column
  int nchunks_p;          // chunks that produced normpart for column jp
  int ndot_p;             // dot blocks of that symv launch
  int nunits_p;           // units of that symv launch
  // update part (next column)
  int update;             // 0/1
  int j, i;               // global / in-panel index of the column to update (i terms)
};

__global__ __launch_bounds__(CH) void colupd_kernel(ColupdArgs p) {
  __shared__ double s_pv[NBP], s_pw[NBP], s_Vj[NBP], s_Wj[NBP];
  __shared__ double s_red[8];
  const int t = threadIdx.x;
  const int r = p.r0 + blockIdx.x * CH + t;
  const int ldp = p.npad;
  double *__restrict__ Pv = p.b.P;                          // V block
  double *__restrict__ Pw = p.b.P + (size_t)NBP * ldp;      // W block
  double *__restrict__ Pv2 = p.b.P + (size_t)2 * NBP * ldp; // V copy

  double v_r = 0.0, w_r = 0.0, accB = 0.0;
  double wj = 0.0;
  Refl rf{0.0, 0.0, 0.0};
  if (p.finalize) {
    const int ip = p.ip, jp = p.jp;
    rf = reflector(p.b.normpart, p.nchunks_p, p.b.scal[0]);
    // totals of the panel products
    if (t < ip) {
      double a = 0.0;
      for (int c = 0; c < p.ndot_p; ++c) a += p.b.dotpart[(size_t)c * 2 * NBP + t];
      s_pv[t] = a;
    } else if (t >= NBP && t < NBP + ip) {
      const int k = t - NBP;
      double a = 0.0;
      for (int c = 0; c < p.ndot_p; ++c) a += p.b.dotpart[(size_t)c * 2 * NBP + NBP + k];
      s_pw[k] = a;
    }
    // v^T A v
    double part = 0.0;
    for (int u = t; u < p.nunits_p; u += CH) part += p.b.vavpart[u];
    const double vav = block_sum(part, s_red);   // also publishes s_pv / s_pw
    double dvw = 0.0;
    for (int k = 0; k < ip; ++k) dvw += s_pv[k] * s_pw[k];
    const double wv = rf.tau * (vav - 2.0 * dvw);
    const double alpha = -0.5 * rf.tau * wv;
    // row j = jp + 1 of the panel (needed by every workgroup for the column update)
    const int j = jp + 1;
    {
      const int rbj = j / TS;
      double part2 = 0.0;
      const int nS = rbj - p.S0p + 1;
      if (t < nS) part2 = p.b.ypart[(size_t)(p.S0p + t) * p.npad + j];
      const int nseg = (p.NRB - rbj + p.Gp - 1) / p.Gp;
      if (t >= 128 && t - 128 < nseg)
        part2 += p.b.tpart[((size_t)rbj * p.NRB + (t - 128)) * TS + (j % TS)];
      if (t < ip) {
        const double vjk = Pv[(size_t)j + (size_t)t * ldp], wjk = Pw[(size_t)j + (size_t)t * ldp];
        s_Vj[t] = vjk; s_Wj[t] = wjk;
        part2 -= vjk * s_pw[t] + wjk * s_pv[t];
      }
      const double yj = block_sum(part2, s_red);
      wj = rf.tau * yj + alpha * 1.0;   // v_j = 1
    }
    if (r >= j && r < p.npad) {
      // y_r
      const int rb = r / TS;
      double y = 0.0;
      for (int S = p.S0p; S <= rb; ++S) y += p.b.ypart[(size_t)S * p.npad + r];
      const int nseg = (p.NRB - rb + p.Gp - 1) / p.Gp;
      for (int g = 0; g < nseg; ++g) y += p.b.tpart[((size_t)rb * p.NRB + g) * TS + (r % TS)];
      double accA = 0.0;
      for (int k = 0; k < ip; ++k) {
        const double vrk = Pv[(size_t)r + (size_t)k * ldp], wrk = Pw[(size_t)r + (size_t)k * ldp];
        accA += vrk * s_pw[k] + wrk * s_pv[k];
        accB += vrk * s_Wj[k] + wrk * s_Vj[k];
      }
      v_r = (r == j) ? 1.0 : p.b.xbuf[r] * rf.scale;
      if (r >= p.n) v_r = 0.0;
      w_r = (r == j) ? wj : rf.tau * (y - accA) + alpha * v_r;
      if (r >= p.n) w_r = 0.0;
      Pv[(size_t)r + (size_t)ip * ldp] = v_r;
      Pv2[(size_t)r + (size_t)ip * ldp] = v_r;
      Pw[(size_t)r + (size_t)ip * ldp] = w_r;
      if (r < p.n) {
        p.A[(size_t)r + (size_t)jp * p.lda] = (r == j) ? rf.beta : v_r;
        if (p.V) p.V[(size_t)r + (size_t)jp * p.ldv] = v_r;
      }
      if (r == j) { p.e[jp] = rf.beta; p.tau[jp] = rf.tau; }
      accB += v_r * wj + w_r * 1.0;   // k = ip term: V(j,ip) = 1, W(j,ip) = wj
    } else if (r < j && r < p.npad && r >= p.r0) {
      Pv[(size_t)r + (size_t)ip * ldp] = 0.0;
      Pv2[(size_t)r + (size_t)ip * ldp] = 0.0;
      Pw[(size_t)r + (size_t)ip * ldp] = 0.0;
    }
  }
  if (!p.update) return;

  const int j = p.j;
  if (!p.finalize && p.i > 0) {   // not used by the driver (kept for completeness)
    if (t < p.i) {
      s_Vj[t] = Pv[(size_t)j + (size_t)t * ldp];
      s_Wj[t] = Pw[(size_t)j + (size_t)t * ldp];
    }
    __syncthreads();
    if (r >= j && r < p.npad)
      for (int k = 0; k < p.i; ++k)
        accB += Pv[(size_t)r + (size_t)k * ldp] * s_Wj[k] + Pw[(size_t)r + (size_t)k * ldp] * s_Vj[k];
  }
  double sq = 0.0;
  if (r >= j && r < p.n) {
    const double a = p.A[(size_t)r + (size_t)j * p.lda] - accB;
    p.A[(size_t)r + (size_t)j * p.lda] = a;
    if (r == j) { p.d[j] = a; p.b.xbuf[r] = 0.0; }
    else {
      p.b.xbuf[r] = a;
      if (r == j + 1) p.b.scal[0] = a;
      else sq = a * a;
    }
  }
  const double tot = block_sum(sq, s_red);
  if (t == 0) p.b.normpart[blockIdx.x] = tot;
}

// ------------------------------------------------------------------------------ symv
struct SymvArgs {
  int n, npad, lda;
  const double *A;
  SytrdBufs b;
  int j;            // column whose reflector is applied; active rows/cols > j
  int i;            // in-panel index (number of finished panel columns)
  int S0, G, NRB;   // first active strip, row blocks per unit, total row blocks
  int nseg_max;     // grid: nseg_max x (NRB - S0) units, then ndot dot blocks
  int nunits, ndot, dot_r0;
  int nchunks;      // chunks that produced normpart
};

template <bool DIAG>
__device__ __forceinline__ void symv_block(const double *__restrict__ Acol, int lda, int rloc0,
                                           int cloc0, const double *__restrict__ svc, double vr0,
                                           double vr1, double &y0, double &y1, double (&tc)[32]) {
  // Acol points at A(row0 + 2*lane, col0): rows rloc0, rloc0+1 (indices inside the 128x128
  // diagonal tile, only meaningful when DIAG), columns cloc0 .. cloc0+31.
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    double2 a[16];
#pragma unroll
    for (int cc = 0; cc < 16; ++cc)
      a[cc] = *reinterpret_cast<const double2 *>(Acol + (size_t)(h * 16 + cc) * lda);
#pragma unroll
    for (int cc = 0; cc < 16; ++cc) {
      const int c = h * 16 + cc;
      double ax = a[cc].x, ay = a[cc].y;
      if (DIAG) {
        const int cl = cloc0 + c;
        // row part uses r >= c, column part r > c
        const double rx = (rloc0 >= cl) ? ax : 0.0, ry = (rloc0 + 1 >= cl) ? ay : 0.0;
        const double cx = (rloc0 > cl) ? ax : 0.0, cy = (rloc0 + 1 > cl) ? ay : 0.0;
        const double vc = svc[c];
        y0 += rx * vc; y1 += ry * vc;
        tc[c] += cx * vr0 + cy * vr1;
      } else {
        const double vc = svc[c];
        y0 += ax * vc; y1 += ay * vc;
        tc[c] += ax * vr0 + ay * vr1;
      }
    }
  }
}

__global__ __launch_bounds__(256, 2) void symv_kernel(SymvArgs p) {
  __shared__ double s_vc[TS];          // v on the strip's columns
  __shared__ double s_y[2][4][TS];     // per-wave row-part partials, double buffered
  __shared__ double s_red[8];
  __shared__ double s_dot[4][2 * NBP];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const Refl rf = reflector(p.b.normpart, p.nchunks, p.b.scal[0]);
  const int j1 = p.j + 1;
  const double *__restrict__ xbuf = p.b.xbuf;

  if ((int)blockIdx.x >= p.nunits) {
    // ---- panel products: partial V^T v and W^T v over 256 rows
    const int blk = blockIdx.x - p.nunits;
    const int r = p.dot_r0 + blk * 256 + t;
    double v = 0.0;
    if (r < p.n && r >= j1) v = (r == j1) ? 1.0 : xbuf[r] * rf.scale;
    const double *Pv = p.b.P, *Pw = p.b.P + (size_t)NBP * p.npad;
    for (int k = 0; k < p.i; ++k) {
      double a = 0.0, b = 0.0;
      if (r < p.npad) { a = Pv[(size_t)r + (size_t)k * p.npad] * v; b = Pw[(size_t)r + (size_t)k * p.npad] * v; }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) { a += __shfl_down(a, o, 64); b += __shfl_down(b, o, 64); }
      if (lane == 0) { s_dot[wave][k] = a; s_dot[wave][NBP + k] = b; }
    }
    __syncthreads();
    if (t < 2 * NBP) {
      const int k = t & (NBP - 1);
      double a = 0.0;
      if (k < p.i) a = (s_dot[0][t] + s_dot[1][t]) + (s_dot[2][t] + s_dot[3][t]);
      p.b.dotpart[(size_t)blk * 2 * NBP + t] = a;
    }
    return;
  }

  // ---- symv unit (strip S, segment g)
  const int g = blockIdx.x % p.nseg_max, S = p.S0 + blockIdx.x / p.nseg_max;
  const int rb0 = S + g * p.G;
  int rb1 = rb0 + p.G; if (rb1 > p.NRB) rb1 = p.NRB;
  double vav = 0.0;
  if (rb0 < p.NRB) {
    if (t < TS) {
      const int c = S * TS + t;
      double v = 0.0;
      if (c >= j1 && c < p.n) v = (c == j1) ? 1.0 : xbuf[c] * rf.scale;
      s_vc[t] = v;
    }
    __syncthreads();
    double tc[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) tc[c] = 0.0;
    const int col0 = S * TS + wave * 32;
    const double *svc = s_vc + wave * 32;
    int buf = 0;
    for (int rb = rb0; rb < rb1; ++rb) {
      const int row = rb * TS + 2 * lane;
      // v on this lane's two rows
      const double2 x = *reinterpret_cast<const double2 *>(xbuf + row);
      double vr0 = 0.0, vr1 = 0.0;
      if (row >= j1 && row < p.n) vr0 = (row == j1) ? 1.0 : x.x * rf.scale;
      if (row + 1 >= j1 && row + 1 < p.n) vr1 = (row + 1 == j1) ? 1.0 : x.y * rf.scale;
      const double *Acol = p.A + (size_t)row + (size_t)col0 * p.lda;
      double y0 = 0.0, y1 = 0.0;
      if (rb == S) symv_block<true>(Acol, p.lda, 2 * lane, wave * 32, svc, vr0, vr1, y0, y1, tc);
      else symv_block<false>(Acol, p.lda, 0, 0, svc, vr0, vr1, y0, y1, tc);
      *reinterpret_cast<double2 *>(&s_y[buf][wave][2 * lane]) = make_double2(y0, y1);
      __syncthreads();
      if (t < TS) {
        const double ys = (s_y[buf][0][t] + s_y[buf][1][t]) + (s_y[buf][2][t] + s_y[buf][3][t]);
        const int rr = rb * TS + t;
        p.b.ypart[(size_t)S * p.npad + rr] = ys;
        double vr = 0.0;
        if (rr >= j1 && rr < p.n) vr = (rr == j1) ? 1.0 : xbuf[rr] * rf.scale;
        vav += vr * ys;
      }
      buf ^= 1;
    }
    // column part: reduce the 32 accumulators across the wave, once per unit
#pragma unroll
    for (int c = 0; c < 32; ++c) {
      double v = tc[c];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
      tc[c] = v;
    }
    if (lane == 0) {
#pragma unroll
      for (int c = 0; c < 32; ++c) {
        p.b.tpart[((size_t)S * p.NRB + g) * TS + wave * 32 + c] = tc[c];
        vav += svc[c] * tc[c];
      }
    }
  }
  const double tot = block_sum(vav, s_red);
  if (t == 0) p.b.vavpart[blockIdx.x] = tot;
}

inline size_t al256(size_t b) { return (b + 255) & ~(size_t)255; }

// Optional instrumentation for bench.py's roofline line: HIP events around every symv launch
// on the launch stream.  Off by default (events cost host time).
struct SymvProfile {
  bool enabled = false;
  std::vector<hipEvent_t> ev;
  size_t used = 0;
  double bytes = 0.0;
  long long launches = 0;
  double seconds = 0.0;
} g_prof;

struct Layout {
  int npad, NRB, nch;
  size_t off_x, off_P, off_y, off_t, off_vav, off_norm, off_dot, off_scal, total;
  explicit Layout(int n) {
    npad = round_up(n > 0 ? n : 1, TS); NRB = npad / TS; nch = npad / CH + 2;
    size_t o = 0;
    off_x = o; o += al256((size_t)npad * 8);
    off_P = o; o += al256((size_t)npad * 3 * NBP * 8);
    off_y = o; o += al256((size_t)NRB * npad * 8);
    off_t = o; o += al256((size_t)NRB * NRB * TS * 8);
    off_vav = o; o += al256((size_t)NRB * NRB * 8 + 64);
    off_norm = o; o += al256((size_t)nch * 8);
    off_dot = o; o += al256((size_t)nch * 2 * NBP * 8);
    off_scal = o; o += 256;
    total = o;
  }
};

}  // namespace

size_t sytrd_work_bytes(int n) { return Layout(n).total; }

void sytrd_lower(hipStream_t s, int n, double *A, int lda, double *d, double *e, double *tau,
                 double *V, int ldv, void *work) {
  if (n <= 0) return;
  const Layout L(n);
  char *w = (char *)work;
  SytrdBufs b;
  b.xbuf = (double *)(w + L.off_x); b.P = (double *)(w + L.off_P);
  b.ypart = (double *)(w + L.off_y); b.tpart = (double *)(w + L.off_t);
  b.vavpart = (double *)(w + L.off_vav); b.normpart = (double *)(w + L.off_norm);
  b.dotpart = (double *)(w + L.off_dot); b.scal = (double *)(w + L.off_scal);
  (void)hipMemsetAsync(work, 0, L.total, s);
  const int npad = L.npad, NRB = L.NRB;

  ColupdArgs c{};
  c.n = n; c.npad = npad; c.lda = lda; c.ldv = ldv; c.A = A; c.V = V; c.d = d; c.e = e; c.tau = tau;
  c.b = b; c.NRB = NRB;
  SymvArgs sv{};
  sv.n = n; sv.npad = npad; sv.lda = lda; sv.A = A; sv.b = b; sv.NRB = NRB;

  auto launch_colupd = [&](int row_from) {
    c.r0 = (row_from / CH) * CH;
    const int nblk = ceil_div(npad - c.r0, CH);
    hipLaunchKernelGGL(colupd_kernel, dim3(nblk), dim3(CH), 0, s, c);
    return nblk;
  };

  int nchunks_cur = 0;
  for (int j0 = 0; j0 < n - 1; j0 += NBP) {
    const int pw = (n - 1 - j0 < NBP) ? n - 1 - j0 : NBP;   // reflector columns j0 .. j0+pw-1
    // first column of the panel: nothing deferred yet
    c.finalize = 0; c.update = 1; c.j = j0; c.i = 0;
    nchunks_cur = launch_colupd(j0);
    for (int i = 0; i < pw; ++i) {
      const int j = j0 + i;
      // y = A22 v and the panel products
      sv.j = j; sv.i = i; sv.S0 = (j + 1) / TS;
      const int T = NRB - sv.S0;
      int G = (T * T) / 1024; if (G < 1) G = 1; if (G > 8) G = 8;
      sv.G = G; sv.nseg_max = ceil_div(T, G); sv.nunits = sv.nseg_max * T;
      sv.dot_r0 = ((j + 1) / 256) * 256;
      sv.ndot = (i > 0) ? ceil_div(npad - sv.dot_r0, 256) : 0;
      sv.nchunks = nchunks_cur;
      hipEvent_t e0 = nullptr, e1 = nullptr;
      if (g_prof.enabled) {
        if (g_prof.used + 2 > g_prof.ev.size()) {
          const size_t old = g_prof.ev.size();
          g_prof.ev.resize(old + 4096);
          for (size_t q = old; q < g_prof.ev.size(); ++q) (void)hipEventCreate(&g_prof.ev[q]);
        }
        e0 = g_prof.ev[g_prof.used++]; e1 = g_prof.ev[g_prof.used++];
        (void)hipEventRecord(e0, s);
      }
      hipLaunchKernelGGL(symv_kernel, dim3(sv.nunits + sv.ndot), dim3(256), 0, s, sv);
      if (g_prof.enabled) {
        (void)hipEventRecord(e1, s);
        const double m = (double)(n - j - 1);
        g_prof.bytes += 8.0 * m * (m + 1.0) * 0.5;   // lower triangle of the active matrix, once
        g_prof.launches += 1;
      }
      // finish w_j, then update column j+1 (unless the panel ends here)
      c.finalize = 1; c.jp = j; c.ip = i; c.S0p = sv.S0; c.Gp = G; c.nchunks_p = nchunks_cur;
      c.ndot_p = sv.ndot; c.nunits_p = sv.nunits;
      c.update = (i + 1 < pw) ? 1 : 0; c.j = j + 1; c.i = i + 1;
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
  c.finalize = 0; c.update = 1; c.j = n - 1; c.i = 0;
  launch_colupd(n - 1);
}

void symv_profile_enable(bool on) {
  g_prof.enabled = on;
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
