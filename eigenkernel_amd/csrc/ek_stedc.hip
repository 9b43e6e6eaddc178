// ek_stedc.hip -- divide & conquer eigensolver for the symmetric tridiagonal matrix,
// T = Z diag(w) Z^T.  Replaces PDSTEDC('I') at solver_scalapack_all.f90:96 (1x1 grid).
//
// Cuppen's method with Gu/Eisenstat stabilisation, organised for the GPU:
//   * the recursion tree is built on the host once; all sub-problems of one tree height
//     are processed by the same batched launches, so the whole solve is a fixed,
//     input-independent launch sequence with NO device->host synchronisation
//     (deflation counts stay on the device);
//   * leaves (order <= 32): implicit QL, one wavefront per leaf, Z block in LDS;
//   * a merge is: rank-sort of the poles -> deflation scan (DLAED2, incl. the column types
//     top-only / dense / bottom-only) -> column permutation + Givens rotations -> secular
//     equation (a wave per root, origin shifted to the nearer pole) -> Loewner weights ->
//     eigenvector matrix S of the rank-one update -> ONE batched MFMA launch per tree height
//     holding two GEMMs per merge, Q(top) = W(top, [top|dense]) S and Q(bottom) =
//     W(bottom, [dense|bottom]) S, whose sizes and offsets (functions of the deflation
//     counts) are written by the deflation kernel and read by the GEMM on the device.
//     Deflated columns are copied.
// The O(n^3) part (4/3 N^3 nominal) is therefore entirely in ek::gemm and
// shrinks with deflation exactly as LAPACK's does.
#include "ek_common.h"

#include <algorithm>
#include <vector>

namespace ek {
namespace {

constexpr int LEAF = 32;

struct Merge { int off, n1, n, pad; };

struct DcBufs {
  double *d, *e;                 // n: current eigenvalues (column order), scaled off-diagonals
  double *dsort, *zsort;         // n: poles / weights in ascending pole order
  double *dl, *zl;               // n: surviving poles / weights
  double *zhat, *tauv;           // n: Loewner weights, secular shifts
  int *korig;                    // n: origin pole of each root
  int *perm, *wcol;              // n: sorted position -> local column / W column
  int *grp;                      // n: survivor (ascending pole order) -> W column = row of S
  int *spos;                     // n: survivor -> sorted position (scratch of the deflation scan)
  long long *goffs;              // per merge 2 x {A, B, C} element offsets of the two merge GEMMs
  int *gdims;                    // per merge 2 x {M, N, K}
  int ldw, ldq, lds;             // leading dimensions of W (the output array), Q, S
  // Compact storage of the bases (selected columns <= half of them: a grid cell's share, a *_select arm): below the top
  // merge every block lives inside one half of the matrix, so the blocks of the second half are stored hq / hw / hs
  // columns to the left -- an n x (n - n/2) array holds a whole level.  INT_MAX: no shift (columns as numbered).
  int hq = 0x7fffffff, hw = 0x7fffffff, hs = 0x7fffffff;
  int *rotp, *rotn;              // n: rotation list (sorted positions)
  double *rotc, *rots;           // n
  double *rho;                   // per merge
  int *k, *nrot;                 // per merge
  Merge *merges;
  double *orgnrm;                // 1
  // team form of a height (StedcTeam): S is formed for the columns of rank tR's strips only (tP == 0: all columns)
  int tP = 0, tR = 0;
  // ... and the secular equation / the Loewner weights for the roots (weights) tc * tR .. tc * (tR + 1) - 1 of every merge only
  int tc = 0;
  double *kd = nullptr;          // n: the origin pole of each root as a double (it travels with tauv in the team's all-gathers)
};
constexpr int kStrip = 128;      // strip width of the team form (compact basis columns)

__device__ __forceinline__ size_t cq(const DcBufs &b, int c) { return (size_t)(c >= b.hq ? c - b.hq : c); }
__device__ __forceinline__ size_t cw(const DcBufs &b, int c) { return (size_t)(c >= b.hw ? c - b.hw : c); }
__device__ __forceinline__ size_t cs(const DcBufs &b, int c) { return (size_t)(c >= b.hs ? c - b.hs : c); }

// ------------------------------------------------------------------ scaling / splits
__global__ void dc_scale_kernel(int n, const double *__restrict__ din, const double *__restrict__ ein,
                                DcBufs b) {
  __shared__ double red[256];
  const int t = threadIdx.x;
  double m = 0.0;
  for (int i = t; i < n; i += 256) m = fmax(m, fabs(din[i]));
  for (int i = t; i < n - 1; i += 256) m = fmax(m, fabs(ein[i]));
  red[t] = m;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (t < o) red[t] = fmax(red[t], red[t + o]); __syncthreads(); }
  const double nrm = red[0] > 0.0 ? red[0] : 1.0;
  if (t == 0) b.orgnrm[0] = nrm;
  const double r = 1.0 / nrm;
  for (int i = t; i < n; i += 256) { b.d[i] = din[i] * r; b.e[i] = (i < n - 1) ? ein[i] * r : 0.0; }
}

__global__ void dc_split_kernel(int nmerges, DcBufs b) {
  // each split position belongs to exactly one merge: no conflicts
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= nmerges) return;
  const Merge mg = b.merges[m];
  const double rho = fabs(b.e[mg.off + mg.n1 - 1]);
  b.d[mg.off + mg.n1 - 1] -= rho;
  b.d[mg.off + mg.n1] -= rho;
}

// ------------------------------------------------------------------ leaves: implicit QL
struct Leaf { int off, n; };

__global__ __launch_bounds__(64) void dc_leaf_kernel(const Leaf *__restrict__ leaves, DcBufs b,
                                                     double *__restrict__ Q, int ldq, int *info) {
  __shared__ double sd[LEAF], se[LEAF + 1];
  __shared__ double sZ[LEAF][LEAF + 1];   // sZ[col][row]
  const Leaf lf = leaves[blockIdx.x];
  const int n = lf.n, lane = threadIdx.x;
  if (lane < n) { sd[lane] = b.d[lf.off + lane]; se[lane] = (lane < n - 1) ? b.e[lf.off + lane] : 0.0; }
  for (int idx = lane; idx < LEAF * (LEAF + 1); idx += 64) (&sZ[0][0])[idx] = 0.0;
  __syncthreads();
  if (lane < n) sZ[lane][lane] = 1.0;
  __syncthreads();
  const double eps = 1.1102230246251565e-16;
  // every lane runs the same scalar recurrence (uniform control flow); lane r owns row r of Z
  for (int l = 0; l < n; ++l) {
    int iter = 0;
    while (true) {
      int m;
      for (m = l; m < n - 1; ++m) {
        const double dd = fabs(sd[m]) + fabs(sd[m + 1]);
        if (fabs(se[m]) <= eps * dd) break;
      }
      if (m == l) break;
      if (iter++ == 60) { if (lane == 0) atomicMax(info, lf.off + l + 1); break; }
      double g = (sd[l + 1] - sd[l]) / (2.0 * se[l]);
      double r = hypot(g, 1.0);
      g = sd[m] - sd[l] + se[l] / (g + copysign(r, g));
      double s = 1.0, c = 1.0, p = 0.0;
      int i;
      bool under = false;
      for (i = m - 1; i >= l; --i) {
        const double f = s * se[i], bb = c * se[i];
        r = hypot(f, g);
        __syncthreads();
        if (lane == 0) se[i + 1] = r;
        if (r == 0.0) {
          __syncthreads();
          if (lane == 0) { sd[i + 1] -= p; se[m] = 0.0; }
          under = true;
          break;
        }
        s = f / r; c = g / r;
        g = sd[i + 1] - p;
        r = (sd[i] - g) * s + 2.0 * c * bb;
        p = s * r;
        __syncthreads();
        if (lane == 0) sd[i + 1] = g + p;
        g = c * r - bb;
        if (lane < n) {
          const double f2 = sZ[i + 1][lane], z0 = sZ[i][lane];
          sZ[i + 1][lane] = s * z0 + c * f2;
          sZ[i][lane] = c * z0 - s * f2;
        }
      }
      __syncthreads();
      if (under) continue;
      if (lane == 0) { sd[l] -= p; se[l] = g; se[m] = 0.0; }
      __syncthreads();
    }
    __syncthreads();
  }
  __syncthreads();
  if (lane < n) {
    b.d[lf.off + lane] = sd[lane];
    for (int c = 0; c < n; ++c) Q[(size_t)(lf.off + lane) + cq(b, lf.off + c) * ldq] = sZ[c][lane];
  }
}

// ------------------------------------------------------------------ merge step 1: z and rank sort
constexpr int RP = 16;   // lanes that share one element of a rank sort (they split the comparisons)
__device__ __forceinline__ int rank_sum(int v) {
#pragma unroll
  for (int o = 1; o < RP; o <<= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__global__ void dc_sort_kernel(int mbeg, DcBufs b, const double *__restrict__ Q, int ldq) {
  const Merge mg = b.merges[mbeg + blockIdx.y];
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int t = gid / RP, sub = gid % RP;
  if (t >= mg.n) return;
  const double *d = b.d + mg.off;
  // NaN (numerical breakdown upstream) sorts as +inf so that ranks stay a permutation and
  // every later index stays in bounds; the breakdown is reported through info at the end
  const double dt = (d[t] == d[t]) ? d[t] : INFINITY;
  int rank = 0;
  for (int i = sub; i < mg.n; i += RP) {
    const double di = (d[i] == d[i]) ? d[i] : INFINITY;
    rank += (di < dt || (di == dt && i < t)) ? 1 : 0;
  }
  rank = rank_sum(rank);
  if (sub != 0) return;
  const double rho_in = b.e[mg.off + mg.n1 - 1];
  const double is2 = 0.70710678118654752440;
  double z;
  if (t < mg.n1) z = Q[(size_t)(mg.off + mg.n1 - 1) + cq(b, mg.off + t) * ldq] * is2;
  else z = (rho_in < 0.0 ? -1.0 : 1.0) * Q[(size_t)(mg.off + mg.n1) + cq(b, mg.off + t) * ldq] * is2;
  b.perm[mg.off + rank] = t;
  b.dsort[mg.off + rank] = dt;
  b.zsort[mg.off + rank] = z;
}

// ------------------------------------------------------------------ merge step 2: deflation (DLAED2)
__global__ __launch_bounds__(256) void dc_deflate_kernel(int mbeg, DcBufs b) {
  __shared__ double red[256];
  const int mi = mbeg + blockIdx.x;
  const Merge mg = b.merges[mi];
  const int n = mg.n, off = mg.off, t = threadIdx.x;
  double *ds = b.dsort + off, *zs = b.zsort + off;
  double dm = 0.0, zm = 0.0;
  for (int i = t; i < n; i += 256) { dm = fmax(dm, fabs(ds[i])); zm = fmax(zm, fabs(zs[i])); }
  red[t] = dm; __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (t < o) red[t] = fmax(red[t], red[t + o]); __syncthreads(); }
  dm = red[0]; __syncthreads();
  red[t] = zm; __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (t < o) red[t] = fmax(red[t], red[t + o]); __syncthreads(); }
  zm = red[0];
  // the scan itself is sequential (thread 0); the other threads stage the sorted poles,
  // weights and origins through LDS in chunks so that thread 0 never waits on global memory
  constexpr int CHUNK = 1024;
  __shared__ double c_d[CHUNK], c_z[CHUNK];
  __shared__ int c_p[CHUNK];
  const double eps = 1.1102230246251565e-16;
  const double rho = fabs(2.0 * b.e[off + mg.n1 - 1]);
  const double tol = 8.0 * eps * fmax(dm, zm);
  int *wcol = b.wcol + off, *rotp = b.rotp + off, *rotn = b.rotn + off;
  int *grp = b.grp + off, *spos = b.spos + off;
  const int *perm = b.perm + off;
  // column types (DLAED2's COLTYP): 1 = non-zero in the top n1 rows only (from Q1),
  // 3 = bottom rows only (from Q2), 2 = dense (a rotated pair from both); stored in grp[]
  // temporarily as the survivor's type
  double *rotc = b.rotc + off, *rots = b.rots + off, *dl = b.dl + off, *zl = b.zl + off;
  double *dout = b.d + off;
  int k = 0, ndf = 0, nrot = 0;
  const bool all_deflate = (rho * zm <= tol);

  // ---- parallel fast path: when no two neighbouring surviving poles are close enough to be
  // rotated together (the generic case) the scan has no loop-carried state, so deflation is a
  // stream compaction: each thread owns a contiguous chunk, a block-level prefix sum places
  // the outputs.  Any close pair sends the whole merge to the sequential DLAED2 scan below.
  __shared__ int s_cnt[3][256], s_pre[3][256], s_last[256];
  __shared__ int s_anyclose;
  if (!all_deflate) {
    const int per = (n + 255) / 256, i0 = t * per, i1 = (i0 + per < n) ? i0 + per : n;
    int ct = 0, c1 = 0, c3 = 0, last = -1, first = -1;
    bool close = false;
    double dprev = 0.0, zprev = 0.0;
    if (t == 0) s_anyclose = 0;
    for (int i = i0; i < i1; ++i) {
      const double zi = zs[i], di = ds[i];
      if (rho * fabs(zi) <= tol) { ++ct; continue; }
      if (perm[i] < mg.n1) ++c1; else ++c3;
      if (last >= 0) close = close || (fabs((di - dprev) * zi * zprev) <= tol * (zi * zi + zprev * zprev));
      else first = i;
      last = i; dprev = di; zprev = zi;
    }
    s_cnt[0][t] = ct; s_cnt[1][t] = c1; s_cnt[2][t] = c3; s_last[t] = last;
    __syncthreads();
    if (first >= 0) {
      int u = t - 1;
      while (u >= 0 && s_last[u] < 0) --u;
      if (u >= 0) {
        const int pidx = s_last[u];
        const double zi = zs[first], zp = zs[pidx];
        close = close || (fabs((ds[first] - ds[pidx]) * zi * zp) <= tol * (zi * zi + zp * zp));
      }
    }
    if (close) s_anyclose = 1;
    __syncthreads();
    if (!s_anyclose) {
      if (t == 0) {
        int a0 = 0, a1 = 0, a3 = 0;
        for (int u = 0; u < 256; ++u) {
          s_pre[0][u] = a0; s_pre[1][u] = a1; s_pre[2][u] = a3;
          a0 += s_cnt[0][u]; a1 += s_cnt[1][u]; a3 += s_cnt[2][u];
        }
        s_cnt[0][0] = a0; s_cnt[1][0] = a1; s_cnt[2][0] = a3;   // totals
      }
      __syncthreads();
      const int ntiny = s_cnt[0][0], k1f = s_cnt[1][0], k3f = s_cnt[2][0];
      const int kf = n - ntiny;
      int tt = s_pre[0][t], j1 = s_pre[1][t], j3 = s_pre[2][t];
      int a = ((i0 < n) ? i0 : n) - tt;
      for (int i = i0; i < i1; ++i) {
        const double zi = zs[i], di = ds[i];
        if (rho * fabs(zi) <= tol) { wcol[i] = n - 1 - tt; dout[n - 1 - tt] = di; ++tt; continue; }
        const int g = (perm[i] < mg.n1) ? j1++ : k1f + j3++;
        dl[a] = di; zl[a] = zi; spos[a] = i; grp[a] = g; wcol[i] = g; ++a;
      }
      if (t == 0) {
        const long long o = off;
        long long *go = b.goffs + 6 * (size_t)mi;
        int *gd = b.gdims + 6 * (size_t)mi;
        const long long ow = (long long)cw(b, off), os = (long long)cs(b, off), oq = (long long)cq(b, off);   // (block columns as stored)
        go[0] = o + ow * b.ldw;                      go[1] = o + os * b.lds;       go[2] = o + oq * b.ldq;
        const int k1e = k1f & ~1;                    // even start of the second product (see below)
        go[3] = o + mg.n1 + (ow + k1e) * b.ldw;      go[4] = o + k1e + os * b.lds; go[5] = o + mg.n1 + oq * b.ldq;
        gd[0] = mg.n1;        gd[1] = kf; gd[2] = k1f;
        gd[3] = n - mg.n1;    gd[4] = kf; gd[5] = k3f + (k1f - k1e);
        b.k[mi] = kf; b.nrot[mi] = 0; b.rho[mi] = rho;
      }
      return;
    }
  }

  int pj = -1, tpj = 0;
  double dpj = 0.0, zpj = 0.0;
  for (int c0 = 0; c0 < n; c0 += CHUNK) {
    const int cn = (n - c0 < CHUNK) ? n - c0 : CHUNK;
    __syncthreads();
    for (int i = t; i < cn; i += 256) { c_d[i] = ds[c0 + i]; c_z[i] = zs[c0 + i]; c_p[i] = perm[c0 + i]; }
    __syncthreads();
    if (t != 0) continue;
    if (all_deflate) {
      for (int ii = 0; ii < cn; ++ii) { wcol[c0 + ii] = n - 1 - ndf; dout[n - 1 - ndf] = c_d[ii]; ++ndf; }
      continue;
    }
    for (int ii = 0; ii < cn; ++ii) {
      const int i = c0 + ii;
      const double zi = c_z[ii], di = c_d[ii];
      const int ti = (c_p[ii] < mg.n1) ? 1 : 3;
      if (rho * fabs(zi) <= tol) { wcol[i] = n - 1 - ndf; dout[n - 1 - ndf] = di; ++ndf; continue; }
      if (pj < 0) { pj = i; dpj = di; zpj = zi; tpj = ti; continue; }
      // DLAED2's test |t c s| <= tol with c = z_i/tau, s = -z_pj/tau, tau^2 = z_i^2 + z_pj^2,
      // evaluated division- and sqrt-free on the (serial) common path
      const double tau2 = zi * zi + zpj * zpj, tt = di - dpj;
      if (fabs(tt * zi * zpj) <= tol * tau2) {
        // deflate pj: rotate columns (pj, i); the combined weight moves to i
        const double tau = sqrt(tau2);
        const double c = zi / tau, s = -zpj / tau;
        rotp[nrot] = pj; rotn[nrot] = i; rotc[nrot] = c; rots[nrot] = s; ++nrot;
        const double dnew_p = dpj * c * c + di * s * s;
        const double dnew_i = dpj * s * s + di * c * c;
        wcol[pj] = n - 1 - ndf; dout[n - 1 - ndf] = dnew_p; ++ndf;
        pj = i; dpj = dnew_i; zpj = tau; tpj = (tpj == ti) ? ti : 2;
      } else {
        spos[k] = pj; grp[k] = tpj; dl[k] = dpj; zl[k] = zpj; ++k;
        pj = i; dpj = di; zpj = zi; tpj = ti;
      }
    }
  }
  if (t != 0) return;
  if (pj >= 0) { spos[k] = pj; grp[k] = tpj; dl[k] = dpj; zl[k] = zpj; ++k; }
  // group the survivors' W columns by type: [top-only | dense | bottom-only]
  int cnt[4] = {0, 0, 0, 0};
  for (int a = 0; a < k; ++a) ++cnt[grp[a]];
  int nxt[4] = {0, 0, cnt[1], cnt[1] + cnt[2]};
  for (int a = 0; a < k; ++a) { const int g = nxt[grp[a]]++; grp[a] = g; wcol[spos[a]] = g; }
  const int k1 = cnt[1], k12 = cnt[1] + cnt[2], k23 = cnt[2] + cnt[3];
  // the two merge GEMMs: Q(top, 0:k) = W(top, 0:k12) S(0:k12, 0:k); Q(bot, 0:k) = W(bot, k1:k) S(k1:k, 0:k)
  {
    const long long o = off;
    long long *go = b.goffs + 6 * (size_t)mi;
    int *gd = b.gdims + 6 * (size_t)mi;
    const long long ow = (long long)cw(b, off), os = (long long)cs(b, off), oq = (long long)cq(b, off);       // (block columns as stored)
    go[0] = o + ow * b.ldw;                      go[1] = o + os * b.lds;       go[2] = o + oq * b.ldq;
    // (the second product starts on an EVEN column of W / row of S: if k1 is odd it takes the last top-only
    // column along, whose bottom rows are zero -- the operands of both products then start on even offsets
    // whenever off and n1 are even, which is what the GEMM's 16-byte loads need)
    const int k1e = k1 & ~1;
    go[3] = o + mg.n1 + (ow + k1e) * b.ldw;      go[4] = o + k1e + os * b.lds; go[5] = o + mg.n1 + oq * b.ldq;
    gd[0] = mg.n1;        gd[1] = k; gd[2] = k12;
    gd[3] = n - mg.n1;    gd[4] = k; gd[5] = k23 + (k1 - k1e);
  }
  b.k[mi] = k; b.nrot[mi] = nrot; b.rho[mi] = rho;
}

// ------------------------------------------------------------------ merge step 3: W = permuted (and rotated) basis
__global__ void dc_permute_kernel(int mbeg, DcBufs b, const double *__restrict__ Q, int ldq,
                                  double *__restrict__ W, int ldw) {
  const Merge mg = b.merges[mbeg + blockIdx.z];
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= mg.n) return;
  const int off = mg.off;
  // compact bases: the merge that spans both halves (the top one) reads a child basis whose OTHER half's rows hold the
  // other child (stored hq columns to the left): those entries are zeros of the block-diagonal basis, not what is stored
  const bool crossing = off < b.hq && off + mg.n > b.hq;
  const bool row_hi = off + r >= b.hq;
  for (int t = blockIdx.y; t < mg.n; t += gridDim.y) {
    const int src = b.perm[off + t], dst = b.wcol[off + t];
    double v = Q[(size_t)(off + r) + cq(b, off + src) * ldq];
    if (crossing && ((off + src >= b.hq) != row_hi)) v = 0.0;
    W[(size_t)(off + r) + cw(b, off + dst) * ldw] = v;
  }
}

__global__ void dc_rotate_kernel(int mbeg, DcBufs b, double *__restrict__ W, int ldw) {
  const int mi = mbeg + blockIdx.y;
  const Merge mg = b.merges[mi];
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= mg.n) return;
  const int off = mg.off, nrot = b.nrot[mi];
  double *Wr = W + (size_t)(off + r) + cw(b, off) * ldw;
  for (int q = 0; q < nrot; ++q) {
    const int cp = b.wcol[off + b.rotp[off + q]], cn = b.wcol[off + b.rotn[off + q]];
    const double c = b.rotc[off + q], s = b.rots[off + q];
    const double a = Wr[(size_t)cp * ldw], bb = Wr[(size_t)cn * ldw];
    Wr[(size_t)cp * ldw] = c * a + s * bb;
    Wr[(size_t)cn * ldw] = c * bb - s * a;
  }
}

// ------------------------------------------------------------------ merge step 4: secular equation (DLAED4)
// root i of 1 + rho * sum_j z_j^2 / (d_j - lambda) = 0 in shifted form lambda = d_K + tau,
// K the nearer pole, so that d_j - lambda_i = (d_j - d_K) - tau keeps full relative accuracy.
#ifndef EK_DC_SP
#define EK_DC_SP 64
#endif
// lanes that share one root / one weight (they split the sum or product over the poles): a whole wave.  A lane's
// pass over its poles is a chain of dependent-latency loads of d and z (L2) with a division each; with 4 lanes
// per root the top merge of N = 16384 took 6 ms in the secular kernel alone.  stedc at N = 16384, same box:
// SP = 4: 0.0862 s (0.0911 with one lane per Loewner weight as well), 8: 0.0797, 16: 0.0769, 32: 0.0735, 64: 0.0728.
constexpr int SP = EK_DC_SP;

// butterfly: every lane of the group ends with the same bits
__device__ __forceinline__ double group_sum(double v) {
#pragma unroll
  for (int o = 1; o < SP; o <<= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double group_prod(double v) {
#pragma unroll
  for (int o = 1; o < SP; o <<= 1) v *= __shfl_xor(v, o, 64);
  return v;
}

// The SP lanes of a group run the same control flow on the same (reduced) values.
__device__ void secular_root(int k, int i, int sub, const double *__restrict__ d,
                             const double *__restrict__ z, double rho, int *Kout, double *tauout) {
  if (k == 1) { *Kout = 0; *tauout = rho * z[0] * z[0]; return; }
  const double eps = 1.1102230246251565e-16;
  int K;
  double lo, hi;
  if (i < k - 1) {
    const double di = d[i];
    const double mid = 0.5 * (d[i + 1] - di);
    double f = 0.0;
    for (int j = sub; j < k; j += SP) f += z[j] * z[j] / ((d[j] - di) - mid);
    f = 1.0 + rho * group_sum(f);
    if (f > 0.0) { K = i; lo = 0.0; hi = mid; }
    else { K = i + 1; lo = -mid; hi = 0.0; }
  } else {
    double zn = 0.0;
    for (int j = sub; j < k; j += SP) zn += z[j] * z[j];
    zn = group_sum(zn);
    K = k - 1; lo = 0.0; hi = rho * zn;
  }
  const double dK = d[K];
  double tau = 0.5 * (lo + hi);
  if (i == k - 1) {
    double f = 0.0;
    for (int j = sub; j < k; j += SP) f += z[j] * z[j] / ((d[j] - dK) - tau);
    f = 1.0 + rho * group_sum(f);
    if (f > 0.0) hi = tau; else lo = tau;
    tau = 0.5 * (lo + hi);
  }
  for (int it = 0; it < 100; ++it) {
    double psi = 0.0, dpsi = 0.0, phi = 0.0, dphi = 0.0, err = 0.0;
    for (int j = sub; j <= i; j += SP) {
      const double t = z[j] / ((d[j] - dK) - tau);
      psi += z[j] * t; dpsi += t * t; err += fabs(psi);
    }
    for (int j = i + 1 + sub; j < k; j += SP) {
      const double t = z[j] / ((d[j] - dK) - tau);
      phi += z[j] * t; dphi += t * t; err += fabs(phi);
    }
    psi = rho * group_sum(psi); dpsi = rho * group_sum(dpsi);
    phi = rho * group_sum(phi); dphi = rho * group_sum(dphi);
    err = rho * group_sum(err);
    const double f = 1.0 + psi + phi;
    err = 8.0 * (phi - psi) + err + 2.0 + fabs(tau) * (dpsi + dphi);
    if (fabs(f) <= eps * err) break;
    if (f > 0.0) hi = tau; else lo = tau;
    if (!(hi - lo > 2.0 * eps * fmax(fabs(lo), fabs(hi)))) break;
    double next;
    const double dl_ = (d[i] - dK) - tau;            // < 0
    const double S = dpsi * dl_ * dl_, s = psi - dpsi * dl_;
    if (i < k - 1) {
      const double dr = (d[i + 1] - dK) - tau;        // > 0
      const double R = dphi * dr * dr, r = phi - dphi * dr;
      const double a = 1.0 + s + r;
      const double bq = -(a * (dl_ + dr) + S + R);
      const double cq = a * dl_ * dr + S * dr + R * dl_;
      double eta;
      if (a == 0.0) eta = (bq != 0.0) ? -cq / bq : 0.0;
      else {
        double disc = bq * bq - 4.0 * a * cq;
        if (disc < 0.0) disc = 0.0;
        const double q = -0.5 * (bq + copysign(sqrt(disc), bq));
        const double e1 = q / a, e2 = (q != 0.0) ? cq / q : e1;
        const bool in1 = (e1 > dl_ && e1 < dr), in2 = (e2 > dl_ && e2 < dr);
        eta = in1 ? e1 : e2;
        if (in1 && in2) eta = (fabs(e1) < fabs(e2)) ? e1 : e2;
      }
      next = tau + eta;
    } else {
      const double a = 1.0 + s;
      next = (a > 0.0) ? tau + (dl_ + S / a) : hi;
    }
    if (!(next > lo && next < hi)) next = 0.5 * (lo + hi);
    tau = next;
  }
  *Kout = K; *tauout = tau;
}

__global__ void dc_secular_kernel(int mbeg, DcBufs b) {
  const int mi = mbeg + blockIdx.y;
  const Merge mg = b.merges[mi];
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = gid / SP, sub = gid % SP;
  const int k = b.k[mi];
  if (i >= k) return;
  if (b.tP > 0 && i / b.tc != b.tR) return;      // (another rank's roots)
  int K; double tau;
  secular_root(k, i, sub, b.dl + mg.off, b.zl + mg.off, b.rho[mi], &K, &tau);
  if (sub == 0) {
    b.korig[mg.off + i] = K;
    b.tauv[mg.off + i] = tau;
    b.d[mg.off + i] = b.dl[mg.off + K] + tau;    // new eigenvalue, W column i
    if (b.tP > 0) b.kd[mg.off + i] = (double)K;
  }
}
// after the team's all-gather of kd: the origins as the later kernels read them
__global__ void dc_korig_kernel(int mbeg, DcBufs b) {
  const int mi = mbeg + blockIdx.y;
  const Merge mg = b.merges[mi];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < b.k[mi]) b.korig[mg.off + i] = (int)b.kd[mg.off + i];
}

// Loewner / Gu-Eisenstat weights: zhat_j = sign(z_j) sqrt(| prod_i (d_j - lam_i) / prod_{i!=j} (d_j - d_i) |)
__global__ void dc_zhat_kernel(int mbeg, DcBufs b) {
  const int mi = mbeg + blockIdx.y;
  const Merge mg = b.merges[mi];
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = gid / SP, sub = gid % SP;          // SP lanes share one weight (they split the product)
  const int k = b.k[mi];
  if (j >= k) return;
  if (b.tP > 0 && j / b.tc != b.tR) return;      // (another rank's weights)
  const double *dl = b.dl + mg.off, *tauv = b.tauv + mg.off;
  const int *korig = b.korig + mg.off;
  const double dj = dl[j];
  double p = 1.0;
  for (int i = sub; i < k; i += SP) {
    const double del = (dj - dl[korig[i]]) - tauv[i];   // d_j - lambda_i
    p *= (i == j) ? del : del / (dj - dl[i]);
  }
  p = group_prod(p);
  if (sub == 0) b.zhat[mg.off + j] = copysign(sqrt(fabs(p)), b.zl[mg.off + j]);
}

// Column c < k of S: the normalised eigenvector of the rank-one update, entry of pole a stored
// in row grp[a] (the W column that carries pole a).  Rows >= k are never read by the GEMMs.
__global__ __launch_bounds__(256) void dc_vectors_kernel(int mbeg, DcBufs b, double *__restrict__ S,
                                                         int lds) {
  __shared__ double red[4];
  const int mi = mbeg + blockIdx.y;
  const Merge mg = b.merges[mi];
  const int c = blockIdx.x, t = threadIdx.x;
  const int k = b.k[mi];
  if (c >= k) return;
  if (b.tP > 0 && (int)(cq(b, mg.off + c) / kStrip) % b.tP != b.tR) return;   // (a column of another rank's strip)
  double *col = S + (size_t)mg.off + cs(b, mg.off + c) * lds;
  const double *dl = b.dl + mg.off, *zh = b.zhat + mg.off;
  const int *grp = b.grp + mg.off;
  const double dK = dl[b.korig[mg.off + c]], tau = b.tauv[mg.off + c];
  double ss = 0.0;
  for (int j = t; j < k; j += 256) {
    const double u = zh[j] / ((dl[j] - dK) - tau);
    ss += u * u;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) ss += __shfl_down(ss, o, 64);
  if ((t & 63) == 0) red[t >> 6] = ss;
  __syncthreads();
  const double inv = 1.0 / sqrt((red[0] + red[1]) + (red[2] + red[3]));
  for (int j = t; j < k; j += 256) col[grp[j]] = zh[j] / ((dl[j] - dK) - tau) * inv;
}

// Team form of a height: the two products of every merge cut into the pieces that fall into one rank's strips of the
// compact basis array.  The table (which merge, which product, first column inside the merge, width) depends on the
// order and the team only and is built on the host; the deflation's counts turn it into GEMM operands here.
struct TeamEntry { int mi, prod, c0, wd; };
__global__ void dc_team_entries_kernel(const TeamEntry *__restrict__ tab, int cnt, DcBufs b, long long *__restrict__ eoffs,
                                       int *__restrict__ edims) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= cnt) return;
  const TeamEntry te = tab[e];
  const long long *go = b.goffs + 6 * (size_t)te.mi + 3 * te.prod;
  const int *gd = b.gdims + 6 * (size_t)te.mi + 3 * te.prod;
  int nn = gd[1] - te.c0;                      // columns of the product (roots) from c0 on
  nn = nn < 0 ? 0 : (nn > te.wd ? te.wd : nn);
  eoffs[3 * e] = go[0]; eoffs[3 * e + 1] = go[1] + (long long)te.c0 * b.lds; eoffs[3 * e + 2] = go[2] + (long long)te.c0 * b.ldq;
  edims[3 * e] = gd[0]; edims[3 * e + 1] = nn; edims[3 * e + 2] = gd[2];
}

// Deflated eigenpairs: columns c >= k of W are final eigenvectors
__global__ void dc_copy_deflated_kernel(int mbeg, DcBufs b, const double *__restrict__ W, int ldw,
                                        double *__restrict__ Q, int ldq) {
  const int mi = mbeg + blockIdx.z;
  const Merge mg = b.merges[mi];
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= mg.n) return;
  const int k = b.k[mi], off = mg.off;
  for (int c = k + blockIdx.y; c < mg.n; c += gridDim.y)
    Q[(size_t)(off + r) + cq(b, off + c) * ldq] = W[(size_t)(off + r) + cw(b, off + c) * ldw];
}

// ------------------------------------------------------------------ final ordering
__global__ void dc_final_rank_kernel(int n, DcBufs b, double *__restrict__ w, int *__restrict__ perm,
                                     int *info) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int t = gid / RP, sub = gid % RP;
  if (t >= n) return;
  const double raw = b.d[t];
  const double dt = (raw == raw) ? raw : INFINITY;
  int rank = 0;
  for (int i = sub; i < n; i += RP) {
    const double di = (b.d[i] == b.d[i]) ? b.d[i] : INFINITY;
    rank += (di < dt || (di == dt && i < t)) ? 1 : 0;
  }
  rank = rank_sum(rank);
  if (sub != 0) return;
  w[rank] = raw * b.orgnrm[0];
  perm[rank] = t;
  if (!(fabs(raw) <= 1.7e308)) atomicMax(info, n + 1);   // NaN / Inf eigenvalue: breakdown
}

// ------------------------------------------------------------------ selected eigenvectors only
__device__ __forceinline__ int sel_rank(const StedcSelect &q, int l) {
  return ((l / q.nb) * q.npcol + q.mycol) * q.nb + l % q.nb;
}

// selcol[l] = basis column (root / deflated column of the top merge) that carries rank r(l);
// thread 0 also narrows the top merge's two GEMMs to nsel output columns.
__global__ void dc_select_kernel(StedcSelect q, const int *__restrict__ fperm, int *__restrict__ selcol,
                                 int *gd) {
  const int l = blockIdx.x * blockDim.x + threadIdx.x;
  if (l < q.nsel) selcol[l] = fperm[sel_rank(q, l)];
  if (l == 0 && gd) { gd[1] = q.nsel; gd[4] = q.nsel; }
}

// Top merge, selected columns: column l of S = the rank-one eigenvector of root selcol[l]
// (as dc_vectors_kernel), or zero when selcol[l] is a deflated column (copied afterwards).
__global__ __launch_bounds__(256) void dc_vectors_sel_kernel(int mi, DcBufs b, const int *__restrict__ selcol,
                                                             double *__restrict__ S, int lds) {
  __shared__ double red[4];
  const Merge mg = b.merges[mi];
  const int l = blockIdx.x, t = threadIdx.x;
  const int k = b.k[mi];
  const int c = selcol[l];
  double *col = S + (size_t)mg.off + (size_t)(mg.off + l) * lds;
  if (c >= k) {
    for (int j = t; j < k; j += 256) col[j] = 0.0;
    return;
  }
  const double *dl = b.dl + mg.off, *zh = b.zhat + mg.off;
  const int *grp = b.grp + mg.off;
  const double dK = dl[b.korig[mg.off + c]], tau = b.tauv[mg.off + c];
  double ss = 0.0;
  for (int j = t; j < k; j += 256) {
    const double u = zh[j] / ((dl[j] - dK) - tau);
    ss += u * u;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) ss += __shfl_down(ss, o, 64);
  if ((t & 63) == 0) red[t >> 6] = ss;
  __syncthreads();
  const double inv = 1.0 / sqrt((red[0] + red[1]) + (red[2] + red[3]));
  for (int j = t; j < k; j += 256) col[grp[j]] = zh[j] / ((dl[j] - dK) - tau) * inv;
}

// selected deflated eigenpairs of the top merge: Q(:, l) = W(:, selcol[l]) where selcol[l] >= k
__global__ void dc_copy_deflated_sel_kernel(int mi, DcBufs b, int nsel, const int *__restrict__ selcol,
                                            const double *__restrict__ W, int ldw, double *__restrict__ Q,
                                            int ldq) {
  const Merge mg = b.merges[mi];
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= mg.n) return;
  const int k = b.k[mi];
  for (int l = blockIdx.y; l < nsel; l += gridDim.y) {
    const int c = selcol[l];
    if (c >= k) Q[(size_t)r + (size_t)l * ldq] = W[(size_t)r + (size_t)c * ldw];
  }
}

inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

struct Plan {
  std::vector<Leaf> leaves;
  std::vector<std::vector<Merge>> levels;   // by height
  int height(int off, int n) {
    if (n <= LEAF) { leaves.push_back({off, n}); return 0; }
    const int n1 = n / 2;
    const int h1 = height(off, n1), h2 = height(off + n1, n - n1);
    const int h = std::max(h1, h2) + 1;
    if ((int)levels.size() < h) levels.resize(h);
    levels[h - 1].push_back({off, n1, n, 0});
    return h;
  }
};

// nsel: eigenvector columns the caller wants (-1 or n: all).  With at most half of them (a grid cell's share of a team of
// two or more, a *_select arm) the bases are kept COMPACT: below the top merge every block lies inside one half of the
// matrix, so a level fits n x h2 (h2 = n - n/2) with the second half's blocks stored n/2 columns to the left (DcBufs::hq,
// hw, hs); the top merge needs the permuted basis at full width once, where the lower levels' W and S used to be, and
// its S and result have nsel columns: 1.5 n^2 + n nsel doubles instead of the 3 n^2 of the full form with a separate
// scratch for the permuted bases.
struct WorkLayout {
  size_t off_Q, off_S, off_Ssel, off_vec, off_int, off_merge, off_leaf, off_offs, off_dims, off_eoffs, off_edims, total;
  int nmerge_cap, nleaf_cap, nentry_cap;
  bool compact;
  int h2;
  explicit WorkLayout(int n, int nsel = -1) {
    nleaf_cap = n / (LEAF / 2) + 2; nmerge_cap = nleaf_cap;
    h2 = n - n / 2;
    compact = nsel >= 0 && nsel < n && nsel <= h2 && n > LEAF;
    size_t o = 0;
    off_Q = o; o += al256((size_t)n * (compact ? h2 : n) * 8);
    off_S = o; o += al256((size_t)n * (compact ? 2 * h2 : n) * 8);      // compact: [W | S] of a lower level, W of the top merge
    off_Ssel = o; o += compact ? al256((size_t)n * (nsel > 0 ? nsel : 1) * 8) : 0;
    off_vec = o; o += 12 * al256((size_t)(n + 8) * 8) + al256((size_t)nmerge_cap * 8) + 256;
    off_int = o; o += 8 * al256((size_t)(n + 8) * 4) + 2 * al256((size_t)nmerge_cap * 4);
    off_merge = o; o += al256((size_t)nmerge_cap * sizeof(Merge));
    off_leaf = o; o += al256((size_t)nleaf_cap * sizeof(Leaf));
    off_offs = o; o += al256((size_t)nmerge_cap * 6 * 8);
    off_dims = o; o += al256((size_t)nmerge_cap * 6 * 4);
    nentry_cap = 4 * (ceil_div(h2, kStrip) + 64);            // team form: pieces of one rank at one height (both products)
    off_eoffs = o; o += al256((size_t)nentry_cap * 3 * 8);
    off_edims = o; o += al256((size_t)nentry_cap * 3 * 4);
    total = o;
  }
};

}  // namespace

size_t stedc_work_bytes(int n, int nsel) { return WorkLayout(n > 0 ? n : 1, nsel).total; }
bool stedc_compact(int n, int nsel) { return WorkLayout(n > 0 ? n : 1, nsel).compact; }

namespace {
// flops of the eigenvector products this solve really ran: 2 M N K over the two GEMMs of every merge, with the
// dimensions the deflation (and the column selection of the top merge) left on the device
__global__ void dc_flops_kernel(int nmerge, const int *__restrict__ gdims, double *out) {
  __shared__ double s_sum[256];
  double a = 0.0;
  for (int i = threadIdx.x; i < 2 * nmerge; i += 256)
    a += 2.0 * (double)gdims[3 * i] * (double)gdims[3 * i + 1] * (double)gdims[3 * i + 2];
  s_sum[threadIdx.x] = a;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) s_sum[threadIdx.x] += s_sum[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) *out = s_sum[0];
}
}  // namespace

// ---- team form: which heights, and the rehearsal's clock
namespace {
int g_team_levels = -1;
struct TeamProfile {
  bool on = false;
  hipEvent_t call[2] = {nullptr, nullptr};
  bool have_call = false;
  struct Sec { hipEvent_t e0, e1; int level, rank; };
  std::vector<Sec> secs;
  void clear() {
    for (auto &q : secs) { (void)hipEventDestroy(q.e0); (void)hipEventDestroy(q.e1); }
    secs.clear();
    if (call[0]) (void)hipEventDestroy(call[0]);
    if (call[1]) (void)hipEventDestroy(call[1]);
    call[0] = call[1] = nullptr; have_call = false;
  }
} g_tprof;
}  // namespace

// Two heights below the top merge from order 8192 on.  The top merge is 3/4 of the D&C's products and already forms the
// rank's own columns only; heights 2 and 3 are 3/16 and 3/64, everything below them 1/64 (1.6 % of 4/3 n^3: replicated,
// it costs a rank of 8 less than the two exchanges of a further height would).  A sharded height moves 8 n (n - n/2)
// bytes per rank whatever its merges' order (whole columns of the compact array), so going deeper buys less and less.
int stedc_team_levels(int n, int nranks) {
  if (g_team_levels >= 0) return nranks >= 2 ? g_team_levels : 0;
  return (nranks >= 2 && n >= 8192) ? 2 : 0;
}
void stedc_team_set_levels(int levels) { g_team_levels = levels; }
void stedc_team_profile(bool on) { g_tprof.clear(); g_tprof.on = on; }
void stedc_team_profile_collect(double *seconds) {
  seconds[0] = seconds[1] = seconds[2] = 0.0;
  float ms = 0.f;
  if (g_tprof.have_call && hipEventElapsedTime(&ms, g_tprof.call[0], g_tprof.call[1]) == hipSuccess) seconds[0] = ms * 1e-3;
  std::vector<double> longest;
  for (auto &q : g_tprof.secs) {
    if (hipEventElapsedTime(&ms, q.e0, q.e1) != hipSuccess) continue;
    seconds[1] += ms * 1e-3;
    if ((int)longest.size() <= q.level) longest.resize(q.level + 1, 0.0);
    if (ms * 1e-3 > longest[q.level]) longest[q.level] = ms * 1e-3;
  }
  for (double v : longest) seconds[2] += v;
  const bool on = g_tprof.on;
  g_tprof.clear(); g_tprof.on = on;
}

void stedc(hipStream_t s, int n, const double *d, const double *e, double *w, double *Z, int ldz,
           void *work, int *d_info, const StedcSelect *sel, double *d_flops, double *wscratch, const StedcTeam *team) {
  if (n <= 0) return;
  const WorkLayout L(n, sel ? sel->nsel : -1);
  char *base = (char *)work;
  double *Q = (double *)(base + L.off_Q);      // current eigenvector basis (ld = n)
  double *S = (double *)(base + L.off_S);      // rank-one eigenvector matrices (ld = n)
  const int ldq = n, lds = n;
  const bool compact = L.compact;
  const int half = n / 2;                      // n1 of the top merge (Plan::height)
  double *Wbig = S;                            // compact: the permuted bases live in the workspace (ld = n)
  if (compact) S = Wbig + (size_t)n * L.h2;
  const int ldw = compact ? n : ldz;
  DcBufs b;
  {
    char *p = base + L.off_vec;
    auto dv = [&](size_t cnt) { double *r = (double *)p; p += al256(cnt * 8); return r; };
    b.d = dv(n + 8); b.e = dv(n + 8); b.dsort = dv(n + 8); b.zsort = dv(n + 8); b.dl = dv(n + 8);
    b.zl = dv(n + 8); b.zhat = dv(n + 8); b.tauv = dv(n + 8); b.rotc = dv(n + 8); b.rots = dv(n + 8);
    b.kd = dv(n + 8);
    double *spare2 = dv(n + 8); (void)spare2;
    b.rho = dv(L.nmerge_cap); b.orgnrm = (double *)p;
    char *q = base + L.off_int;
    auto iv = [&](size_t cnt) { int *r = (int *)q; q += al256(cnt * 4); return r; };
    b.korig = iv(n + 8); b.perm = iv(n + 8); b.wcol = iv(n + 8); b.rotp = iv(n + 8); b.rotn = iv(n + 8);
    int *fperm = iv(n + 8); (void)fperm;
    b.grp = iv(n + 8); b.spos = iv(n + 8);
    b.k = iv(L.nmerge_cap); b.nrot = iv(L.nmerge_cap);
    b.merges = (Merge *)(base + L.off_merge);
    b.goffs = (long long *)(base + L.off_offs);
    b.gdims = (int *)(base + L.off_dims);
    b.ldw = ldw; b.ldq = ldq; b.lds = lds;
    if (compact) b.hq = b.hw = b.hs = half;
  }
  int *fperm = (int *)(base + L.off_int + 5 * al256((size_t)(n + 8) * 4));
  Leaf *d_leaves = nullptr;

  // The recursion tree depends on n only: it is built and uploaded once per order and kept in
  // a small persistent device buffer, so a solve never synchronises with the host here.
  struct CachedPlan {
    int n = -1;
    Plan plan;
    std::vector<Merge> all;
    std::vector<int> lvl_beg;
    Leaf *d_leaves = nullptr;
    Merge *d_merges = nullptr;
  };
  static CachedPlan cache;
  if (cache.n != n) {
    (void)hipStreamSynchronize(s);   // a previous order's plan may still be in use on the stream
    if (cache.d_leaves) (void)hipFree(cache.d_leaves);
    if (cache.d_merges) (void)hipFree(cache.d_merges);
    cache = CachedPlan();
    cache.plan.height(0, n);
    for (auto &lv : cache.plan.levels) {
      cache.lvl_beg.push_back((int)cache.all.size());
      for (auto &m : lv) cache.all.push_back(m);
    }
    cache.lvl_beg.push_back((int)cache.all.size());
    (void)hipMalloc((void **)&cache.d_leaves, cache.plan.leaves.size() * sizeof(Leaf) + 64);
    (void)hipMalloc((void **)&cache.d_merges, cache.all.size() * sizeof(Merge) + 64);
    (void)hipMemcpy(cache.d_leaves, cache.plan.leaves.data(), cache.plan.leaves.size() * sizeof(Leaf),
                    hipMemcpyHostToDevice);
    if (!cache.all.empty())
      (void)hipMemcpy(cache.d_merges, cache.all.data(), cache.all.size() * sizeof(Merge), hipMemcpyHostToDevice);
    cache.n = n;
  }
  const Plan &plan = cache.plan;
  const std::vector<Merge> &all = cache.all;
  const std::vector<int> &lvl_beg = cache.lvl_beg;
  d_leaves = cache.d_leaves;
  b.merges = cache.d_merges;

  // Team form: the heights right below the top merge whose products are cut into strips (compact bases only), and the
  // pieces of every rank at every such height (a function of the order and the team: built and uploaded once).
  const int nlev = (int)plan.levels.size();
  int tlevels = (team && compact && team->nranks >= 2) ? team->levels : 0;
  if (tlevels > nlev - 1) tlevels = nlev - 1;
  while (tlevels > 0 && (int)plan.levels[nlev - 1 - tlevels].size() > 32) --tlevels;
  struct TeamPlan {
    int n = -1, P = 0, levels = 0;
    std::vector<int> beg;            // [(height index from the top - 1) * P + rank] -> first piece; one more at the end
    TeamEntry *d_tab = nullptr;
  };
  static TeamPlan tplan;
  const int tP = tlevels > 0 ? team->nranks : 0;
  if (tlevels > 0 && (tplan.n != n || tplan.P != tP || tplan.levels != tlevels)) {
    (void)hipStreamSynchronize(s);
    if (tplan.d_tab) (void)hipFree(tplan.d_tab);
    tplan = TeamPlan();
    std::vector<TeamEntry> tab;
    for (int t = 1; t <= tlevels; ++t) {
      const int lv = nlev - 1 - t, mbeg = lvl_beg[lv];
      std::vector<std::vector<TeamEntry>> per(tP);
      for (size_t i = 0; i < plan.levels[lv].size(); ++i) {
        const Merge &m = plan.levels[lv][i];
        const int q0 = m.off >= half ? m.off - half : m.off;            // first column of the block as stored
        for (int S = q0 / kStrip; S * kStrip < q0 + m.n; ++S) {
          const int lo = std::max(q0, S * kStrip), hi = std::min(q0 + m.n, (S + 1) * kStrip);
          for (int pr = 0; pr < 2; ++pr) per[S % tP].push_back({mbeg + (int)i, pr, lo - q0, hi - lo});
        }
      }
      for (int r = 0; r < tP; ++r) {
        tplan.beg.push_back((int)tab.size());
        tab.insert(tab.end(), per[r].begin(), per[r].end());
      }
    }
    tplan.beg.push_back((int)tab.size());
    (void)hipMalloc((void **)&tplan.d_tab, tab.size() * sizeof(TeamEntry) + 64);
    (void)hipMemcpy(tplan.d_tab, tab.data(), tab.size() * sizeof(TeamEntry), hipMemcpyHostToDevice);
    tplan.n = n; tplan.P = tP; tplan.levels = tlevels;
  }
  long long *eoffs = (long long *)(base + L.off_eoffs);
  int *edims = (int *)(base + L.off_edims);
  const bool tprof = g_tprof.on && tlevels > 0;
  if (tprof) {
    g_tprof.clear(); g_tprof.on = true;
    (void)hipEventCreate(&g_tprof.call[0]); (void)hipEventCreate(&g_tprof.call[1]);
    (void)hipEventRecord(g_tprof.call[0], s);
  }

  hipLaunchKernelGGL(dc_scale_kernel, dim3(1), dim3(256), 0, s, n, d, e, b);
  if (!all.empty())
    hipLaunchKernelGGL(dc_split_kernel, dim3(ceil_div((int)all.size(), 256)), dim3(256), 0, s,
                       (int)all.size(), b);
  (void)hipMemsetAsync(Q, 0, (size_t)n * (compact ? L.h2 : n) * 8, s);
  hipLaunchKernelGGL(dc_leaf_kernel, dim3((int)plan.leaves.size()), dim3(64), 0, s, d_leaves, b, Q, ldq,
                     d_info);

  auto count_flops = [&]() {
    if (d_flops) hipLaunchKernelGGL(dc_flops_kernel, dim3(1), dim3(256), 0, s, (int)all.size(), b.gdims, d_flops);
  };
  double *W = compact ? Wbig : (wscratch ? wscratch : Z);   // (full form: the output array doubles as the permuted-basis scratch until the end)
  const bool selecting = sel && sel->nsel < n;
  bool sel_done = false;
  for (size_t lv = 0; lv < plan.levels.size(); ++lv) {
    const int mbeg = lvl_beg[lv], cnt = lvl_beg[lv + 1] - mbeg;
    int maxn = 0;
    bool even = ((ldw | ldq | lds) & 1) == 0;      // every merge of this height starts on even rows and columns
    for (auto &m : plan.levels[lv]) { maxn = std::max(maxn, m.n); even = even && ((m.off | m.n1) & 1) == 0; }
    const int gx = ceil_div(maxn, 256);
    const bool top = lv + 1 == plan.levels.size();
    if (compact && top) {                 // the top merge: W at full width, S (n x nsel) in its own array; Q stays compact
      b.hw = b.hs = 0x7fffffff;
      S = (double *)(base + L.off_Ssel);
    }
    hipLaunchKernelGGL(dc_sort_kernel, dim3(ceil_div(maxn * RP, 256), cnt), dim3(256), 0, s, mbeg, b, Q, ldq);
    hipLaunchKernelGGL(dc_deflate_kernel, dim3(cnt), dim3(256), 0, s, mbeg, b);
    const int gy = std::min(maxn, std::max(1, 4096 / std::max(1, gx * cnt)));
    hipLaunchKernelGGL(dc_permute_kernel, dim3(gx, gy, cnt), dim3(256), 0, s, mbeg, b, Q, ldq, W, ldw);
    hipLaunchKernelGGL(dc_rotate_kernel, dim3(gx, cnt), dim3(256), 0, s, mbeg, b, W, ldw);
    const int tdepth = nlev - 1 - (int)lv;            // 0: the top merge, 1: right below it
    const bool tshard = tlevels > 0 && tdepth <= tlevels;
    if (tshard) {
      // Team form: the roots (then the Loewner weights) of every merge of this height in P equal runs, a run per rank; the
      // three vectors (shift, origin, new eigenvalue -- then the weights) change hands in all-gathers of n_m doubles per
      // merge.  Same arithmetic per root: same bits.
      const int tc = round_up(ceil_div(maxn, tP), 2);
      const int r0 = team->rank >= 0 ? team->rank : 0, r1 = team->rank >= 0 ? team->rank + 1 : tP;
      auto exchange = [&](double *vec) {
        if (team->rank < 0) return;
        for (auto &m : plan.levels[lv]) {
          size_t offs[kMaxTeam], counts[kMaxTeam];
          for (int r = 0; r < tP; ++r) {
            const int a = std::min(r * tc, m.n), e2 = std::min((r + 1) * tc, m.n);
            offs[r] = (size_t)m.off + a; counts[r] = (size_t)(e2 - a);
          }
          double *bufs[1] = {vec};
          team->x->allgatherv(s, 1, team->rank, bufs, offs, counts, tP, team->x->user);
        }
      };
      for (int pass = 0; pass < 2; ++pass) {
        for (int r = r0; r < r1; ++r) {
          hipEvent_t e0 = nullptr, e1 = nullptr;
          if (tprof && team->rank < 0) { (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); (void)hipEventRecord(e0, s); }
          b.tP = tP; b.tR = r; b.tc = tc;
          if (pass == 0) hipLaunchKernelGGL(dc_secular_kernel, dim3(ceil_div(maxn * SP, 256), cnt), dim3(256), 0, s, mbeg, b);
          else hipLaunchKernelGGL(dc_zhat_kernel, dim3(ceil_div(maxn * SP, 256), cnt), dim3(256), 0, s, mbeg, b);
          b.tP = 0;
          if (e0) { (void)hipEventRecord(e1, s); g_tprof.secs.push_back({e0, e1, 64 + 2 * tdepth + pass, r}); }
        }
        if (pass == 0) {
          exchange(b.tauv); exchange(b.kd); exchange(b.d);
          if (team->rank >= 0) hipLaunchKernelGGL(dc_korig_kernel, dim3(ceil_div(maxn, 256), cnt), dim3(256), 0, s, mbeg, b);
        } else {
          exchange(b.zhat);
        }
      }
    } else {
      hipLaunchKernelGGL(dc_secular_kernel, dim3(ceil_div(maxn * SP, 256), cnt), dim3(256), 0, s, mbeg, b);
      hipLaunchKernelGGL(dc_zhat_kernel, dim3(ceil_div(maxn * SP, 256), cnt), dim3(256), 0, s, mbeg, b);
    }
    if (selecting && lv + 1 == plan.levels.size()) {
      // Top merge (off = 0, order n): every eigenvalue is known now, so the final order can be
      // fixed before the eigenvector product and only the selected columns are multiplied,
      // straight into their final positions.
      int *selcol = b.rotp;               // the rotation list is spent once dc_rotate has run
      hipLaunchKernelGGL(dc_final_rank_kernel, dim3(ceil_div(n * RP, 256)), dim3(256), 0, s, n, b, w, fperm, d_info);
      sel_done = true;
      if (sel->nsel <= 0) break;
      hipLaunchKernelGGL(dc_select_kernel, dim3(ceil_div(sel->nsel, 256)), dim3(256), 0, s, *sel, fperm, selcol,
                         b.gdims + 6 * (size_t)mbeg);
      hipLaunchKernelGGL(dc_vectors_sel_kernel, dim3(sel->nsel), dim3(256), 0, s, mbeg, b, selcol, S, lds);
      GemmDesc g{};
      g.M = (maxn + 1) / 2; g.N = sel->nsel; g.K = maxn; g.transA = false; g.transB = false;
      g.alpha = 1.0; g.beta = 0.0;
      g.A = W; g.lda = ldw; g.strideA = 0; g.B = S; g.ldb = lds; g.strideB = 0;
      g.C = Q; g.ldc = ldq; g.strideC = 0; g.batch = 2; g.lower_only = false;
      g.d_offs = b.goffs + 6 * (size_t)mbeg; g.d_dims = b.gdims + 6 * (size_t)mbeg; g.even_offs = even;
      gemm(s, g);
      const int gys = std::min(sel->nsel, std::max(1, 4096 / gx));
      hipLaunchKernelGGL(dc_copy_deflated_sel_kernel, dim3(gx, gys), dim3(256), 0, s, mbeg, b, sel->nsel, selcol,
                         W, ldw, Q, ldq);
      copy_matrix(s, n, sel->nsel, Q, ldq, Z, ldz);
      break;
    }
    if (tlevels > 0 && tdepth >= 1 && tdepth <= tlevels) {
      // team form of this height (StedcTeam): S and the products of a rank's own strips only, then the strips change hands
      const int r0 = team->rank >= 0 ? team->rank : 0, r1 = team->rank >= 0 ? team->rank + 1 : tP;
      for (int r = r0; r < r1; ++r) {
        const int ib = tplan.beg[(size_t)(tdepth - 1) * tP + r], ie = tplan.beg[(size_t)(tdepth - 1) * tP + r + 1];
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (tprof && team->rank < 0) { (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); (void)hipEventRecord(e0, s); }
        b.tP = tP; b.tR = r;
        hipLaunchKernelGGL(dc_vectors_kernel, dim3(maxn, cnt), dim3(256), 0, s, mbeg, b, S, lds);
        b.tP = 0;
        if (ie > ib) {
          hipLaunchKernelGGL(dc_team_entries_kernel, dim3(ceil_div(ie - ib, 256)), dim3(256), 0, s, tplan.d_tab + ib, ie - ib,
                             b, eoffs, edims);
          GemmDesc g{};
          g.M = (maxn + 1) / 2; g.N = kStrip; g.K = maxn; g.transA = false; g.transB = false;
          g.alpha = 1.0; g.beta = 0.0;
          g.A = W; g.lda = ldw; g.strideA = 0; g.B = S; g.ldb = lds; g.strideB = 0;
          g.C = Q; g.ldc = ldq; g.strideC = 0; g.batch = ie - ib; g.lower_only = false;
          g.d_offs = eoffs; g.d_dims = edims; g.even_offs = even;
          gemm(s, g);
        }
        if (e0) { (void)hipEventRecord(e1, s); g_tprof.secs.push_back({e0, e1, tdepth, r}); }
      }
      // (the deflated columns are copies of the replicated W: every rank writes all of them)
      hipLaunchKernelGGL(dc_copy_deflated_kernel, dim3(gx, gy, cnt), dim3(256), 0, s, mbeg, b, W, ldw, Q, ldq);
      if (team->rank >= 0) {
        // one in-place all-gather per round of P strips: whole columns of the compact array (ld = n, so a strip is one
        // contiguous piece and a round's pieces lie in rank order: ncclAllGather's own layout)
        const int NS = ceil_div(L.h2, kStrip);
        double *bufs[1] = {Q};
        for (int q = 0; q * tP < NS; ++q) {
          size_t offs[kMaxTeam], counts[kMaxTeam];
          for (int r = 0; r < tP; ++r) {
            const int Sx = q * tP + r;
            const int cols = Sx < NS ? std::min(kStrip, L.h2 - Sx * kStrip) : 0;
            offs[r] = (size_t)std::min(Sx, NS) * kStrip * ldq; counts[r] = (size_t)cols * ldq;
          }
          team->x->allgatherv(s, 1, team->rank, bufs, offs, counts, tP, team->x->user);
        }
      }
      continue;
    }
    hipLaunchKernelGGL(dc_vectors_kernel, dim3(maxn, cnt), dim3(256), 0, s, mbeg, b, S, lds);
    // two GEMMs per merge (top rows x [top-only|dense] columns, bottom rows x [dense|bottom-only]
    // columns), all merges of this height in one launch; sizes and offsets come from the
    // deflation kernel, so nothing returns to the host
    GemmDesc g{};
    g.M = (maxn + 1) / 2; g.N = maxn; g.K = maxn; g.transA = false; g.transB = false;
    g.alpha = 1.0; g.beta = 0.0;
    g.A = W; g.lda = ldw; g.strideA = 0; g.B = S; g.ldb = lds; g.strideB = 0;
    g.C = Q; g.ldc = ldq; g.strideC = 0; g.batch = 2 * cnt; g.lower_only = false;
    g.d_offs = b.goffs + 6 * (size_t)mbeg; g.d_dims = b.gdims + 6 * (size_t)mbeg; g.even_offs = even;
    gemm(s, g);
    hipLaunchKernelGGL(dc_copy_deflated_kernel, dim3(gx, gy, cnt), dim3(256), 0, s, mbeg, b, W, ldw, Q, ldq);
  }
  count_flops();
  if (tprof) { (void)hipEventRecord(g_tprof.call[1], s); g_tprof.have_call = true; }   // (the final gather of a full spectrum is not a team's)
  if (sel_done) return;
  hipLaunchKernelGGL(dc_final_rank_kernel, dim3(ceil_div(n * RP, 256)), dim3(256), 0, s, n, b, w, fperm, d_info);
  if (selecting) {
    // no merge at all (a single leaf): pick the selected columns out of the leaf's eigenvectors
    if (sel->nsel <= 0) return;
    int *selcol = b.rotp;
    hipLaunchKernelGGL(dc_select_kernel, dim3(ceil_div(sel->nsel, 256)), dim3(256), 0, s, *sel, fperm, selcol,
                       (int *)nullptr);
    gather_columns(s, n, sel->nsel, Q, ldq, selcol, Z, ldz);
    return;
  }
  gather_columns(s, n, n, Q, ldq, fperm, Z, ldz);
}

}  // namespace ek
