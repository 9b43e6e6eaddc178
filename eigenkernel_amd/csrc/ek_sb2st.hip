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
// of the band.  Task (s, k): reflector k of sweep s on I_k = [s+1+64k, s+64(k+1)], applied from both
// sides to D_k = A(I_k, I_k) and from the right to B_k = A(I_{k+1}, I_k); then reflector k+1 is made from
// column 0 of the new B_k and applied to B_k from the left, so that the task hands on a finished block.
// Task (s+1, j) may start once task (s, j) is complete and receives 65 late numbers of task (s, j+1)
// through a mailbox (see chase_kernel): the sweeps form a pipeline one task apart.
//
// MI355X shape: ONE persistent launch; a workgroup takes sweeps from a ticket counter (in order, so
// a workgroup only ever waits for a sweep whose owner is already running) and walks down the band (row
// per lane, 8 columns per wave).  Sweeps synchronise through one progress word per sweep.  The band
// lives in L2 / Infinity Cache (16 MB at
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
#include <type_traits>
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
// Sums across lanes with DPP moves (a few cycles each) instead of __shfl_xor (an LDS-crossbar round trip
// per step): in the latency-bound chase every such chain is on the critical path of a task.
template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
// sum over aligned groups of W consecutive lanes (W = 2, 4, 8, 16), result in every lane of the group
template <int W>
__device__ __forceinline__ double group_sum(double v) {
  if (W >= 2) v += dpp_mov<0xB1>(v);       // quad_perm [1,0,3,2]
  if (W >= 4) v += dpp_mov<0x4E>(v);       // quad_perm [2,3,0,1]
  if (W >= 8) v += dpp_mov<0x141>(v);      // row_half_mirror
  if (W >= 16) v += dpp_mov<0x140>(v);     // row_mirror
  return v;
}
__device__ __forceinline__ double lane_value(double v, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}
// sum over the wave, result in every lane.  The four row sums meet by two DPP row broadcasts (lane 15 of a row
// into the next row, lane 31 into rows 2 and 3) and ONE v_readlane pair of lane 63 -- a pair costs about 40
// cycles and four of them do not overlap (measured in ek_block64.h); the value is (r2 + r3) + (r0 + r1), the
// same bits as (r0 + r1) + (r2 + r3).
template <int CTRL, int ROWMASK>
__device__ __forceinline__ double dpp_mov_rows(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROWMASK, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROWMASK, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum(double v) {
  v = group_sum<16>(v);
  v += dpp_mov_rows<0x142, 0xA>(v);        // row_bcast15: rows 1 and 3 += the sum of the row before
  v += dpp_mov_rows<0x143, 0xC>(v);        // row_bcast31: rows 2 and 3 += r0 + r1
  return lane_value(v, 63);
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
  unsigned *ctl;         // [0] ticket, [1] abort, [3] "the position-owned kernel gave up" (this kernel then runs), [4] census
  double *mail; int kmax;   // mailbox lines [4][kmax][MAILW]
  long long *prof;       // optional: [0..5] shader cycles per phase of a task summed over workgroup 0's tasks, [6] tasks
  int only_if_abandoned; // run only if ctl[3] is set (the fall-back behind chase_pos_kernel)
  const int *skip;       // device: non-zero = the band is not valid (the first stage raised its flag): do nothing
};

constexpr unsigned kSpinLimit = 1u << 22;

// ------------------------------------------------------------------------ the arithmetic of a task
// Shared by the two kernels below, which differ only in WHERE the blocks live and how the sweeps hand them
// on; with contraction off and every fused operation written out, both produce the same bits.
// Layout of a task's blocks: lane = row of the block, wave w holds columns 8 w .. 8 w + 7 (CW per wave).
struct Reflector { double v, tau, beta; };

// One wave: DLARFG on x (lane = row, x = 0 beyond the block): v (v_0 = 1), tau, beta in every lane.
__device__ __forceinline__ Reflector reflector_of(double x, int lane) {
#pragma clang fp contract(off)
  const double ssq = wave_sum((lane >= 1) ? x * x : 0.0);
  const double alpha0 = lane_value(x, 0);
  Reflector r;
  r.beta = alpha0; r.tau = 0.0;
  double scale = 0.0;
  // The path keeps |A| within 1e+-90, but the fill of a NARROW band decays geometrically across the 64 diagonals: a
  // column of a bulge can consist of entries around 1e-160, whose squares are denormal -- a norm with three
  // significant bits, and a "reflector" that is not orthogonal in the third digit (seen: a bandwidth-5 input of order
  // 5000 lost its spectrum at the 1e-2 level).  A column whose squared norm is below 1e-290 is 1e-55 of the smallest
  // matrix the path admits: it is dropped (H = I; its entries below the first are the zeros the caller stores anyway).
  const double nrm2 = __builtin_fma(alpha0, alpha0, ssq);
  if (ssq != 0.0 && nrm2 > 1e-290) {
    r.beta = -copysign(sqrt(nrm2), alpha0);
    r.tau = (r.beta - alpha0) / r.beta;
    scale = 1.0 / (alpha0 - r.beta);
  }
  r.v = (lane == 0) ? 1.0 : x * scale;    // rows beyond the block carry x = 0
  return r;
}
// partial sums of p = D v and q = B v over this wave's columns
template <int CW>
__device__ __forceinline__ void task_partials(const double (&dd)[CW], const double (&bk)[CW], const double (&vc)[CW],
                                              double &pp, double &qq) {
#pragma clang fp contract(off)
  pp = 0.0; qq = 0.0;
#pragma unroll
  for (int j = 0; j < CW; ++j) { pp = __builtin_fma(dd[j], vc[j], pp); qq = __builtin_fma(bk[j], vc[j], qq); }
}
// column 0 of B H (rows = lanes), from q = B v summed over the waves
__device__ __forceinline__ double task_col0(double bk0, double tau, double qs, double vc0) {
#pragma clang fp contract(off)
  const double t = tau * qs;
  return __builtin_fma(-t, vc0, bk0);
}
// w = tau p + alpha v with alpha = -tau/2 (p^T v) tau ... the vector of the symmetric rank-2 update D - v w^T - w v^T
__device__ __forceinline__ double task_w(double tau, double psum, double v_r) {
#pragma clang fp contract(off)
  const double p_r = tau * psum;
  const double dot = wave_sum(p_r * v_r);
  const double alpha = (-0.5 * tau) * dot;
  return __builtin_fma(alpha, v_r, p_r);
}
// D(r, c) -= v_r w_c + w_r v_c, written so that entries (r, c) and (c, r) get the same bits: a block that stays in
// registers as a full image stays exactly symmetric, and a block whose lower triangle travels through memory and
// is mirrored gets the same numbers.
__device__ __forceinline__ double task_rank2(double d, double v_r, double w_c, double w_r, double v_c) {
#pragma clang fp contract(off)
  const double t1 = v_r * w_c, t2 = w_r * v_c;
  return d - (t1 + t2);
}
__device__ __forceinline__ double task_right(double b, double q_r, double v_c) {
#pragma clang fp contract(off)
  return __builtin_fma(-q_r, v_c, b);
}
// B <- H' B with the new reflector (vn, tau_n), rows = lanes: v'^T B by a transposing reduction through the wave's
// LDS buffer st (CW x 65 doubles); column 0 of the block (c == 0) is left alone: it is (beta, 0, ..., 0).
template <int CW>
__device__ __forceinline__ void task_left(double (&bp)[CW], double vn_r, double tau_n, double *st, int lane, bool skip0) {
#pragma clang fp contract(off)
  constexpr int LPC = 64 / CW;
#pragma unroll
  for (int j = 0; j < CW; ++j) st[j * 65 + lane] = vn_r * bp[j];
  wave_sync();
  double tot_b;
  {
    const int j = lane / LPC, q = lane % LPC;
    const double *src = st + j * 65 + CW * q;
    double a0 = 0.0, a1 = 0.0;
#pragma unroll
    for (int l = 0; l < CW; l += 2) { a0 += src[l]; a1 += src[l + 1]; }
    tot_b = group_sum<LPC>(a0 + a1);                 // lanes j LPC .. : v'^T (column j of this wave's part of B)
  }
  const double u = tau_n * vn_r;
#pragma unroll
  for (int j = 0; j < CW; ++j)
    if (!(skip0 && j == 0)) bp[j] = __builtin_fma(-u, lane_value(tot_b, j * LPC), bp[j]);
  wave_sync();
}

// Dependencies between sweeps.  Task (s, k) works on rows and columns s+1+64k .. s+64(k+2) of the lower
// band.  The textbook rule is "task (s, k) after task (s-1, k+2)"; what is really needed is less:
//  * of task (s-1, k+2): one entry, the corner A(s+64(k+2), s+64(k+1)) -- entry (0,0) of ITS block
//    B_{k+1}, which its reflector turns into beta, a number known when that reflector is made, at the end
//    of task (s-1, k+1);
//  * of task (s-1, k+1): column 0 of ITS diagonal block (our last column of B_k and the corner of D_k) and
//    that beta -- 65 numbers, and only when our own block images are being put together;
//  * everything else is final when task (s-1, k) ends, PROVIDED a task finishes the block it hands on:
//    so a task applies the NEXT reflector from the left to the new B_k at its own end (it holds both),
//    instead of the next task doing that first.
//
// TWO kernels are built on that.
//
// chase_pos_kernel (the default; orders up to 64 x the number of workgroups the chip holds at once):
// a workgroup owns a POSITION k of the band and keeps D_k and B_k in its REGISTERS for the whole stage
// (lane = row, 8 columns per wave); sweep after sweep passes through it.  From one sweep to the next the blocks
// move one row and one column down the band: the images shift by one (DPP wave shift for the rows, a register
// rename and one LDS exchange for the columns), the new last row of D_k is the old first row of B_k, and the
// new last column of B_k and the corner of D_k are the 65 late numbers of position k+1.  Nothing but the
// reflectors (for Q2), d and e ever goes to memory; between positions travel two self-flagging mailbox lines per
// sweep: the reflector (64 + tau) forward to k+1, the late numbers (64 + beta) back to k-1.  The stage is
// the cycle "reflector of (s, k+1) -> late numbers of (s, k+1) -> reflector of (s+1, k+1)": two hops across
// the chip and the two short chains between them per sweep (tools/chain_hop.hip: 2.5 us for the bare cycle),
// instead of a whole task plus fetch, drain and progress word (7.9 us).
//
// chase_kernel (the older form; any order, no co-residency needed; also the fall-back if the positions could
// not all become resident): a workgroup takes SWEEPS from a ticket counter and walks down the band, blocks
// travel through memory.  Task (s, k) starts when task (s-1, k) is complete (one progress word per sweep); the 65
// late numbers of task (s-1, k+1) come through a MAILBOX line its last wave polls directly (an "empty" bit
// pattern marks a slot; the reader empties it again; four lines per task index in rotation) -- data that
// is its own flag costs no drain, no barrier and no second round trip.  Sweeps follow each other ONE task
// apart (plus the hand-off), not three; the pipeline is latency-bound and its length is the number of
// sweeps times that distance.  The 65 mailbox numbers are exactly the entries BOTH sweeps would store (the
// follower rewrites them in the same task): the sender leaves them to the follower, so no store of one
// sweep ever has to wait for a store of another.
//
// Per task the workgroup synchronises four times.  NW waves per workgroup, each with CW = 64 / NW
// columns of a block (row per lane).  Both kernels run the arithmetic above: same bits.
constexpr int MAILW = 72;                              // doubles per mailbox line: 64 (column 0 of D_k) + beta + padding
constexpr unsigned long long kMailEmpty = 0x7ff8dead0000beefull;   // a NaN no computation produces

__device__ __forceinline__ bool mail_empty(double v) { return (unsigned long long)__double_as_longlong(v) == kMailEmpty; }

__global__ void mail_init_kernel(double *mail, int count) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) mail[i] = __longlong_as_double((long long)kMailEmpty);
}

template <int NW>
__global__ __launch_bounds__(64 * NW) void chase_kernel(ChaseArgs p) {
  constexpr int CW = SB / NW;            // columns of a block per wave
  __shared__ __attribute__((aligned(16))) double s_v[2][SB];
  __shared__ double s_p[NW][SB], s_q[NW][SB];
  __shared__ double s_t[NW][CW * 65];
  __shared__ double s_D[SB * DLD];
  __shared__ double s_tau[2];
  __shared__ int s_sweep, s_ok;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int n = p.n;
  const int c0w = CW * wave;                               // this wave's columns of a block
  double *AB = p.AB;
  if (p.only_if_abandoned && !__hip_atomic_load(&p.ctl[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
  if (p.skip && (*p.skip & 0xff)) return;
  if (t == 0) s_ok = 1;
  auto give_up = [&]() { __hip_atomic_store(&p.ctl[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
  // bounded wait until *w >= need
  auto wait_word = [&](const unsigned *w, unsigned need) -> bool {
    unsigned spins = 0;
    while (__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
      __builtin_amdgcn_s_sleep(1);
      if ((++spins & 63u) == 0u &&
          (spins > kSpinLimit || __hip_atomic_load(&p.ctl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)))
        return false;
    }
    return true;
  };
  auto put_reflector = [&](const Reflector &r, int L, int i0, int s, int k, double *sv, double *stau) {
    sv[lane] = r.v;
    if (lane == 0) { *stau = r.tau; p.tau2[(size_t)k + (size_t)s * p.ldt] = r.tau; }
    if (lane < L) p.V2[(size_t)(i0 + lane) + (size_t)s * p.ldv2] = r.v;
  };
  while (true) {
    __syncthreads();
    if (t == 0) s_sweep = (int)atomicAdd(&p.ctl[0], 1u);
    __syncthreads();
    const int s = s_sweep;
    if (s >= p.nsweeps) break;
    const int K = (n - 3 - s) / SB + 1;
    const int Kprev = (s > 0) ? (n - 2 - s) / SB + 1 : 0;
    const bool has_follower = s + 1 < p.nsweeps;
    for (int k = 0; k < K; ++k) {
      const int i0 = s + 1 + k * SB;                       // first index of I_k
      const int L = (n - i0 < SB) ? n - i0 : SB;           // its length (>= 2)
      int L1 = n - i0 - SB; if (L1 > SB) L1 = SB; if (L1 < 0) L1 = 0;
      long long tc0 = 0, tc1 = 0, tc2 = 0, tc3 = 0, tc4 = 0;
      const bool prof = p.prof && blockIdx.x == 0 && t == 0;
      if (prof) tc0 = clock64();
      // the previous sweep's task k+1, if it has one: its late numbers come through the mailbox
      const bool lead = s > 0 && k + 1 < Kprev;
      double *mail = p.mail + ((size_t)((s - 1) & 3) * p.kmax + (k + 1)) * MAILW;
      // ---- gate: the previous sweep's task k is complete (or that sweep is finished); fetch D_k and B_k
      if (t == 0 && s > 0) {
        if (!wait_word(&p.prog[s - 1], (unsigned)(k + 1 < Kprev ? k + 1 : Kprev))) { give_up(); s_ok = 0; }
      }
      __syncthreads();
      if (!s_ok) return;
      double dl[CW], bk[CW];
#pragma unroll
      for (int j = 0; j < CW; ++j) {
        const int c = c0w + j;
        dl[j] = (lane < L && c <= lane) ? ld_sc1(AB + (unsigned)((lane - c) + (i0 + c) * LDAB)) : 0.0;
        bk[j] = (lane < L1 && c < L) ? ld_sc1(AB + (unsigned)((SB + lane - c) + (i0 + c) * LDAB)) : 0.0;
      }
      // a first look into the mailbox travels with the blocks (last wave: lane r takes entry r+1, the last lane
      // beta and entry 0)
      double m_a = 0.0, m_b = 0.0;
      if (lead && wave == NW - 1) {
        m_a = ld_sc1(mail + ((lane < 63) ? lane + 1 : 64));
        if (lane == 63) m_b = ld_sc1(mail);
      }
      if (prof) tc1 = clock64();
      // ---- (a) the reflector of task 0: x = A(I_0, s) (final: the previous sweep's task 0 is complete).  Those of
      // the later tasks were made at the end of the previous task (below).
      const int cur = k & 1;
      if (k == 0) {
        if (wave == 0) {
          const double x = (lane < L) ? ld_sc1(AB + (size_t)(1 + lane) + (size_t)s * LDAB) : 0.0;
          const Reflector r = reflector_of(x, lane);
          put_reflector(r, L, i0, s, k, s_v[cur], &s_tau[cur]);
          if (lane < L) st_sc1(AB + (size_t)(1 + lane) + (size_t)s * LDAB, (lane == 0) ? r.beta : 0.0);
        }
        __syncthreads();                                                     // #1 (first task only)
      }
      if (prof) tc2 = clock64();
      const double tau = s_tau[cur];
      const double v_r = s_v[cur][lane];
      double vc[CW];
#pragma unroll
      for (int j = 0; j < CW; ++j) vc[j] = s_v[cur][c0w + j];          // broadcast reads (16-byte pairs): 0.1286 -> 0.1272 s
                                                                       // at N = 16384 against eight v_readlane pairs of v_r
      // ---- the late numbers of the previous sweep's task k+1: our last column of B_k and the corner of D_k
      if (lead && wave == NW - 1) {
        unsigned spins = 0;
        bool ok = true;
        while (__any(mail_empty(m_a) || (lane == 63 && mail_empty(m_b)))) {
          if ((++spins & 63u) == 0u &&
              (spins > kSpinLimit || __hip_atomic_load(&p.ctl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
            ok = false; break;
          }
          __builtin_amdgcn_s_sleep(1);
          if (mail_empty(m_a)) m_a = ld_sc1(mail + ((lane < 63) ? lane + 1 : 64));
          if (lane == 63 && mail_empty(m_b)) m_b = ld_sc1(mail);
        }
        if (!ok) { if (lane == 0) { give_up(); s_ok = 0; } }
        else {
          bk[CW - 1] = (lane < L1 && SB - 1 < L) ? m_a : 0.0;
          if (lane == 63 && SB - 1 < L) dl[CW - 1] = m_b;
          // the line is empty again for the task that uses it four sweeps on (drained before this task completes)
          const double e = __longlong_as_double((long long)kMailEmpty);
          st_sc1(mail + ((lane < 63) ? lane + 1 : 64), e);
          if (lane == 63) st_sc1(mail, e);
        }
      }
      // ---- D_k as a full symmetric image
#pragma unroll
      for (int j = 0; j < CW; ++j) {
        const int c = c0w + j;
        if (c <= lane) { s_D[lane * DLD + c] = dl[j]; s_D[c * DLD + lane] = dl[j]; }
      }
      __syncthreads();                                                       // #2
      if (!s_ok) return;
      if (prof) tc3 = clock64();
      // ---- partial sums of p = D v and q = B_k v
      double dd[CW];
      {
        double pp, qq;
#pragma unroll
        for (int j = 0; j < CW; ++j) dd[j] = s_D[lane * DLD + c0w + j];
        task_partials<CW>(dd, bk, vc, pp, qq);
        s_p[wave][lane] = pp; s_q[wave][lane] = qq;
      }
      __syncthreads();                                                       // #3
      if (prof) tc4 = clock64();
      // ---- the reflector of task k+1 first: it needs column 0 of the new B_k = B_k H only (first wave), and the
      // next sweep is waiting for its beta.  The column goes to memory at once as (beta, 0, ..., 0), beta into
      // the mailbox (in the last task: entry (0,0) of the final B_k, or nothing).
      if (wave == 0) {
        double qs = 0.0;
#pragma unroll
        for (int w = 0; w < NW; ++w) qs += s_q[w][lane];
        const double col0 = (L1 > 0) ? task_col0(bk[0], tau, qs, vc[0]) : 0.0;
        double b00 = col0;                                 // lane 0: entry (0,0) of the new B_k
        if (k + 1 < K) {
          const int i0n = i0 + SB;
          const Reflector r = reflector_of(col0, lane);
          put_reflector(r, L1, i0n, s, k + 1, s_v[cur ^ 1], &s_tau[cur ^ 1]);
          b00 = r.beta;
          // (beta itself travels by mailbox when there is a follower to rewrite that entry: see phase (c))
          if (lane < L1 && !(lane == 0 && k > 0 && has_follower))
            st_sc1(AB + (unsigned)((SB + lane) + i0 * LDAB), (lane == 0) ? r.beta : 0.0);
        }
        if (lane == 0 && k > 0 && has_follower)
          st_sc1(p.mail + ((size_t)(s & 3) * p.kmax + k) * MAILW + 64, b00);
      }
      // ---- (c) D_k <- H D_k H
      double psum = 0.0;
#pragma unroll
      for (int w = 0; w < NW; ++w) psum += s_p[w][lane];
      const double w_r = task_w(tau, psum, v_r);
#pragma unroll
      for (int j = 0; j < CW; ++j) {
        const int c = c0w + j;
        // (w of column c is w of row c, which lane c of this wave has: a v_readlane instead of more LDS reads per
        // column -- eight waves share one LDS pipe)
        dd[j] = task_rank2(dd[j], v_r, lane_value(w_r, c), w_r, vc[j]);
        if (c == 0 && k > 0 && has_follower)               // column 0 of D_k: the follower's late numbers (entries >= L are 0)
          st_sc1(p.mail + ((size_t)(s & 3) * p.kmax + k) * MAILW + lane, dd[j]);
        // (column 0 of a task k > 0 goes to the follower through the mailbox only: the follower rewrites exactly
        // these 64 entries -- the corner of its D_{k-1} and the last column of its B_{k-1} -- and two stores of one
        // entry from two sweeps would have to be ordered by a wait)
        if (c <= lane && lane < L && !(c == 0 && k > 0 && has_follower))
          st_sc1(AB + (unsigned)((lane - c) + (i0 + c) * LDAB), dd[j]);
      }
      // ---- (d) B_k <- B_k H (rows I_{k+1}, columns I_k)
      double bp[CW];
#pragma unroll
      for (int j = 0; j < CW; ++j) bp[j] = 0.0;
      if (L1 > 0) {
        double qsum = 0.0;
#pragma unroll
        for (int w = 0; w < NW; ++w) qsum += s_q[w][lane];
        const double q_r = tau * qsum;
#pragma unroll
        for (int j = 0; j < CW; ++j) bp[j] = task_right(bk[j], q_r, vc[j]);
        if (k == K - 1) {                                  // no further task in this sweep: the block is final
#pragma unroll
          for (int j = 0; j < CW; ++j) {
            const int c = c0w + j;
            if (lane < L1 && c < L && !(lane == 0 && c == 0 && k > 0 && has_follower))
              st_sc1(AB + (unsigned)((SB + lane - c) + (i0 + c) * LDAB), bp[j]);
          }
        }
      }
      if (prof) {
        const long long tc5 = clock64();
        p.prof[0] += tc1 - tc0; p.prof[1] += tc2 - tc1; p.prof[2] += tc3 - tc2; p.prof[3] += tc4 - tc3;
        p.prof[4] += tc5 - tc4; p.prof[6] += 1;
      }
      if (k + 1 < K) {
        __syncthreads();                                                     // #4: the new reflector is in LDS
        // ---- (b) B_k <- H' B_k with the NEW reflector (rows I_{k+1}): the block is final for this sweep
        task_left<CW>(bp, s_v[cur ^ 1][lane], s_tau[cur ^ 1], s_t[wave], lane, wave == 0);
#pragma unroll
        for (int j = 0; j < CW; ++j) {
          const int c = c0w + j;
          if (c > 0 && lane < L1)           // column 0 is (beta, 0, ..., 0) and went to memory with the reflector
            st_sc1(AB + (unsigned)((SB + lane - c) + (i0 + c) * LDAB), bp[j]);
        }
        // the task is complete once its stores are (the last task is told below)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t == 0) __hip_atomic_store(&p.prog[s], (unsigned)(k + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    // ---- the last task of the sweep: publish once its stores have completed
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0) __hip_atomic_store(&p.prog[s], (unsigned)K, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// ------------------------------------------------------------------------ positions: blocks in registers
constexpr int PMAILW = 80;     // doubles per line of the position kernel: 64 numbers = four whole 128-byte lines, the 65th in a fifth
struct PosArgs {
  int n, K0;             // K0 = number of positions = tasks of sweep 0
  double *AB;
  double *V2; int ldv2;
  double *tau2; int ldt;
  double *fwd, *bwd;     // [K0 + 1][4][MAILW]: line (k, s & 3) is written by position k-1 (fwd) / k+1 (bwd) for position k
  unsigned *retired;     // [K0 + 1]: position k has finished its last task and left A(n-1, n-1) in the band array
  unsigned *ctl;         // [3] abandoned, [4] census of the workgroups
  int per;               // 0: workgroup b holds position b; > 0: position (b & 7) * per + (b >> 3), neighbours mostly on one XCD
  unsigned census_spins;
  const int *skip;       // device: non-zero = the band is not valid (the first stage raised its flag): do nothing
  unsigned jitter;       // test aid (EK_SB2ST_JITTER): pseudo-random pauses of single positions, to shake the timing
  long long *trace; int trace_k;   // EK_SB2ST_TRACE=<position>: wall-clock stamps (10 ns) of that position's first 2048 sweeps
};

template <int CTRL>
__device__ __forceinline__ double dpp_shift(double v) {   // 0x130: lane i <- lane i + 1 (lane 63 keeps its own value)
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(v), __double2loint(v), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(v), __double2hiint(v), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}

__global__ __launch_bounds__(512) void chase_pos_kernel(PosArgs p) {
  constexpr int NW = 8, CW = SB / NW;
  __shared__ __attribute__((aligned(16))) double s_v[2][SB];   // [0] the reflector of this task, [1] the one it makes
  __shared__ double s_p[NW][SB], s_q[NW][SB];
  __shared__ double s_t[NW][CW * 65];
  __shared__ double s_x[NW][SB], s_y[NW][SB], s_row[SB], s_b0[SB];
  __shared__ double s_tau[2], s_b00;
  __shared__ int s_ok;
  constexpr int RW = 1;                                    // the wave that makes the reflectors
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int k = p.per > 0 ? (int)(blockIdx.x & 7) * p.per + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  if (k >= p.K0) return;
  if (p.skip && (*p.skip & 0xff)) return;
  const int n = p.n;
  const int c0w = CW * wave;
  double *AB = p.AB;
  unsigned *abandoned = &p.ctl[3];
  auto give_up = [&]() { __hip_atomic_store(abandoned, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
  // ---- census: every position must be resident before any of them waits for a neighbour
  if (t == 0) {
    int ok = 1;
    atomicAdd(&p.ctl[4], 1u);
    unsigned spins = 0;
    while (__hip_atomic_load(&p.ctl[4], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)p.K0) {
      __builtin_amdgcn_s_sleep(8);
      if (++spins > p.census_spins || __hip_atomic_load(abandoned, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { ok = 0; break; }
    }
    if (!ok) give_up();
    s_ok = ok;
  }
  __syncthreads();
  if (!s_ok) return;
  // one mailbox line: lane l takes entry l, lane 0 entry 64 as well; bounded; the reader empties the line again
  // (two polls in flight half a round trip apart: a message is seen about a quarter of a round trip after it lands
  // instead of half of one)
  auto take_line = [&](double *line, bool with65, double &a, double &b) -> bool {
    const bool l65 = with65 && lane == 0;
    a = ld_sc1(line + lane); b = l65 ? ld_sc1(line + 64) : 0.0;
    __builtin_amdgcn_s_sleep(2);
    double a2 = ld_sc1(line + lane), b2 = l65 ? ld_sc1(line + 64) : 0.0;
    unsigned spins = 0;
    while (__any(mail_empty(a) || (l65 && mail_empty(b)))) {
      if ((++spins & 63u) == 0u &&
          (spins > kSpinLimit || __hip_atomic_load(abandoned, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)))
        return false;
      a = a2; b = b2;
      a2 = ld_sc1(line + lane); b2 = l65 ? ld_sc1(line + 64) : 0.0;
    }
    const double e = __longlong_as_double((long long)kMailEmpty);
    st_sc1(line + lane, e);
    if (l65) st_sc1(line + 64, e);
    return true;
  };
  // ---- the blocks of sweep 0 from the packed band (written by an earlier kernel): D_k as a full image
  const int s_last = n - 3 - SB * k;                     // the last sweep that has a task at this position (L = 2 then)
  double dd[CW], bk[CW];
  {
    const int i0 = 1 + SB * k;
    const int L = (n - i0 < SB) ? n - i0 : SB;
    int L1 = n - i0 - SB; if (L1 > SB) L1 = SB; if (L1 < 0) L1 = 0;
#pragma unroll
    for (int j = 0; j < CW; ++j) {
      const int c = c0w + j;
      const int hi = (lane > c) ? lane : c, lo = (lane > c) ? c : lane;
      dd[j] = (lane < L && c < L) ? AB[(size_t)(hi - lo) + (size_t)(i0 + lo) * LDAB] : 0.0;
      bk[j] = (lane < L1 && c < L) ? AB[(size_t)(SB + lane - c) + (size_t)(i0 + c) * LDAB] : 0.0;
    }
  }
  double xnext = 0.0;                                      // position 0, wave 0: column s+1 below the diagonal
  for (int s = 0; s <= s_last; ++s) {
    const int i0 = s + 1 + k * SB;
    const int L = (n - i0 < SB) ? n - i0 : SB;
    int L1 = n - i0 - SB; if (L1 > SB) L1 = SB; if (L1 < 0) L1 = 0;
    const int K = (n - 3 - s) / SB + 1;
    const int Kprev = (s > 0) ? (n - 2 - s) / SB + 1 : 0;
    const bool lead = s > 0 && k + 1 < Kprev;              // position k+1 had a task in sweep s-1: its late numbers come by mail
    // every store of the previous task (mail, emptied lines) has completed before this task sends anything
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef EK_POS_TRACE      // diagnostic build only: the stamps cost 15 % of the stage even when no position is traced
    const bool tr = p.trace && k == p.trace_k && s < 2048 && lane == 0;
    auto stamp = [&](int slot) { if (tr) p.trace[16 * s + slot] = (long long)wall_clock64(); };
#else
    auto stamp = [](int) {};
#endif
    if (wave == 0) stamp(0);
    if (p.jitter) {                                        // (uniform per workgroup and sweep)
      const unsigned h = ((unsigned)s * 2654435761u) ^ ((unsigned)k * 40503u * p.jitter);
      if (((h >> 9) & 15u) == 0u) for (unsigned q = 0; q < ((h >> 3) & 63u); ++q) __builtin_amdgcn_s_sleep(64);
    }
    // ---- the reflector of this task: position 0 makes it from its own column, the others receive it
    if (wave == 0) {
      if (k == 0) {
        const double x = (s == 0) ? ((lane < L) ? AB[(size_t)(1 + lane)] : 0.0) : xnext;
        const Reflector r = reflector_of(x, lane);
        s_v[0][lane] = r.v;
        if (lane == 0) { s_tau[0] = r.tau; p.tau2[(size_t)s * p.ldt] = r.tau; AB[(size_t)1 + (size_t)s * LDAB] = r.beta; }   // (e(s): read after the kernel)
        if (lane < L) p.V2[(size_t)(i0 + lane) + (size_t)s * p.ldv2] = r.v;
      } else {
        double a, b;
        // (v_0 = 1 always: tau travels in its place, the message is four whole 128-byte lines)
        if (!take_line(p.fwd + ((size_t)k * 4 + (s & 3)) * PMAILW, false, a, b)) { if (lane == 0) { give_up(); s_ok = 0; } }
        s_v[0][lane] = (lane == 0) ? 1.0 : a;
        if (lane == 0) s_tau[0] = a;
      }
      stamp(1);
    }
    // ---- the entering column: the late numbers of position k+1's task of the previous sweep
    if (wave == NW - 1 && s > 0) {
      if (lead) {
        double a, b;
        if (!take_line(p.bwd + ((size_t)k * 4 + ((s - 1) & 3)) * PMAILW, true, a, b)) { if (lane == 0) { give_up(); s_ok = 0; } }
        // a: entry l = D_{k+1}(l, 0) of the previous sweep; b (lane 0): beta.  Row r of our new last column is entry r + 1
        const double up = dpp_shift<0x130>(a);
        const double beta = lane_value(b, 0), corner = lane_value(a, 0);
        const double col = (lane < 63) ? up : beta;
        bk[CW - 1] = (lane < L1 && SB - 1 < L) ? col : 0.0;
        if (lane == 63 && SB - 1 < L) dd[CW - 1] = corner;
        stamp(2);
      } else if (L == SB) {
        // the one sweep after position k+1 has retired (or never existed): the corner is A(n-1, n-1) in the band array
        int ok = 1;
        if (lane == 0) {
          unsigned spins = 0;
          while (!__hip_atomic_load(&p.retired[k + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 63u) == 0u &&
                (spins > kSpinLimit || __hip_atomic_load(abandoned, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) { ok = 0; break; }
          }
          if (!ok) { give_up(); s_ok = 0; }
        }
        wave_sync();
        const double corner = ld_sc1(AB + (size_t)(i0 + SB - 1) * LDAB);
        if (lane == 63) dd[CW - 1] = corner;
      }
    }
    __syncthreads();                                                         // #1
    if (!s_ok) return;
    if (wave == 0) stamp(3);
    const double tau = s_tau[0];
    const double v_r = s_v[0][lane];
    double vc[CW];
#pragma unroll
    for (int j = 0; j < CW; ++j) vc[j] = s_v[0][c0w + j];
    {
      double pp, qq;
      task_partials<CW>(dd, bk, vc, pp, qq);
      s_p[wave][lane] = pp; s_q[wave][lane] = qq;
      if (wave == 0) s_b0[lane] = bk[0];
    }
    __syncthreads();                                                         // #2
    if (wave == 0) stamp(7);
    // ---- the reflector of position k+1: position k+1 is waiting for it.  Made by wave RW from column 0 of B_k (which
    // wave 0 has put into LDS) while wave 0 updates column 0 of D_k, the late numbers position k-1 is waiting for.
    if (wave == RW) {
      double qs = 0.0;
#pragma unroll
      for (int w = 0; w < NW; ++w) qs += s_q[w][lane];
      const double col0 = (L1 > 0) ? task_col0(s_b0[lane], tau, qs, s_v[0][0]) : 0.0;
      double b00 = col0;
      if (k + 1 < K) {
        const Reflector r = reflector_of(col0, lane);
        double *line = p.fwd + ((size_t)(k + 1) * 4 + (s & 3)) * PMAILW;
        st_sc1(line + lane, (lane == 0) ? r.tau : r.v);
        stamp(4);
        s_v[1][lane] = r.v;
        if (lane == 0) { s_tau[1] = r.tau; p.tau2[(size_t)(k + 1) + (size_t)s * p.ldt] = r.tau; }
        if (lane < L1) p.V2[(size_t)(i0 + SB + lane) + (size_t)s * p.ldv2] = r.v;
        b00 = r.beta;
      }
      if (lane == 0) {                                     // entry (0, 0) of the new B_k
        if (k > 0) st_sc1(p.bwd + ((size_t)(k - 1) * 4 + (s & 3)) * PMAILW + 64, b00);
        else s_b00 = b00;
      }
    }
    // ---- D_k <- H D_k H
    double psum = 0.0;
#pragma unroll
    for (int w = 0; w < NW; ++w) psum += s_p[w][lane];
    const double w_r = task_w(tau, psum, v_r);
    if (wave == 0) {
      // column 0 of D_k leaves the position: the late numbers of position k-1, or (position 0) d and the next column
      dd[0] = task_rank2(dd[0], v_r, lane_value(w_r, 0), w_r, vc[0]);
      if (k > 0) { st_sc1(p.bwd + ((size_t)(k - 1) * 4 + (s & 3)) * PMAILW + lane, dd[0]); stamp(5); }
      else if (lane == 0) AB[(size_t)(s + 1) * LDAB] = dd[0];              // d(s+1): read after the kernel -- a plain store: the
                                                                            // wait at the top of the next sweep would sit out an sc1 store's way to memory
    }
#pragma unroll
    for (int j = 0; j < CW; ++j)
      if (!(wave == 0 && j == 0)) dd[j] = task_rank2(dd[j], v_r, lane_value(w_r, c0w + j), w_r, vc[j]);
    // ---- B_k <- B_k H
    double bp[CW];
    {
      double qsum = 0.0;
#pragma unroll
      for (int w = 0; w < NW; ++w) qsum += s_q[w][lane];
      const double q_r = tau * qsum;
#pragma unroll
      for (int j = 0; j < CW; ++j) bp[j] = (L1 > 0) ? task_right(bk[j], q_r, vc[j]) : 0.0;
    }
    if (wave == 0) stamp(8);
    if (k + 1 < K) {
      __syncthreads();                                                       // #3: the new reflector is in LDS
      if (wave == 0) stamp(9);
      task_left<CW>(bp, s_v[1][lane], s_tau[1], s_t[wave], lane, wave == 0);
    }
    if (wave == 0) stamp(6);
    if (s == s_last) break;
    // ---- the blocks of the next sweep: one row and one column further down the band
    if (lane == 0) {
#pragma unroll
      for (int j = 0; j < CW; ++j) s_row[c0w + j] = bp[j];                   // row 0 of B_k: the new last row and column of D_k
    }
    s_x[wave][lane] = dd[0]; s_y[wave][lane] = bp[0];
    __syncthreads();                                                         // #4
    if (wave == 0) stamp(10);
    if (k == 0 && wave == 0) {
      const double up = dpp_shift<0x130>(dd[0]);
      xnext = (lane < 63) ? up : s_b00;
    }
#pragma unroll
    for (int j = 0; j + 1 < CW; ++j) { dd[j] = dd[j + 1]; bk[j] = bp[j + 1]; }
    dd[CW - 1] = (wave + 1 < NW) ? s_x[(wave + 1) & (NW - 1)][lane] : 0.0;
    bk[CW - 1] = (wave + 1 < NW) ? s_y[(wave + 1) & (NW - 1)][lane] : 0.0;
#pragma unroll
    for (int j = 0; j < CW; ++j) {
      const double brow0 = lane_value(bk[j], 0);             // B_k(0, c + 1)
      const double du = dpp_shift<0x130>(dd[j]), bu = dpp_shift<0x130>(bk[j]);
      dd[j] = (lane == 63) ? brow0 : du;
      bk[j] = (lane == 63) ? 0.0 : bu;
    }
    if (wave == NW - 1) dd[CW - 1] = (lane < 63) ? s_row[lane + 1] : 0.0;    // the corner and B's last column: next sweep's mail
    if (wave == 0) stamp(11);
  }
  // ---- retirement: the last task had L = 2; what is left of D_k is A(n-1, n-1) (position 0: also e(n-2))
  if (wave == 0) {
    if (lane == 1) {
      st_sc1(AB + (size_t)(n - 1) * LDAB, dd[1]);
      if (k == 0) st_sc1(AB + (size_t)1 + (size_t)(n - 2) * LDAB, dd[0]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_store(&p.retired[k], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

__global__ void set_word_kernel(unsigned *p, unsigned v) { *p = v; }

// the band again if the position-owned kernel gave up half way: the state chase_kernel starts from
__global__ void repack_band_kernel(int n, const double *__restrict__ AB0, double *__restrict__ AB, const unsigned *ctl) {
  if (!ctl[3]) return;
  const int c = blockIdx.x;
  for (int d = threadIdx.x; d < LDAB; d += blockDim.x) AB[(size_t)d + (size_t)c * LDAB] = AB0[(size_t)d + (size_t)c * LDAB];
}

// ------------------------------------------------------------------------ Q2: compact-WY factors
// Block S of sweeps = sweeps 32 S - 1 .. 32 S + 30 (block 0 has 31): with that offset every window of
// rows starts on an EVEN row (o = 32 S + 64 k), so rows travel as aligned 16-byte pairs.
constexpr int QG = 32;                 // sweeps per compact-WY block
constexpr int QR = QG + SB;            // rows of a block's window (95 used, 96 with padding)
constexpr int QVLD = QG + 1;           // LDS leading dimension of the V image (row-major)
constexpr int QREC = 2 * QR * QG;      // doubles per group record: V and -(V T), 96 x 32 row-major each (the LDS images
                                       // have the leading dimension QVLD: the padding is added on the way in)

struct Q2Geom {
  int n, nsweeps, nS, kmax;            // kmax: groups of block 0 (the most any block has)
  const unsigned *offS;                // [nS + 1] device: first record of block S (block S has q2_groups_of_block(n, S) of them:
                                       // the store is a triangle, not nS x kmax)
};
__host__ __device__ inline int q2_first_sweep(int S) { return S * QG - 1; }
__host__ __device__ inline int q2_groups_of_block(int n, int S) {   // number of k for which any reflector exists
  int s0 = q2_first_sweep(S); if (s0 < 0) s0 = 0;
  return (s0 <= n - 3) ? (n - 3 - s0) / SB + 1 : 0;
}

// V image of group (S, k): rows o .. o + QR - 1 (o = 32 S + 64 k), column i = sweep 32 S - 1 + i,
// non-zero for i <= row - o < i + SB where the reflector (s, k) exists.
__device__ __forceinline__ double q2_v_entry(const Q2Geom &g, const double *__restrict__ V2, int ldv2, int S, int k,
                                             int rr, int i) {
  const int s = q2_first_sweep(S) + i, o = S * QG + k * SB, row = o + rr;
  if (s < 0 || s >= g.nsweeps || rr < i || rr >= i + SB || row >= g.n) return 0.0;
  if (s + 1 + k * SB > g.n - 2) return 0.0;                 // task (s, k) does not exist
  return V2[(size_t)row + (size_t)s * ldv2];
}

// One workgroup (4 waves) per group: the record [V | -V T] the application streams.  The Gram matrix
// G = V^T V and the product V T run on the matrix cores; between them one wave forms T by DLARFT's forward
// columnwise recurrence T(0:i, i) = -tau_i T(0:i, 0:i) G(0:i, i) -- the only serial part (32 steps).
// (The first version did all of it in one wave with per-lane dot products out of LDS: 8 ms per solve at
// N = 16384; this one takes 3.)
__global__ __launch_bounds__(256) void q2_tfactor_kernel(Q2Geom g, const double *__restrict__ V2, int ldv2,
                                                         const double *__restrict__ tau2, int ldt,
                                                         double *__restrict__ Rec) {
  __shared__ double sV[QR * QVLD];
  __shared__ double sG[QG * QVLD];
  __shared__ double sT[QG * QVLD];
  __shared__ double s_tau[QG];
  const int S = blockIdx.y, k = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int l15 = lane & 15, l4 = lane >> 4;
  if (k >= q2_groups_of_block(g.n, S)) return;
  double *rec = Rec + ((size_t)g.offS[S] + k) * QREC;
  for (int idx = t; idx < QR * QG; idx += 256) {
    const int rr = idx % QR, i = idx / QR;
    const double v = q2_v_entry(g, V2, ldv2, S, k, rr, i);
    sV[rr * QVLD + i] = v;
    rec[rr * QG + i] = v;
  }
  for (int idx = t; idx < QG * QVLD; idx += 256) sT[idx] = 0.0;
  if (t < QG) {
    const int s = q2_first_sweep(S) + t;
    const bool exists = s >= 0 && s < g.nsweeps && s + 1 + k * SB <= g.n - 2;
    s_tau[t] = exists ? tau2[(size_t)k + (size_t)s * ldt] : 0.0;
  }
  __syncthreads();
  // G = V^T V: 2 x 2 tiles of 16 x 16, one per wave, 24 k-steps over the 96 rows
  {
    const int ti = wave >> 1, tj = wave & 1;
    double4_t acc = (double4_t){0.0, 0.0, 0.0, 0.0};
    for (int kk = 0; kk < QR; kk += 4) {
      const double x = sV[(kk + l4) * QVLD + 16 * ti + l15];
      const double y = sV[(kk + l4) * QVLD + 16 * tj + l15];
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) sG[(16 * ti + l4 + 4 * r) * QVLD + 16 * tj + l15] = acc[r];
  }
  __syncthreads();
  if (wave == 0) {
    for (int i = 0; i < QG; ++i) {
      const double ti = s_tau[i];
      if (lane < i) {
        double a0 = 0.0, a1 = 0.0;
        int l = lane;
        for (; l + 1 < i; l += 2) {
          a0 += sT[lane * QVLD + l] * sG[l * QVLD + i];
          a1 += sT[lane * QVLD + l + 1] * sG[(l + 1) * QVLD + i];
        }
        if (l < i) a0 += sT[lane * QVLD + l] * sG[l * QVLD + i];
        sT[lane * QVLD + i] = -ti * (a0 + a1);
      } else if (lane == i) sT[i * QVLD + i] = ti;
      wave_sync();
    }
  }
  __syncthreads();
  // -(V T): 6 x 2 tiles, three per wave, 8 k-steps over the 32 reflectors
  for (int tile = wave; tile < 12; tile += 4) {
    const int ti = tile >> 1, tj = tile & 1;
    double4_t acc = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < QG; kk += 4) {
      const double x = sV[(16 * ti + l15) * QVLD + kk + l4];
      const double y = sT[(kk + l4) * QVLD + 16 * tj + l15];
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) rec[QR * QG + (16 * ti + l4 + 4 * r) * QG + 16 * tj + l15] = -acc[r];
  }
}

// ------------------------------------------------------------------------ Q2: application
// Z <- Q2 Z as a two-dimensional pipeline of PASSES (bundle of NBLK blocks of sweeps, slab): a pass applies the
// blocks of sweeps S + NBLK - 1, ..., S (S = NBLK q) to one slab of 64 columns, walking down the rows in chunks of
// 64 starting at row 32 S.  In step k a window of 2 NBLK + 4 row tiles (16 rows each) sits in registers; block b of
// the bundle acts on window rows 32 b .. 32 b + 95: the groups (S + NBLK - 1, k), ..., (S, k) in that order, after
// which the first 64 rows of the window are final for the whole bundle and leave, and 64 new rows come in.  Z is
// streamed once per BUNDLE (n^3 / (8 NBLK) bytes each way for all columns; one block per pass was bound by that
// stream), and the fetch, the two LDS transposes and the store of a chunk are shared by NBLK groups.  The pass of the
// bundle above starts 32 NBLK rows further down, so the rows a step takes in are that pass's chunk k + 1: passes of
// one slab follow each other a few chunks apart, few columns (the *_select arms) still fill the chip, and many columns
// run two passes per CU (two waves per SIMD is what the fp64 matrix pipe needs, profiles/r02_mfma_peak.txt).
// Persistent workgroups take passes from a ticket counter in dependency order (bundles descending), so a workgroup
// only waits for passes whose owners are running; a finished chunk is handed over through memory with agent-scope
// (sc1) stores, a drain, a barrier and one progress word per pass, like the sweeps of chase_kernel.
//
// Inside a pass every wave streams its own 16 columns and keeps its window of Z in REGISTERS: the
// accumulator layout of v_mfma_f64_16x16x4 (D(i,j) in lane (j = lane & 15, i = lane / 16 + 4 reg)) is
// also the layout of its second operand for the k-step over rows 4 reg .. 4 reg + 3, so the same
// registers serve as the operand of W1 = V^T Zw and as the accumulator of Zw += (-V T) W1, and W1
// goes from the first product into the second without leaving the registers.  Only the factors
// [V | -V T] of a group pass through LDS.  The order of the groups on any element of Z is the same for every NBLK,
// so 2, 3 and 4 blocks per pass give the same bits (and the same as round 2's pair kernel: tools/q2_anchor.py).
// Registers: window 8 (2 NBLK + 4), record halves in flight 48, chunk in flight 32, W1 16: 256 at NBLK = 4.
constexpr int QNC = 64;
constexpr int QSTLD = 34;              // per-wave transposing buffer: 16 columns x 32 rows (+2), column-major
// LDS images of a record, laid out for the 64-bit reads of the two products (ds_read_b64: two groups of 32 lanes, 64
// banks of 4 bytes -- MI355X_MICROARCH.md, LDS): the first product reads V by rows (lanes l15 = 16 consecutive columns,
// l4 = four consecutive rows): leading dimension 32 with the two halves of every ODD row swapped, so the rows of a lane
// group lie in opposite halves of the bank row; the second reads -(V T) by columns (lanes l15 = 16 rows, l4 = four
// consecutive columns): leading dimension 34, 16 rows x 2 columns = 64 different banks.  With 33 for both (round 2:
// right for 16-lane groups) every one of these reads took two LDS cycles per group: SQ_LDS_BANK_CONFLICT as large
// as SQ_ACTIVE_INST_LDS.
constexpr int QVS = QG, QTS = QG + 2;
constexpr int QOPSZ = QR * QVS + QR * QTS;   // doubles of the operand buffer: V image, then -(V T) image
constexpr unsigned kQ2Done = 0x7fffffffu;

struct Q2ApplyArgs {
  Q2Geom g;
  const double *Rec;
  double *Z; int ldz; int ncols;
  int nslab, npass, npair;   // npair = ceil(nS / NBLK) bundles of blocks of sweeps (NBLK q ..); bundle q = 0 is the lowest
  unsigned *prog;        // [npass] chunks stored by pass (bundle, slab) at index (npair-1-bundle) * nslab + slab
  unsigned *ctl;         // [2] ticket, [1] abort (shared with the chase)
  int extra;             // chunks a pass keeps behind its predecessor beyond the two it must
};

typedef double d2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void st_sc1_x2(double *p, double a, double b) {
  d2_t v = {a, b};
  asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
}

#ifndef Q2_PF
#define Q2_PF 1        // LDS operands read this many steps ahead of their use (1, 2, 3 inside a solve: 188.6 / 190.8 / 188.6 ms;
#endif                 // one step leaves four blocks per pass without a spilled register)
#ifndef Q2_PUTV0
#define Q2_PUTV0 24      // second product: slot of the first of the six writes of the next V image, and their spacing
#define Q2_PUTVS 4
#endif
template <int NBLK>
__global__ __launch_bounds__(256, 2) void q2_apply_nb_kernel(Q2ApplyArgs p) {
  constexpr int WT = 2 * NBLK + 4;                        // row tiles of the window
  extern __shared__ __attribute__((aligned(16))) double q2smem[];
  double *sOp = q2smem;
  __shared__ int s_pass, s_ok;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l15 = lane & 15, l4 = lane >> 4;
  double *st = q2smem + QOPSZ + wave * 16 * QSTLD;      // this wave's transposing buffer
  const int n = p.g.n;
  const double *sV = sOp, *sVT = sOp + QR * QVS;
  // (Measured and withdrawn, round 4: the passes of slab c dealt to the workgroups that sit on XCD c % 8 -- read from
  // HW_REG_XCC_ID, with stealing once a queue is empty -- so that the ~32 passes walking down one bundle side by side
  // share its group records in ONE L2: 0.2033 s against 0.2015 at N = 16384, full spectrum; the passes drift apart
  // within a few hundred steps and the records come from the Infinity Cache either way.)
  while (true) {
    __syncthreads();
    if (t == 0) s_pass = (int)atomicAdd(&p.ctl[2], 1u);
    __syncthreads();
    const int pass = __builtin_amdgcn_readfirstlane(s_pass);
    if (pass >= p.npass) break;
    const int bundle = p.npair - 1 - pass / p.nslab, slab = pass % p.nslab;      // (npair: number of bundles)
    const int S = NBLK * bundle;                         // lowest block of the bundle
    // groups and first record of each block of the bundle: wave-uniform scalars with names of their own (as the arrays
    // KSb[NBLK], offb[NBLK] of round 4 they were indexed with the block a step starts at, so the compiler kept them in
    // scratch memory -- 32 bytes per lane, a scratch load in front of every group's record address)
    auto ks_of = [&](int b) -> int { return (S + b < p.g.nS) ? q2_groups_of_block(n, S + b) : 0; };
    auto off_of = [&](int b) -> unsigned {
      return (S + b < p.g.nS) ? (unsigned)__builtin_amdgcn_readfirstlane((int)p.g.offS[S + b]) : 0u;
    };
    const int KS0 = ks_of(0), KS1 = NBLK > 1 ? ks_of(1) : 0, KS2 = NBLK > 2 ? ks_of(2) : 0, KS3 = NBLK > 3 ? ks_of(3) : 0;
    const unsigned of0 = off_of(0), of1 = NBLK > 1 ? off_of(1) : 0u, of2 = NBLK > 2 ? off_of(2) : 0u, of3 = NBLK > 3 ? off_of(3) : 0u;
    auto KSb = [&](int b) -> int { return b == 0 ? KS0 : (b == 1 ? KS1 : (b == 2 ? KS2 : KS3)); };
    const int KS = KS0;
    const int colw = slab * QNC + 16 * wave;
    const unsigned *pprog = (bundle + 1 < p.npair) ? p.prog + (size_t)(pass - p.nslab) : nullptr;   // the bundle above, same slab
    unsigned *myprog = p.prog + pass;
    if (KS <= 0) { if (t == 0) __hip_atomic_store(myprog, kQ2Done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); continue; }
    const int o0 = S * QG;                               // first row of the window at step 0 (even)
    auto wait_for = [&](unsigned need) -> bool {         // thread 0 only
      if (!pprog) return true;
      unsigned spins = 0;
      while (__hip_atomic_load(pprog, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
        __builtin_amdgcn_s_sleep(2);
        if ((++spins & 63u) == 0u &&
            (spins > kSpinLimit || __hip_atomic_load(&p.ctl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)))
          return false;
      }
      return true;
    };
    double4_t w[WT];
    // A record travels global -> registers -> LDS in two halves of 6 x 16 bytes per thread (pair j = t + 256 q of a
    // half is row j / 16, columns 2 (j % 16), +1 of the 96 x 32 matrix), each half written into its LDS image WHILE the
    // matrix instructions of the other product run: the image of -(V T) of group g during g's first product (which
    // reads V only), the image of V of group g + 1 during g's second product (which reads -(V T) only).  Written in one
    // piece in front of a group (round 2) the 50 KB cost 16 of the kernel's 206 ms at N = 16384 (timing-only build).
    static_assert(QR * QG == 6 * 256 * 2, "a half record is six 16-byte pairs per thread");
    constexpr int HALF = QR * QG / 2;                     // pairs per half
    double zreg[16];
    d2_t oreg[12];                                        // [0..5]: V of the next group, [6..11]: -(V T) of this / the next one
    const int obase_v = (t >> 4) * QVS + ((2 * (t & 15) + 16 * ((t >> 4) & 1)) & 31);   // (odd rows: halves swapped)
    const int obase_t = (t >> 4) * QTS + 2 * (t & 15);
    const int vsw = 16 * (l4 & 1);                        // first product: the column half this lane finds columns 0..15 of its row in
    auto rec_of = [&](int b, int k) -> const d2_t * {     // block b of the bundle, group k (no memory access)
      const unsigned o = b == 0 ? of0 : (b == 1 ? of1 : (b == 2 ? of2 : of3));
      return reinterpret_cast<const d2_t *>(p.Rec + ((size_t)o + k) * QREC) + t;
    };
    auto fetch_v = [&](const d2_t *rec) {
#pragma unroll
      for (int q = 0; q < 6; ++q) oreg[q] = rec[256 * q];
    };
    auto fetch_t = [&](const d2_t *rec) {
#pragma unroll
      for (int q = 0; q < 6; ++q) oreg[6 + q] = rec[HALF + 256 * q];
    };
    auto put_v = [&](int q) { *reinterpret_cast<d2_t *>(sOp + obase_v + q * 16 * QVS) = oreg[q]; };
    auto put_t = [&](int q) { *reinterpret_cast<d2_t *>(sOp + QR * QVS + obase_t + q * 16 * QTS) = oreg[6 + q]; };
    const bool cols_in = colw + 16 <= p.ncols;
    // 64 rows from global row `row0` on -> zreg: lane = row, 16 columns (sc1: another pass may have written them)
    auto fetch_rows = [&](int row0) {
      const int row = row0 + lane;
      const double *src = p.Z + (size_t)row + (size_t)colw * p.ldz;
      if (cols_in && row0 + SB <= n) {                   // interior: no predicates
#pragma unroll
        for (int c = 0; c < 16; ++c) zreg[c] = ld_sc1(src + (size_t)c * p.ldz);
      } else {
#pragma unroll
        for (int c = 0; c < 16; ++c)
          zreg[c] = (row < n && colw + c < p.ncols) ? ld_sc1(src + (size_t)c * p.ldz) : 0.0;
      }
    };
    // half h (32 rows) of zreg -> two tiles in the accumulator layout, through the per-wave buffer
    auto half_to_tiles = [&](int h, double4_t &ta, double4_t &tb) {
      if ((lane >> 5) == h) {
#pragma unroll
        for (int c = 0; c < 16; ++c) st[c * QSTLD + (lane & 31)] = zreg[c];
      }
      wave_sync();
#pragma unroll
      for (int r = 0; r < 4; ++r) { ta[r] = st[l15 * QSTLD + l4 + 4 * r]; tb[r] = st[l15 * QSTLD + 16 + l4 + 4 * r]; }
      wave_sync();
    };
    // two tiles (32 rows from global row `row0`) -> memory as aligned row pairs (16-byte write-through stores)
    auto tiles_to_rows = [&](int row0, const double4_t &ta, const double4_t &tb) {
#pragma unroll
      for (int r = 0; r < 4; ++r) { st[l15 * QSTLD + l4 + 4 * r] = ta[r]; st[l15 * QSTLD + 16 + l4 + 4 * r] = tb[r]; }
      wave_sync();
      {
        const int pr = lane & 15, cg = lane >> 4;        // row pair, 4 columns per lane
        const int row = row0 + 2 * pr;
        double *dst0 = p.Z + (size_t)row + (size_t)(colw + 4 * cg) * p.ldz;
        if (cols_in && row0 + 32 <= n) {                 // interior: no predicates
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int c = 4 * cg + q;
            st_sc1_x2(dst0 + (size_t)q * p.ldz, st[c * QSTLD + 2 * pr], st[c * QSTLD + 2 * pr + 1]);
          }
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int c = 4 * cg + q;
            if (colw + c < p.ncols) {
              const double a = st[c * QSTLD + 2 * pr], b = st[c * QSTLD + 2 * pr + 1];
              if (row + 1 < n) st_sc1_x2(dst0 + (size_t)q * p.ldz, a, b);
              else if (row < n) st_sc1(dst0 + (size_t)q * p.ldz, a);
            }
          }
        }
      }
      wave_sync();
    };
    // one group on window tiles OFF .. OFF + 5.  The operands of the matrix instructions come from LDS one to two
    // steps AHEAD of their use (PF): left to itself the compiler reads an operand pair, waits for it, and issues the two
    // instructions that need it -- an LDS round trip every 128 cycles of the matrix pipe.  In the second product the
    // accumulation chains of two tiles alternate (a chain on one accumulator issues only every other slot).
    // nrec: the record of the group after this one (its own if it is the last); row_next >= 0: the 64 rows of Z to fetch
    auto apply_group = [&](auto off_c, const d2_t *nrec, int row_next) {
      constexpr int OFF = decltype(off_c)::value;
      constexpr int PF = Q2_PF;                            // steps of look-ahead
      double4_t w1[2];
      w1[0] = (double4_t){0.0, 0.0, 0.0, 0.0}; w1[1] = (double4_t){0.0, 0.0, 0.0, 0.0};
      {
        double pa[PF + 1][2];
        auto ld1 = [&](int sidx, int slot) {
          const double *vrow = sV + (16 * (sidx >> 2) + 4 * (sidx & 3) + l4) * QVS + l15;
          pa[slot][0] = vrow[vsw]; pa[slot][1] = vrow[16 - vsw];
        };
#pragma unroll
        for (int q = 0; q < PF; ++q) ld1(q, q);
#pragma unroll
        for (int sidx = 0; sidx < 24; ++sidx) {
          if (sidx + PF < 24) ld1(sidx + PF, (sidx + PF) % (PF + 1));
          if (sidx < 12 && (sidx & 1) == 0) put_t(sidx >> 1);       // this group's -(V T) image (nobody reads it before B1)
          if (sidx == 12) {                                         // (behind the writes: their wait must not cover these)
            fetch_v(nrec);
            if (row_next >= 0) fetch_rows(row_next);
          }
          const int tile = sidx >> 2, r = sidx & 3;
          const double y = w[tile + OFF][r];
          if (tile < 5) w1[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[sidx % (PF + 1)][0], y, w1[0], 0, 0, 0);
          if (tile > 0) w1[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[sidx % (PF + 1)][1], y, w1[1], 0, 0, 0);
        }
      }
      __syncthreads();                                     // B1: -(V T) complete; everybody has finished reading V
      fetch_t(nrec);
      {
        // tiles in pairs (0,1), (2,3), (4,5): step q of a pair = k-step kk = 4 (q / 2) of tile 2 pr + (q & 1); tile 5 has
        // only the k-steps 16.. (rows 80.. of V T: columns 16.. only), its first four slots are skipped
        double px[PF + 1];
        auto seq_tile = [](int q) { return 2 * (q >> 4) + (q & 1); };
        auto seq_kk = [](int q) { return 4 * ((q & 15) >> 1); };
        auto seq_ok = [&](int q) { return !(seq_tile(q) == 5 && seq_kk(q) < 16); };
        auto ld2 = [&](int q, int slot) {
          if (seq_ok(q)) px[slot] = sVT[(16 * seq_tile(q) + l15) * QTS + l4 + seq_kk(q)];
        };
#pragma unroll
        for (int q = 0; q < PF; ++q) ld2(q, q);
#pragma unroll
        for (int q = 0; q < 48; ++q) {
          if (q + PF < 48) ld2(q + PF, (q + PF) % (PF + 1));
          if (q >= Q2_PUTV0 && q < Q2_PUTV0 + 6 * Q2_PUTVS && (q - Q2_PUTV0) % Q2_PUTVS == 0) put_v((q - Q2_PUTV0) / Q2_PUTVS);   // the next group's V image
          if (seq_ok(q)) {
            const int tile = seq_tile(q), kk = seq_kk(q);
            w[tile + OFF] = __builtin_amdgcn_mfma_f64_16x16x4f64(px[q % (PF + 1)], w1[kk >> 4][(kk & 15) >> 2], w[tile + OFF], 0, 0, 0);
          }
        }
      }
    };
    // the groups in the order they are applied: step k ascending, inside a step the blocks descending
    auto first_block = [&](int k) -> int {               // highest block that has a group k (block 0 has: k < KS)
      int b = 0;
#pragma unroll
      for (int q = 1; q < NBLK; ++q) if (k < KSb(q)) b = q;
      return b;
    };
    auto next_group = [&](int k, int b) -> const d2_t * {      // record of the group after (b, k); its own if it is the last
      int nb = -1, nk = k;
#pragma unroll
      for (int q = NBLK - 1; q >= 0; --q) if (q < b && nb < 0 && k < KSb(q)) nb = q;
      if (nb < 0) { nk = k + 1; if (nk < KS) nb = first_block(nk); }
      if (nb < 0) { nb = b; nk = k; }
      return rec_of(nb, nk);
    };
    // ---- prologue: the window at step 0; its rows from 32 NBLK on are chunk 0 of the bundle above (+ slack)
    if (t == 0) s_ok = wait_for((unsigned)(1 + p.extra)) ? 1 : 0;
    __syncthreads();
    if (!s_ok) { if (t == 0) __hip_atomic_store(&p.ctl[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
#pragma unroll
    for (int c = 0; c < WT / 4; ++c) {
      fetch_rows(o0 + 64 * c);
      half_to_tiles(0, w[4 * c], w[4 * c + 1]);
      half_to_tiles(1, w[4 * c + 2], w[4 * c + 3]);
    }
    if (WT % 4) {                                        // NBLK odd: half a chunk more
      fetch_rows(o0 + 64 * (WT / 4));
      half_to_tiles(0, w[WT - 2], w[WT - 1]);
    }
    fetch_v(rec_of(first_block(0), 0));
    fetch_t(rec_of(first_block(0), 0));
#pragma unroll
    for (int q = 0; q < 6; ++q) put_v(q);                 // (visible after the barrier at the top of step 0)
    unsigned early = 0;                                   // thread 0: an early look at the predecessor's progress
    for (int k = 0; k < KS; ++k) {
      // gate of this step: the 64 rows fetched below are chunk k + 1 of the bundle above
      if (t == 0) {
        const unsigned need = (unsigned)(k + 2 + p.extra);
        s_ok = (!pprog || early >= need || wait_for(need)) ? 1 : 0;
      }
      __syncthreads();
      if (!s_ok) { if (t == 0) __hip_atomic_store(&p.ctl[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
      // every wave has passed the drain in front of its stores of chunk k-1, so chunks <= k-2 are in memory
      if (t == 0 && k > 1) __hip_atomic_store(myprog, (unsigned)(k - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int bfirst = first_block(k);
      auto do_group = [&](auto b_c) {
        constexpr int B = decltype(b_c)::value;
        if (k >= KSb(B)) return;                         // (uniform)
        // (here: the V image of this group is in LDS and visible, its -(V T) half is on its way into registers)
        if (B == bfirst && t == 0 && pprog) early = __hip_atomic_load(pprog, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        apply_group(std::integral_constant<int, 2 * B>(), next_group(k, B),
                    (B == bfirst && k + 1 < KS) ? o0 + 64 * (k + 1) + 16 * (WT - 4) : -1);
        if (B > 0) __syncthreads();                      // B2: the next V image complete; everybody has finished reading -(V T)
      };
      if constexpr (NBLK >= 4) do_group(std::integral_constant<int, 3>());
      if constexpr (NBLK >= 3) do_group(std::integral_constant<int, 2>());
      if constexpr (NBLK >= 2) do_group(std::integral_constant<int, 1>());
      do_group(std::integral_constant<int, 0>());
      // (the stores of the previous chunk, issued a whole step ago, have long completed: this wait
      // only makes that certain before the progress word of the next step tells the follower)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      tiles_to_rows(o0 + 64 * k, w[0], w[1]);            // chunk k is final for every block of the bundle
      tiles_to_rows(o0 + 64 * k + 32, w[2], w[3]);
#pragma unroll
      for (int i = 0; i + 4 < WT; ++i) w[i] = w[i + 4];
      if (k + 1 < KS) {
        half_to_tiles(0, w[WT - 4], w[WT - 3]);
        half_to_tiles(1, w[WT - 2], w[WT - 1]);
      } else {
#pragma unroll
        for (int i = 0; i + 4 < WT; i += 2) tiles_to_rows(o0 + 64 * (k + 1) + 16 * i, w[i], w[i + 1]);   // the rest of the window
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0) __hip_atomic_store(myprog, kQ2Done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

__global__ void forward_abort_kernel(const unsigned *ctl, int *flag) { if (ctl[1]) atomicOr(flag, 4); }
__global__ void q2_offsets_kernel(int n, int nS, unsigned *offS) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    unsigned o = 0;
    for (int S = 0; S < nS; ++S) { offS[S] = o; o += (unsigned)q2_groups_of_block(n, S); }
    offS[nS] = o;
  }
}

inline size_t al256(size_t b) { return (b + 255) & ~(size_t)255; }

struct Layout {
  int nsweeps, nS, kmax, ldt;
  size_t off_ab0, off_ab, off_tau, off_prog, off_mail, off_pmail, off_retired, off_ctl, off_T, off_qprog, off_offs, total;
  size_t nrec;
  explicit Layout(int n, bool with_records = true) {
    nsweeps = n > 2 ? n - 2 : 0;
    nS = ceil_div((nsweeps > 0 ? nsweeps : 1) + 1, QG);   // block S = sweeps 32 S - 1 .. 32 S + 30
    kmax = q2_groups_of_block(n, 0); if (kmax < 1) kmax = 1;
    ldt = kmax + 1;
    size_t o = 0;
    off_ab0 = o; o += al256((size_t)LDAB * (round_up(n + 1, 128)) * 8);   // the band as packed (and, on a team, gathered)
    off_ab = o; o += al256((size_t)LDAB * (n + 1) * 8);        // the band the chase works in
    off_tau = o; o += al256((size_t)ldt * (nsweeps + 1) * 8);
    off_prog = o; o += al256((size_t)(nsweeps + 1) * 4);
    off_mail = o; o += al256((size_t)4 * (kmax + 1) * MAILW * 8);
    off_pmail = o; o += al256((size_t)2 * 4 * (kmax + 2) * PMAILW * 8);   // chase_pos_kernel: forward and backward lines
    off_retired = o; o += al256((size_t)(kmax + 2) * 4);
    off_ctl = o; o += 256;
    nrec = 0;
    for (int S = 0; S < nS; ++S) nrec += (size_t)q2_groups_of_block(n, S);
    off_T = o; o += with_records ? al256((nrec + 1) * QREC * 8) : 0;     // (without: the caller lends the records' array)
    off_qprog = o; o += al256((size_t)nS * ceil_div(n, QNC) * 4);
    off_offs = o; o += al256((size_t)(nS + 1) * 4);
    total = o;
  }
};

// how many 512-thread workgroups of chase_pos_kernel the device holds at once (they must all be resident)
int pos_capacity() {
  static int cap = -1;
  if (cap < 0) {
    int per_cu = 0, dev = 0;
    hipDeviceProp_t pr;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&pr, dev) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, chase_pos_kernel, 512, 0) != hipSuccess)
      cap = 0;
    else {
      if (per_cu > 2) per_cu = 2;                          // two per CU is what has been measured
      cap = per_cu * pr.multiProcessorCount;
    }
  }
  return cap;
}

}  // namespace

size_t sb2st_work_bytes(int n, bool with_records) { return Layout(n, with_records).total; }
size_t sb2st_record_bytes(int n) { return al256((Layout(n).nrec + 1) * QREC * 8); }

// Band (lower band of A, half bandwidth 64) -> d, e; the reflectors go to V2 (n x n, ldv2, zero on
// entry; column s = the reflectors of sweep s stacked) and into the workspace (tau).  *d_flag |= 4
// if the persistent kernel had to be abandoned (a bounded spin ran out).
double *sb2st_band(void *work, int n) { return (double *)((char *)work + Layout(n, false).off_ab0); }   // (in front of the records in both layouts)
void pack_band(hipStream_t s, int n, const double *A, int lda, double *AB) {
  if (n <= 0) return;
  hipLaunchKernelGGL(pack_band_kernel, dim3(n), dim3(LDAB), 0, s, n, A, lda, AB);
}

void sb2st_lower(hipStream_t s, int n, const double *A, int lda, double *d, double *e, double *V2, int ldv2,
                 int *d_flag, void *work, bool band_packed, int chase_mode) {
  if (n <= 0) return;
  const Layout L(n, false);     // (nothing here touches the records or what lies behind them)
  char *w = (char *)work;
  double *AB0 = (double *)(w + L.off_ab0);
  double *AB = (double *)(w + L.off_ab), *tau2 = (double *)(w + L.off_tau);
  unsigned *prog = (unsigned *)(w + L.off_prog), *ctl = (unsigned *)(w + L.off_ctl);
  double *mail = (double *)(w + L.off_mail), *pmail = (double *)(w + L.off_pmail);
  unsigned *retired = (unsigned *)(w + L.off_retired);
  const int nmail = 4 * (L.kmax + 1) * MAILW, npmail = 2 * 4 * (L.kmax + 2) * PMAILW;
  if (!band_packed) hipLaunchKernelGGL(pack_band_kernel, dim3(n), dim3(LDAB), 0, s, n, A, lda, AB0);
  (void)hipMemcpyAsync(AB, AB0, (size_t)LDAB * n * 8, hipMemcpyDeviceToDevice, s);
  (void)hipMemsetAsync(tau2, 0, (size_t)L.ldt * (L.nsweeps + 1) * 8, s);
  (void)hipMemsetAsync(prog, 0, (size_t)(L.nsweeps + 1) * 4 + 0, s);
  (void)hipMemsetAsync(ctl, 0, 256, s);
  if (L.nsweeps > 0) {
    // which kernel: 1 = sweeps through memory (chase_kernel), 2 = positions in registers (chase_pos_kernel, with
    // chase_kernel behind it in case its workgroups cannot all become resident); default 2 where the chip holds K0
    int mode = 2;
    static int env_mode = -1;
    if (env_mode < 0) { const char *ev = getenv("EK_SB2ST_CHASE"); env_mode = ev ? atoi(ev) : 0; }
    if (env_mode > 0) mode = env_mode;
    if (chase_mode > 0) mode = chase_mode;
    const int K0 = (n - 3) / SB + 1;
    const bool pos = mode == 2 && K0 <= pos_capacity() && !getenv("EK_SB2ST_WGS") && !getenv("EK_SB2ST_PROF");
    kprof_begin(s, kProfChase);
    if (pos) {
      hipLaunchKernelGGL(mail_init_kernel, dim3(ceil_div(npmail, 256)), dim3(256), 0, s, pmail, npmail);
      (void)hipMemsetAsync(retired, 0, (size_t)(L.kmax + 2) * 4, s);
      hipLaunchKernelGGL(set_word_kernel, dim3(1), dim3(1), 0, s, retired + K0, 1u);   // there is no position K0
      // neighbours on one XCD (EK_SB2ST_XCDMAP=1) buy nothing at one workgroup per CU (N = 16384: 52.0 vs 51.6 ms) and
      // cost at two (N = 32768: 143 vs 125 ms: the two workgroups of a CU are then close in the pipeline and busy together)
      int per = 0;
      if (const char *ev = getenv("EK_SB2ST_XCDMAP")) { if (atoi(ev) != 0) per = ceil_div(K0, 8); }
      unsigned census = 1u << 16;                            // x ~0.3 us: what a workgroup waits for the others to arrive
      if (const char *ev = getenv("EK_SB2ST_CENSUS_SPINS")) census = (unsigned)atoi(ev);
      PosArgs a{n, K0, AB, V2, ldv2, tau2, L.ldt, pmail, pmail + (size_t)4 * (L.kmax + 2) * PMAILW, retired, ctl, per, census, d_flag,
                getenv("EK_SB2ST_JITTER") ? (unsigned)atoi(getenv("EK_SB2ST_JITTER")) : 0u, nullptr, -1};
#ifdef EK_POS_TRACE
      static long long *trace_buf = nullptr;
      if (const char *ev = getenv("EK_SB2ST_TRACE")) {
        if (!trace_buf) (void)hipMalloc((void **)&trace_buf, 2048 * 16 * sizeof(long long));
        (void)hipMemsetAsync(trace_buf, 0, 2048 * 16 * sizeof(long long), s);
        a.trace = trace_buf; a.trace_k = atoi(ev);
      }
#endif
      hipLaunchKernelGGL(chase_pos_kernel, dim3(per > 0 ? per * 8 : K0), dim3(512), 0, s, a);
      hipLaunchKernelGGL(repack_band_kernel, dim3(n), dim3(LDAB), 0, s, n, AB0, AB, ctl);
#ifdef EK_POS_TRACE
      if (a.trace) {                                         // diagnostic: average gaps between the stamps, sweeps 256 .. 2047
        static long long h[2048 * 16];
        (void)hipMemcpyAsync(h, a.trace, sizeof(h), hipMemcpyDeviceToHost, s);
        (void)hipStreamSynchronize(s);
        double acc[16] = {0}; int cnt = 0;
        for (int q = 256; q + 1 < 2048 && q + 1 <= L.nsweeps - 64 * (a.trace_k + 2); ++q) {
          const long long *r = h + 16 * q, *r1 = h + 16 * (q + 1);
          if (!r[0] || !r1[0]) continue;
          acc[0] += (double)(r1[0] - r[0]);                 // period
          for (int j = 1; j < 16; ++j) acc[j] += r[j] ? (double)(r[j] - r[0]) : 0.0;
          ++cnt;
        }
        if (cnt)
          fprintf(stderr, "[sb2st trace] position %d, %d sweeps: period %.0f ns; from the top of a sweep: forward mail taken %.0f, "
                  "backward mail taken %.0f, barrier #1 passed %.0f, #2 passed %.0f, backward sent %.0f, forward sent %.0f, "
                  "D and B H done %.0f, #3 passed %.0f, H B done %.0f, #4 passed %.0f, shifted %.0f ns\n",
                  a.trace_k, cnt, 10 * acc[0] / cnt, 10 * acc[1] / cnt, 10 * acc[2] / cnt, 10 * acc[3] / cnt, 10 * acc[7] / cnt,
                  10 * acc[5] / cnt, 10 * acc[4] / cnt, 10 * acc[8] / cnt, 10 * acc[9] / cnt, 10 * acc[6] / cnt, 10 * acc[10] / cnt,
                  10 * acc[11] / cnt);
      }
#endif
    }
    hipLaunchKernelGGL(mail_init_kernel, dim3(ceil_div(nmail, 256)), dim3(256), 0, s, mail, nmail);
    ChaseArgs c{n, L.nsweeps, AB, V2, ldv2, tau2, L.ldt, prog, ctl, mail, L.kmax + 1, nullptr, pos ? 1 : 0, d_flag};
    if (getenv("EK_SB2ST_PROF")) { c.prof = (long long *)(ctl + 16); }
    // enough workgroups for the pipeline (a sweep starts one task behind its predecessor)
    int nwg = n / SB + 8;
    if (nwg > 256) nwg = 256;
    if (nwg > L.nsweeps) nwg = L.nsweeps;
    if (const char *ev = getenv("EK_SB2ST_WGS")) { const int v = atoi(ev); if (v > 0) nwg = v; }
    static int nw = -1;
    if (nw < 0) { const char *ev = getenv("EK_SB2ST_WAVES"); nw = ev ? atoi(ev) : 8; }
    if (nw == 16) hipLaunchKernelGGL(chase_kernel<16>, dim3(nwg), dim3(1024), 0, s, c);
    else if (nw == 8) hipLaunchKernelGGL(chase_kernel<8>, dim3(nwg), dim3(512), 0, s, c);
    else hipLaunchKernelGGL(chase_kernel<4>, dim3(nwg), dim3(256), 0, s, c);
    kprof_end(s, kProfChase);
  }
  hipLaunchKernelGGL(unpack_de_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, s, n, AB, d, e);
  if (getenv("EK_SB2ST_PROF")) {
    long long h[8];
    (void)hipMemcpyAsync(h, ctl + 16, sizeof(h), hipMemcpyDeviceToHost, s);
    (void)hipStreamSynchronize(s);
    fprintf(stderr, "[sb2st prof] tasks %lld; cycles per task: gate+loads %.0f, reflector %.0f, left-apply+fill %.0f, "
            "partials %.0f, updates %.0f\n", h[6], (double)h[0] / h[6], (double)h[1] / h[6], (double)h[2] / h[6],
            (double)h[3] / h[6], (double)h[4] / h[6]);
  }
  if (getenv("EK_SB2ST_VERBOSE")) {
    unsigned h[8];
    (void)hipMemcpyAsync(h, ctl, sizeof(h), hipMemcpyDeviceToHost, s);
    (void)hipStreamSynchronize(s);
    fprintf(stderr, "[sb2st] n %d: abort %u, positions abandoned %u, census %u\n", n, h[1], h[3], h[4]);
  }
  hipLaunchKernelGGL(forward_abort_kernel, dim3(1), dim3(1), 0, s, ctl, d_flag);   // abort word -> caller's flag
}

// Z(:, 0:ncols) <- Q2 Z with the reflectors left by sb2st_lower (same V2, same workspace)
void sb2st_apply_q2(hipStream_t s, int n, int ncols, const double *V2, int ldv2, double *Z, int ldz, int *d_flag,
                    void *work, double *records) {
  if (n <= 2 || ncols <= 0) return;
  const Layout L(n, records == nullptr);
  char *w = (char *)work;
  const double *tau2 = (const double *)(w + L.off_tau);
  double *Rec = records ? records : (double *)(w + L.off_T);
  unsigned *ctl = (unsigned *)(w + L.off_ctl), *qprog = (unsigned *)(w + L.off_qprog);
  unsigned *offS = (unsigned *)(w + L.off_offs);
  hipLaunchKernelGGL(q2_offsets_kernel, dim3(1), dim3(64), 0, s, n, L.nS, offS);
  Q2Geom g{n, L.nsweeps, L.nS, L.kmax, offS};
  hipLaunchKernelGGL(q2_tfactor_kernel, dim3(L.kmax, L.nS), dim3(256), 0, s, g, V2, ldv2, tau2, L.ldt, Rec);
  constexpr size_t lds = (size_t)(QOPSZ + 4 * 16 * QSTLD) * sizeof(double);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void *)q2_apply_nb_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void *)q2_apply_nb_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void *)q2_apply_nb_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr = true;
  }
  // blocks of sweeps per pass: 4 from order 8192 on, 3 below (N = 16384: 0.194 / 0.199 / 0.212 s with 4 / 3 / 2,
  // N = 4096: 4.9 / 4.7 / 4.9 ms); EK_Q2_NBLK = 2, 3, 4 forces one (all give the same bits)
  int nblk = (n >= 8192) ? 4 : 3;
  if (const char *ev = getenv("EK_Q2_NBLK")) { const int v = atoi(ev); if (v >= 2 && v <= 4) nblk = v; }
  const int per = nblk;
  const int nslab = ceil_div(ncols, QNC), npair = ceil_div(L.nS, per), npass = npair * nslab;
  (void)hipMemsetAsync(qprog, 0, (size_t)npass * 4, s);
  (void)hipMemsetAsync(ctl + 2, 0, 4, s);
  // extra: how many further steps a pass stays behind its predecessor in the slab (fetch distance).  With few slabs
  // the passes of a slab are one dependent chain and every step of distance is paid npair times: 0 / 1 / 2 / 3 give
  // 21.0 / 22.2 / 24.2 / 26.2 ms for 1024 columns at N = 16384, 5.1 / 5.4 / 5.6 / 6.0 ms at N = 4096, and the same
  // 227 ms for all 16384 columns.
  Q2ApplyArgs a{g, Rec, Z, ldz, ncols, nslab, npass, npair, qprog, ctl, 0};
  if (const char *ev = getenv("EK_Q2_EXTRA")) a.extra = atoi(ev);
  int nwg = 512;
  if (const char *ev = getenv("EK_Q2_WGS")) { const int v = atoi(ev); if (v > 0) nwg = v; }
  if (nwg > npass) nwg = npass;
  kprof_begin(s, kProfQ2Apply);
  if (nblk == 3) hipLaunchKernelGGL(q2_apply_nb_kernel<3>, dim3(nwg), dim3(256), lds, s, a);
  else if (nblk == 4) hipLaunchKernelGGL(q2_apply_nb_kernel<4>, dim3(nwg), dim3(256), lds, s, a);
  else hipLaunchKernelGGL(q2_apply_nb_kernel<2>, dim3(nwg), dim3(256), lds, s, a);
  kprof_end(s, kProfQ2Apply);
  if (d_flag) hipLaunchKernelGGL(forward_abort_kernel, dim3(1), dim3(1), 0, s, ctl, d_flag);
}

}  // namespace ek
