// ek_common.h -- shared declarations for libek_hip.so (gfx950 / MI355X only).
//
// Internal C++ interface between the translation units of the library; the public C-ABI
// is include/ek_hip.h.  All matrices are column-major fp64 in device memory.
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>
#include <cstdio>

#define EK_HIP_CHECK(expr)                                                              \
  do {                                                                                  \
    hipError_t _e = (expr);                                                             \
    if (_e != hipSuccess) {                                                             \
      fprintf(stderr, "[ek_hip] %s failed at %s:%d: %s\n", #expr, __FILE__, __LINE__,   \
              hipGetErrorString(_e));                                                   \
      return -1000 - (int)_e;                                                           \
    }                                                                                   \
  } while (0)

namespace ek {

constexpr int kWave = 64;

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
inline int round_up(int a, int b) { return ceil_div(a, b) * b; }

// ---------------------------------------------------------------- GEMM (ek_gemm.hip)
// C = alpha * op(A) * op(B) + beta * C, op(X) = X or X^T, batched over `batch` problems
// with element strides.  lower_only: skip 128x128 tiles strictly above the diagonal
// (SYRK / SYR2K style updates where only the lower triangle is referenced).
struct GemmDesc {
  int M, N, K;
  bool transA, transB;
  double alpha, beta;
  const double *A; int lda; long long strideA;
  const double *B; int ldb; long long strideB;
  double *C; int ldc; long long strideC;
  int batch;
  bool lower_only;
  bool staged_rank_k = false;          // op(A) op(B)^T with K <= 256: the staged 8-wave rank-k kernel (one workgroup per
                                       // CU: slower alone than the default, but leaves room for a second stream's work)
  const long long *d_offs = nullptr;   // device: per-batch element offsets {A, B, C} (added to strides)
  const int *d_dims = nullptr;         // device: per-batch {M, N, K}; host M, N, K are then upper bounds
  bool even_offs = false;              // the caller's promise that every d_offs entry for A and B is even (16-byte loads stay aligned)
  bool small_tiles = false;            // lower_only only: the caller does not need whole 128 x 128 diagonal tiles written -- the
                                       // 64 x 64 tiling may serve (short trailing updates: four times the workgroups, a quarter of the work each)
};
void gemm(hipStream_t s, const GemmDesc &g);

inline void gemm(hipStream_t s, bool ta, bool tb, int M, int N, int K, double alpha,
                 const double *A, int lda, const double *B, int ldb, double beta, double *C,
                 int ldc, bool lower_only = false, bool staged_rank_k = false, bool small_tiles = false) {
  GemmDesc g{M, N, K, ta, tb, alpha, beta, A, lda, 0, B, ldb, 0, C, ldc, 0, 1, lower_only, staged_rank_k};
  g.small_tiles = small_tiles;
  gemm(s, g);
}

// ---------------------------------------------------------------- kernel timing (ek_util.hip)
// Optional HIP-event brackets around the launches of the kernels bench.py reports a roofline for, on
// the stream they are launched on.  Off by default (an event pair costs a few microseconds of host time).
enum { kProfQ2Apply = 0, kProfChase = 1, kProfSymm = 2, kProfSyr2k = 3, kProfCount = 4 };
void kprof_enable(bool on);
bool kprof_enabled();
void kprof_begin(hipStream_t s, int id);
void kprof_end(hipStream_t s, int id);
void kprof_collect(double *seconds /* kProfCount */, long long *launches /* kProfCount */);   // after a stream sync; resets

// ---------------------------------------------------------------- small utilities (ek_util.hip)
void copy_matrix(hipStream_t s, int m, int n, const double *src, int lds, double *dst, int ldd);
void set_matrix(hipStream_t s, int m, int n, double offdiag, double diag, double *A, int lda);
void symmetrize_lower(hipStream_t s, int n, double *A, int lda);   // upper <- lower^T
// dst (n x m, ldd) <- (src(r0 : r0+m, 0 : n))^T
void transpose_rows(hipStream_t s, int n, int m, const double *src, int lds, int r0, double *dst, int ldd);
void gather_columns(hipStream_t s, int m, int n, const double *src, int lds, const int *perm,
                    double *dst, int ldd);                         // dst(:,j) = src(:,perm[j])

// local piece of a block-cyclic matrix: dst (mr x nc local) <- src (global), owner (me_r, me_c)
void gather_block_cyclic(hipStream_t s, int mr, int nc, const double *src, int lds, int nb, int pr,
                         int me_r, int pc, int me_c, double *dst, int ldd);

// global dst <- the local piece src (mr x nc, lds) of owner (me_r, me_c)
void scatter_block_cyclic(hipStream_t s, int mr, int nc, const double *src, int lds, int nb, int pr,
                          int me_r, int pc, int me_c, double *dst, int ldd);

void maxabs_lower(hipStream_t s, int n, const double *A, int lda, double *partial /* 512 */, int bw = 1 << 30);
void scale_lower(hipStream_t s, int n, double alpha, double *A, int lda);
void scale_vector(hipStream_t s, int n, double alpha, double *x);

// ---------------------------------------------------------------- Cholesky & triangular (ek_chol.hip)
constexpr int kDiagNB = 128;   // order of the diagonal blocks factored/inverted by one workgroup
// B = L L^T (lower). invdiag (optional, ld = kDiagNB, ceil(n/128) blocks of 128x128) receives the
// explicit inverses of the diagonal blocks of L.  *d_info (device int, must be 0 on entry)
// receives the LAPACK info (first non-positive pivot, 1-based).
void potrf_lower(hipStream_t s, int n, double *B, int ldb, double *invdiag, int *d_info,
                 double *work /* >= 128 * n doubles */);
// inverses of the 128x128 diagonal blocks of a given lower-triangular L
void trtri_diag_blocks(hipStream_t s, int n, const double *L, int ldl, double *invdiag);
void trsm_rlt(hipStream_t s, int m, int n, const double *L, int ldl, const double *invdiag,
              double *X, int ldx, double *work);   // X <- X L^-T   (X m x n, L n x n)
void trsm_lln(hipStream_t s, int n, int m, const double *L, int ldl, const double *invdiag,
              double *X, int ldx, double *work);   // X <- L^-1 X   (X n x m)
void trsm_llt(hipStream_t s, int n, int m, const double *L, int ldl, const double *invdiag,
              double *X, int ldx, double *work);   // X <- L^-T X   (X n x m)
// scratch: >= sygst_scratch_doubles(n) doubles
inline size_t sygst_scratch_doubles(int n) {
  const size_t h = (size_t)(n / 2 + 128);
  const size_t a = 2 * h * h, b = 2 * 128 * 128;
  return a > b ? a : b;
}
// Leaves of 256 for the three solves above (and the reduction below): inverses of the 256 x 256 diagonal blocks of L
// (n / 256 of them, ld 256) from L and its 128-block inverses; registered for the array `invdiag` they belong to until
// trsm_register_inv256(nullptr, nullptr, 0).  scratch: >= (n / 256) * 128 * 128 doubles.
void trtri256_blocks(hipStream_t s, int n, const double *L, int ldl, const double *invdiag, double *inv256, double *scratch);
void trsm_register_inv256(const double *invdiag, const double *inv256, int n);
void sygst_lower(hipStream_t s, int n, double *A, int lda, const double *L, int ldl,
                 const double *invdiag, double *work, double *scratch);
// Distributed form (PDSYGST on a 1 x P grid): both triangular solves are sharded by columns,
//   Y(:, C_r) = L^-1 A(:, C_r)   on the rank's contiguous column block C_r, one all-gather of Y,
//   A'(:, S)  = L^-1 (Y(S, :))^T on every strip S the rank owns (r, r+P, ...: the strips the
//                                distributed tridiagonalisation reads on this rank; A' = A'^T),
// 2 n^3 / P flops per rank and no second exchange: on return a member's A holds the reduced matrix
// in the columns of its OWN strips only.  scratch: >= sygst_dist_scratch_doubles(n, ld, P).
struct SygstMember {
  double *A; int lda; const double *L; int ldl; const double *invdiag;
  double *work;      // >= 128 * lda doubles
  double *scratch;
  int rank;
};
// Right-looking form with look-ahead on a second stream (single GPU): same results contract as
// potrf_lower; work: >= potrf_rl_work_bytes(n, ldb).
size_t potrf_rl_work_bytes(int n, int ld);
void potrf_lower_rl(hipStream_t s, hipStream_t s2, int n, double *B, int ldb, double *invdiag, int *d_info,
                    void *work);
struct SytrdExchange;
// Distributed Cholesky (PDPOTRF on a 1 x P grid, 128-wide column blocks, right-looking): the owner
// of strip k factors its diagonal block in LDS, solves the panel below it and broadcasts
// [inverse | diagonal block | panel] in one message; every rank stores it (so L and the block
// inverses end up complete on all ranks, as the two solves of the reduction need them) and
// updates the strips it owns.  n^3 / (3P) flops per rank, n^2/2 doubles on the wire in total.
// *d_info receives the first failing pivot on every rank (one small all-reduce at the end).
struct PotrfMember { double *B; int ldb; double *invdiag; int *d_info; void *work; int rank; };
size_t potrf_dist_work_bytes(int n, int ld, int nranks);
// s2: second stream (look-ahead: the next strip's chain and broadcast beside the rest of the update; nullptr: none)
void potrf_lower_dist(hipStream_t s, hipStream_t s2, int n, int nmem, const PotrfMember *mem, const SytrdExchange &x);
void potrf_dist_set_lookahead(int on);               // 0 off, 1 on, -1 default (on)
void potrf_dist_profile(bool on);                    // per-strip HIP events (tools/team_timing.py)
void potrf_dist_profile_collect(double *seconds, int P);   // [0] chains, [1] rest-of-update sections, [2] their cost to a rank of P with look-ahead; after a sync
size_t sygst_dist_scratch_doubles(int n, int ld, int nranks);
void sygst_lower_dist(hipStream_t s, int n, int nmem, const SygstMember *mem, const SytrdExchange &x);

// ---------------------------------------------------------------- tridiagonalisation (ek_sytrd.hip)
struct SytrdWork;   // opaque, sized by sytrd_work_bytes
size_t sytrd_work_bytes(int n);
// A (lower, ld even, base 16-B aligned, padded to a multiple of 128 rows/cols with
// finite values) -> d(n), e(n-1), tau(n-1); V (n x n, ldv) receives the explicit unit
// lower-trapezoidal reflector matrix (column j = v_j, zeros above row j+1); the scaled
// reflectors are also written below the sub-diagonal of A as PDSYTRD does.
void sytrd_lower(hipStream_t s, int n, double *A, int lda, double *d, double *e, double *tau,
                 double *V, int ldv, void *work);

// Distributed form (PDSYTRD on a 1 x P grid with 128-wide column blocks): member
// `rank` of a team of `nranks` owns the strips rank, rank+P, ... of the trailing matrix.  Every
// member passes a full-size A (identical on entry; only the owned strips are kept current by the
// trailing updates), streams only its own strips in symv, and the team all-reduces one window of
// at most 2*npad+1 doubles per column.  d, e, tau, V and the reflector columns of A come out
// complete and bit-identical on every member.  A process holds `nmem` members: 1 in production
// (exchange = ncclAllReduce over RCCL), the whole team in the single-GPU rehearsal
// (exchange = sytrd_team_allreduce).
constexpr int kMaxTeam = 16;
struct SytrdMember {
  double *A; int lda;
  double *d, *e, *tau;
  double *V; int ldv;
  void *work;          // >= sytrd_dist_work_bytes(n, nranks)
  int rank;
};
struct SytrdExchange {
  int nranks;
  // in-place sum over ALL ranks of the team of `count` doubles; bufs = the windows of the nmem
  // members held by this process; stream-ordered
  void (*allreduce)(hipStream_t s, int nmem, double *const *bufs, size_t count, void *user);
  void *user;
  // in-place all-gather of unequal pieces: rank r's piece [offs[r], offs[r] + counts[r]) of its
  // array `bufs` ends up at the same place in every rank's array (offs, counts: nranks entries;
  // members held by this process are the ranks rank0 .. rank0 + nmem - 1)
  void (*allgatherv)(hipStream_t s, int nmem, int rank0, double *const *bufs, const size_t *offs,
                     const size_t *counts, int nranks, void *user) = nullptr;
};
size_t sytrd_dist_work_bytes(int n, int nranks);
void sytrd_lower_dist(hipStream_t s, int n, int nmem, const SytrdMember *mem, const SytrdExchange &x);
void sytrd_team_allreduce(hipStream_t s, int nmem, double *const *bufs, size_t count, void *user);
void team_allgatherv(hipStream_t s, int nmem, int rank0, double *const *bufs, const size_t *offs,
                     const size_t *counts, int nranks, void *user);   // rehearsal: nmem == nranks

void sytrd_debug_split(void *alt, int mask);   // placement experiments only
void sytrd_set_max_cols(int max_cols);   // tuning hooks: stop after this many columns (-1 = all)
int sytrd_get_max_cols();
// instrumentation: HIP events around every symv launch (bench.py roofline line)
void symv_profile_enable(int stride);   // 0 = off, k = time every k-th column's launch
void symv_profile_collect(double *seconds, long long *launches, double *bytes);

// ---------------------------------------------------------------- two-stage tridiagonalisation
// (ek_sy2sb.hip, ek_sb2st.hip): dense -> band (half bandwidth kBandW) on the matrix cores, band ->
// tridiagonal by bulge chasing, and the back-transformations of both stages.  Used by the whole-path
// call for large orders in place of sytrd_lower + ormtr_lower; the results contract is the same.
constexpr int kBandW = 64;
size_t sy2sb_work_bytes(int n);
// A (lower, lda multiple of 128, zero padded) -> band in the lower band of A (A(i,j), 0 <= i-j <= 64;
// the rest of the lower triangle is zeroed except the R factors' upper triangles inside the band).
// Vall (n x n, ldv; must be zero on entry): explicit reflectors, column j = v_j with its unit entry
// at row j + 64; tau1[j] (must be zero on entry).  *d_flag (device int, 0 on entry): bits 8.. count the panels
// CholeskyQR2 could not factor and the Householder rescue did (informational); the low byte stays 0.
// s2: second stream, the panel factorisation of the next panel runs on it beside the trailing update.
void sy2sb_lower(hipStream_t s, hipStream_t s2, int n, double *A, int lda, double *Vall, int ldv, double *tau1,
                 int *d_flag, void *work);

// Team form (1 x P, 128-wide column strips, strip S on rank S mod P): a member passes its copy of the matrix, of which
// only the columns of its own strips need to be valid (and only they are kept current); V, tau1 come out complete and
// identical on every member, the band stays in the members' own strips (see ek_sy2sb.hip).
struct Sy2sbMember { double *A; int lda; double *Vall; int ldv; double *tau1; int *d_flag; void *work; int rank; };
size_t sy2sb_dist_work_bytes(int n, int nranks);
// s2: second stream (look-ahead: the chain and the broadcast of panel p + 1 beside the rest of update p; nullptr: none)
void sy2sb_lower_dist(hipStream_t s, hipStream_t s2, int n, int nmem, const Sy2sbMember *mem, const SytrdExchange &x);
void sy2sb_dist_set_lookahead(int min_rows);         // rows from which the team form looks ahead (0: never, -1: default)
void sy2sb_dist_profile(bool on);                    // per-panel HIP events (tools/team_timing.py)
void sy2sb_dist_profile_collect(double *seconds, int P);   // [0] chains, [1] rest-of-update sections, [2] their cost to a rank of P with look-ahead; after a sync

// with_records = false: without the compact-WY records of the bulge chasing's reflectors (sb2st_record_bytes(n), the one
// large part): the caller then lends sb2st_apply_q2 an array for them that only has to live while Q2 is applied
size_t sb2st_work_bytes(int n, bool with_records = true);
size_t sb2st_record_bytes(int n);
// Band (lower band of A, half bandwidth kBandW) -> d(n), e(n-1) by bulge chasing.  V2 (n x n, ldv2,
// zero on entry) receives the reflectors (column s = those of sweep s, stacked); *d_flag |= 4 if the
// persistent kernel had to be abandoned.  work: >= sb2st_work_bytes(n), shared with sb2st_apply_q2.
// band_packed: the band already lies in sb2st_band(work, n) (kBandLd x n, column c = the 65 diagonals of column c followed
// by zeros: pack_band, then -- on a team whose members hold only their own strips -- an all-gather of its columns)
constexpr int kBandLd = 2 * kBandW;
double *sb2st_band(void *work, int n);
void pack_band(hipStream_t s, int n, const double *A, int lda, double *AB /* kBandLd x n */);
// chase_mode: 0 = default (positions in registers where the chip holds them all), 1 = sweeps through memory only (what
// the whole-path call repeats the stage with after an abandoned wait)
void sb2st_lower(hipStream_t s, int n, const double *A, int lda, double *d, double *e, double *V2, int ldv2,
                 int *d_flag, void *work, bool band_packed = false, int chase_mode = 0);
// Z(:, 0:ncols) <- Q2 Z (Z 16-byte aligned, ldz even); *d_flag |= 4 if the pipeline had to be abandoned
void sb2st_apply_q2(hipStream_t s, int n, int ncols, const double *V2, int ldv2, double *Z, int ldz, int *d_flag,
                    void *work, double *records = nullptr);

// ---------------------------------------------------------------- tridiagonal D&C (ek_stedc.hip)
size_t stedc_work_bytes(int n, int nsel = -1);     // nsel: eigenvector columns wanted (-1: all); <= n - n/2: compact bases
bool stedc_compact(int n, int nsel);               // whether stedc keeps its bases compact for that selection (no wscratch, Z n x nsel)
// d(n), e(n-1) -> eigenvalues ascending in w(n), eigenvectors in Z (n x n, ldz).
// With a selection only the eigenvectors of ranks r(l) = ((l / nb) * npcol + mycol) * nb + l % nb,
// l = 0..nsel-1, are formed (the block-cyclic share of a process column; nb >= n, npcol = 1 gives
// the lowest nsel) and returned in columns 0..nsel-1 of Z: the top-level merge, two thirds of the
// D&C flops, then multiplies only those columns.  All n eigenvalues are always returned.
struct StedcSelect { int nsel, nb, npcol, mycol; };
// d_flops (optional, device): receives the flops of the merge products this solve executed (after deflation)
// wscratch (optional): an n x n array with leading dimension ldz for the permuted bases of the merges; without it Z
// itself serves (and must then be n x n even where only sel->nsel columns are wanted)
// Team form (the reference's PDSTEDC is distributed over its grid, solver_scalapack_all.f90:96): the
// `levels` heights of the tree right below the top merge are sharded over a 1 x P team as well.  Everything that is
// O(n^2) per height (rank sorts, deflation, secular equation, the permuted bases) stays replicated and bit-identical; the
// products Q = W S -- all of the O(n^3) -- are cut into 128-wide strips of the compact basis array (n x (n - n/2), the
// second half's blocks stored n/2 columns to the left), strip S on rank S mod P like every other strip of the library, so
// that what deflation takes away is taken from every rank alike; a rank forms the S columns and the products of its own
// strips only, then ONE in-place all-gather round per P strips completes the height's basis on every rank (whole
// columns of the compact array: 8 n (n - n/2) bytes on the wire per sharded height).  Same GEMM per output element as on one
// GPU (K walked in the same order), so the team's eigenvectors are the 1 x 1 result bit for bit.  Needs the compact
// bases (a team member's share of the columns is at most half of them).
//   rank >= 0: this process is that rank, x is its exchange;  rank == -1: rehearsal on one GPU -- the process plays every
//   rank's strips in turn (no exchange; the per-rank sections are timed when stedc_team_profile is on).
struct StedcTeam { int nranks; int rank; const SytrdExchange *x; int levels; };
int stedc_team_levels(int n, int nranks);          // default number of sharded heights for that order and team (0: none)
void stedc_team_set_levels(int levels);            // test / tool hook: force that many (-1: default)
void stedc_team_profile(bool on);                  // HIP events around the call and around every rank's sections
// after a stream sync: [0] the whole call, [1] all ranks' sections, [2] sum over the heights of the longest rank's section
// (a rank of a real team spends [0] - [1] + [2] computing); resets
void stedc_team_profile_collect(double *seconds);
void stedc(hipStream_t s, int n, const double *d, const double *e, double *w, double *Z,
           int ldz, void *work, int *d_info, const StedcSelect *sel = nullptr, double *d_flops = nullptr,
           double *wscratch = nullptr, const StedcTeam *team = nullptr);

// ---------------------------------------------------------------- back-transformation (ek_ormtr.hip)
size_t ormtr_work_bytes(int n, int ncols, int ncols_global = -1);   // ncols_global: columns of the whole Z (a grid cell holds ncols of them)
// Z(:, 0:ncols) <- Q Z with Q = H(0)...H(n-2) given by explicit V (see sytrd_lower) and tau.
void ormtr_lower(hipStream_t s, int n, int ncols, const double *V, int ldv, const double *tau,
                 double *Z, int ldz, void *work, int ncols_global = -1);
// the same in two parts: the T factors of the block reflectors (independent of Z; prep >= ormtr_prep_bytes(n)),
// then their application (work >= ormtr_work_bytes(n, ncols) - ormtr_prep_bytes(n))
size_t ormtr_prep_bytes(int n);
void ormtr_prepare(hipStream_t s, int n, const double *V, int ldv, const double *tau, void *prep);
void ormtr_apply(hipStream_t s, int n, int ncols, const double *V, int ldv, const void *prep, double *Z, int ldz,
                 void *work, int ncols_global = -1);
// explicit V from the PDSYTRD storage (reflectors below the sub-diagonal of A)
void build_explicit_v(hipStream_t s, int n, const double *A, int lda, double *V, int ldv);
// synthetic SPD generator of SURVEY.md 8(d) on the device
void synth_matrix(hipStream_t s, int n, unsigned long long seed, double *M, int ldm);

// ---------------------------------------------------------------- verifier / IPR (ek_verify.hip)
size_t verify_work_bytes(int n, int ncols);
void residual_norms(hipStream_t s, int n, int n_check, const double *A, int lda, const double *B, int ldb,
                    const double *w, const double *V, int ldv, double *d_result, void *work);
void orthogonality(hipStream_t s, int n, int c0, int nc, const double *B, int ldb, const double *V, int ldv,
                   double *d_result, void *work);
void ipratios(hipStream_t s, int n, int n_vec, const double *B, int ldb, const double *V, int ldv,
              double *d_ipr, void *work);

}  // namespace ek
