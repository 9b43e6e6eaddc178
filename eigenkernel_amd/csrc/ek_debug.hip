// ek_debug.hip -- tuning, profiling and rehearsal hooks (include/ek_hip_debug.h) and the *_team entries of
// include/ek_hip.h that rehearse a whole team inside one process.  Not part of the drop-in surface.
#include "ek_api_internal.h"

using namespace ek;
using namespace ek::api;

extern "C" {

// PDPOTRF('L') on a 1 x P grid, see potrf_lower_dist.  nteam as in ek_hip_sytrd_team.  B_loc
// returns the first local member's factor; *mismatch the number of doubles (lower triangle of L
// and the block inverses) in which another local member differs from it.
int ek_hip_potrf_team(int n, double *B_loc, const int desc_B[9], int nteam, long long *mismatch) {
  if (n < 0) return -1;
  if (!B_loc && n > 0) return -2;
  int rc = check_desc(desc_B, 3, n, n); if (rc) return rc;
  if (nteam < 0 || nteam > kMaxTeam) return -4;
  rc = ensure_init(); if (rc) return rc;
  if (mismatch) *mismatch = 0;
  if (n == 0) return 0;
  std::lock_guard<std::mutex> lk(g_mu);
  if (nteam == 0 && !g_comm.on) return -4;
  hipStream_t s = g_ctx.stream;
  const int ld = pad_ld(n), nblk = ceil_div(n, kDiagNB);
  const int nmem = nteam > 0 ? nteam : 1, P = nteam > 0 ? nteam : g_comm.nranks;
  const size_t wb = potrf_dist_work_bytes(n, ld, P);
  const size_t per = al((size_t)ld * ld * 8) + al((size_t)nblk * kDiagNB * kDiagNB * 8) + al(wb) + 256;
  void *ws;
  rc = workspace(per * nmem + 256, &ws);
  if (rc) return rc;
  Arena a(ws, g_ctx.ws_bytes);
  unsigned long long *d_cnt = a.get<unsigned long long>(1);
  EK_HIP_CHECK(hipMemsetAsync(d_cnt, 0, 8, s));
  PotrfMember mem[kMaxTeam];
  for (int m = 0; m < nmem; ++m) {
    double *dB = a.get<double>((size_t)ld * ld);
    double *dInv = a.get<double>((size_t)nblk * kDiagNB * kDiagNB);
    char *work = a.get<char>(wb);
    int *dinfo = a.get<int>(1);
    EK_HIP_CHECK(hipMemsetAsync(dB, 0, (size_t)ld * ld * 8, s));
    EK_HIP_CHECK(hipMemsetAsync(dinfo, 0, sizeof(int), s));
    rc = h2d_matrix(n, n, B_loc, desc_B[8], dB, ld, s); if (rc) return rc;
    mem[m] = PotrfMember{dB, ld, dInv, dinfo, work, nteam > 0 ? m : g_comm.rank};
  }
  g_comm.err = 0;
  potrf_lower_dist(s, g_ctx.stream2, n, nmem, mem, team_exchange(nteam));
  EK_HIP_CHECK(hipGetLastError());
  for (int m = 1; m < nmem; ++m) {
    count_mismatch(s, n, n, mem[0].B, ld, mem[m].B, ld, 1, d_cnt);
    count_mismatch(s, nblk * kDiagNB * kDiagNB, 1, mem[0].invdiag, 1, mem[m].invdiag, 1, 0, d_cnt);
  }
  rc = d2h_matrix(n, n, mem[0].B, ld, B_loc, desc_B[8], s); if (rc) return rc;
  int infos[kMaxTeam] = {0};
  for (int m = 0; m < nmem; ++m)
    EK_HIP_CHECK(hipMemcpyAsync(&infos[m], mem[m].d_info, sizeof(int), hipMemcpyDeviceToHost, s));
  unsigned long long cnt = 0;
  EK_HIP_CHECK(hipMemcpyAsync(&cnt, d_cnt, 8, hipMemcpyDeviceToHost, s));
  EK_HIP_CHECK(hipStreamSynchronize(s));
  for (int m = 1; m < nmem; ++m) if (infos[m] != infos[0]) cnt += 1;   // info must be known to all
  if (mismatch) *mismatch = (long long)cnt;
  if (g_comm.err) { fprintf(stderr, "[ek_hip] RCCL exchange failed: %s\n", comm_error_string()); return -996; }
  return infos[0];
}

// PDSYGST(1,'L') on a 1 x P grid, see sygst_lower_dist.  nteam as in ek_hip_sytrd_team.  Every
// member leaves the reduced matrix in the columns of its own 128-wide strips; A_loc returns the
// lower triangle assembled from the owners (nteam >= 1) or this rank's own strips with the other
// columns untouched (nteam == 0).
int ek_hip_sygst_team(int n, double *A_loc, const int desc_A[9], const double *L_loc,
                      const int desc_B[9], int nteam) {
  if (n < 0) return -1;
  if (!A_loc && n > 0) return -2;
  int rc = check_desc(desc_A, 3, n, n); if (rc) return rc;
  if (!L_loc && n > 0) return -4;
  rc = check_desc(desc_B, 5, n, n); if (rc) return rc;
  if (nteam < 0 || nteam > kMaxTeam) return -6;
  rc = ensure_init(); if (rc) return rc;
  if (n == 0) return 0;
  std::lock_guard<std::mutex> lk(g_mu);
  if (nteam == 0 && !g_comm.on) return -6;
  hipStream_t s = g_ctx.stream;
  const int ld = pad_ld(n), nblk = ceil_div(n, kDiagNB);
  const int nmem = nteam > 0 ? nteam : 1, P = nteam > 0 ? nteam : g_comm.nranks;
  const size_t scr = sygst_dist_scratch_doubles(n, ld, P);
  const size_t per = al((size_t)ld * ld * 8) + al((size_t)128 * ld * 8) + al(scr * 8);
  void *ws;
  rc = workspace(al((size_t)ld * ld * 8) + al((size_t)nblk * kDiagNB * kDiagNB * 8) + per * nmem, &ws);
  if (rc) return rc;
  Arena a(ws, g_ctx.ws_bytes);
  double *dL = a.get<double>((size_t)ld * ld);
  double *dInv = a.get<double>((size_t)nblk * kDiagNB * kDiagNB);
  EK_HIP_CHECK(hipMemsetAsync(dL, 0, (size_t)ld * ld * 8, s));
  rc = h2d_matrix(n, n, L_loc, desc_B[8], dL, ld, s); if (rc) return rc;
  trtri_diag_blocks(s, n, dL, ld, dInv);
  SygstMember mem[kMaxTeam];
  for (int m = 0; m < nmem; ++m) {
    double *dA = a.get<double>((size_t)ld * ld);
    double *work = a.get<double>((size_t)128 * ld);
    double *scratch = a.get<double>(scr);
    EK_HIP_CHECK(hipMemsetAsync(dA, 0, (size_t)ld * ld * 8, s));
    rc = h2d_matrix(n, n, A_loc, desc_A[8], dA, ld, s); if (rc) return rc;
    mem[m] = SygstMember{dA, ld, dL, ld, dInv, work, scratch, nteam > 0 ? m : g_comm.rank};
  }
  g_comm.err = 0;
  sygst_lower_dist(s, n, nmem, mem, team_exchange(nteam));
  EK_HIP_CHECK(hipGetLastError());
  // strip S comes from its owner
  for (int S = 0; S * kDiagNB < n; ++S) {
    const int owner = S % P;
    const SygstMember *M = nullptr;
    for (int m = 0; m < nmem; ++m) if (mem[m].rank == owner) M = &mem[m];
    if (!M) continue;
    const int c0 = S * kDiagNB, cols = (n - c0 < kDiagNB) ? n - c0 : kDiagNB;
    rc = d2h_matrix(n, cols, M->A + (size_t)c0 * ld, ld, A_loc + (size_t)c0 * desc_A[8], desc_A[8], s);
    if (rc) return rc;
  }
  EK_HIP_CHECK(hipStreamSynchronize(s));
  if (g_comm.err) { fprintf(stderr, "[ek_hip] RCCL exchange failed: %s\n", comm_error_string()); return -996; }
  return 0;
}

// PDSYTRD on a 1 x P grid (column-block-cyclic, 128-wide blocks), see sytrd_lower_dist.
//   nteam >= 1: rehearsal of a whole team of nteam members inside this process on one GPU (each
//               member gets its own copy of A and its own workspace; exchange = a device kernel);
//   nteam == 0: this process is one member of the attached communicator (ek_hip_comm_init).
// A_loc/d/e/tau return the first local member's results; *mismatch (optional) the number of
// doubles (lower triangle of A, d, e, tau) in which any other local member differs from it.
int ek_hip_sytrd_team(int n, double *A_loc, const int desc_A[9], double *d, double *e, double *tau,
                      int nteam, long long *mismatch) {
  if (n < 0) return -1;
  if (!A_loc && n > 0) return -2;
  int rc = check_desc(desc_A, 3, n, n); if (rc) return rc;
  if (n > 0 && !d) return -4;
  if (n > 1 && !e) return -5;
  if (n > 1 && !tau) return -6;
  if (nteam < 0 || nteam > kMaxTeam) return -7;
  rc = ensure_init(); if (rc) return rc;
  if (mismatch) *mismatch = 0;
  if (n == 0) return 0;
  std::lock_guard<std::mutex> lk(g_mu);
  if (nteam == 0 && !g_comm.on) return -7;
  hipStream_t s = g_ctx.stream;
  const int ld = pad_ld(n);
  const int nmem = nteam > 0 ? nteam : 1, P = nteam > 0 ? nteam : g_comm.nranks;
  const size_t wb = sytrd_dist_work_bytes(n, P);
  const size_t per = al((size_t)ld * ld * 8) + al(wb) + 3 * al((size_t)ld * 8);
  void *ws;
  rc = workspace(per * nmem + 256, &ws);
  if (rc) return rc;
  Arena a(ws, g_ctx.ws_bytes);
  unsigned long long *d_cnt = a.get<unsigned long long>(1);
  EK_HIP_CHECK(hipMemsetAsync(d_cnt, 0, 8, s));
  SytrdMember mem[kMaxTeam];
  for (int m = 0; m < nmem; ++m) {
    double *dA = a.get<double>((size_t)ld * ld);
    char *work = a.get<char>(wb);
    double *dd = a.get<double>(ld), *de = a.get<double>(ld), *dt = a.get<double>(ld);
    EK_HIP_CHECK(hipMemsetAsync(dA, 0, (size_t)ld * ld * 8, s));
    EK_HIP_CHECK(hipMemsetAsync(dd, 0, 3 * al((size_t)ld * 8), s));
    rc = h2d_matrix(n, n, A_loc, desc_A[8], dA, ld, s); if (rc) return rc;
    mem[m] = SytrdMember{dA, ld, dd, de, dt, nullptr, 0, work, nteam > 0 ? m : g_comm.rank};
    const char *poison = getenv("EK_HIP_TEAM_POISON");
    if (poison && poison[0] == '1' && P > 1)
      poison_foreign_strips(s, n, dA, ld, P, mem[m].rank);
  }
  const SytrdExchange x = team_exchange(nteam, n);
  g_comm.err = 0;
  sytrd_lower_dist(s, n, nmem, mem, x);
  EK_HIP_CHECK(hipGetLastError());
  for (int m = 1; m < nmem; ++m) {
    count_mismatch(s, n, n, mem[0].A, ld, mem[m].A, ld, 1, d_cnt);
    count_mismatch(s, n, 1, mem[0].d, n, mem[m].d, n, 0, d_cnt);
    if (n > 1) {
      count_mismatch(s, n - 1, 1, mem[0].e, n, mem[m].e, n, 0, d_cnt);
      count_mismatch(s, n - 1, 1, mem[0].tau, n, mem[m].tau, n, 0, d_cnt);
    }
  }
  rc = d2h_matrix(n, n, mem[0].A, ld, A_loc, desc_A[8], s); if (rc) return rc;
  EK_HIP_CHECK(hipMemcpyAsync(d, mem[0].d, (size_t)n * 8, hipMemcpyDeviceToHost, s));
  if (n > 1) {
    EK_HIP_CHECK(hipMemcpyAsync(e, mem[0].e, (size_t)(n - 1) * 8, hipMemcpyDeviceToHost, s));
    EK_HIP_CHECK(hipMemcpyAsync(tau, mem[0].tau, (size_t)(n - 1) * 8, hipMemcpyDeviceToHost, s));
  }
  unsigned long long cnt = 0;
  EK_HIP_CHECK(hipMemcpyAsync(&cnt, d_cnt, 8, hipMemcpyDeviceToHost, s));
  EK_HIP_CHECK(hipStreamSynchronize(s));
  if (mismatch) *mismatch = (long long)cnt;
  if (g_comm.err) { fprintf(stderr, "[ek_hip] RCCL all-reduce failed: %s\n", comm_error_string()); return -996; }
  return 0;
}

int ek_hip_profile_symv(int enable) {
  int rc = ensure_init(); if (rc) return rc;
  std::lock_guard<std::mutex> lk(g_mu);
  EK_HIP_CHECK(hipStreamSynchronize(g_ctx.stream));
  symv_profile_enable(enable > 0 ? enable : 0);
  return 0;
}

int ek_hip_profile_symv_get(double *seconds, long long *launches, double *algorithmic_bytes) {
  int rc = ensure_init(); if (rc) return rc;
  std::lock_guard<std::mutex> lk(g_mu);
  EK_HIP_CHECK(hipStreamSynchronize(g_ctx.stream));
  symv_profile_collect(seconds, launches, algorithmic_bytes);
  return 0;
}

// ---- two-stage tridiagonalisation, piece by piece on host arrays (tests and tools; declared in
// include/ek_hip_debug.h).  Stage 1: A (n x n, lower) -> band (in A) + explicit reflectors V (n x n)
// + tau; *flag = 0, or the reason the CholeskyQR2 panel factorisation gave up.
int ek_hip_debug_sy2sb(int n, double *A, int lda, double *V, int ldv, double *tau, int *flag) {
  if (n < 1) return -1;
  if (!A || lda < n) return -3;
  if (!V || ldv < n) return -5;
  int rc = ensure_init(); if (rc) return rc;
  std::lock_guard<std::mutex> lk(g_mu);
  hipStream_t s = g_ctx.stream;
  const int ld = pad_ld(n);
  const size_t wb = sy2sb_work_bytes(n);
  void *ws;
  rc = workspace(2 * al((size_t)ld * ld * 8) + al(wb) + al((size_t)ld * 8), &ws);
  if (rc) return rc;
  Arena a(ws, g_ctx.ws_bytes);
  double *dA = a.get<double>((size_t)ld * ld), *dV = a.get<double>((size_t)ld * ld);
  char *work = a.get<char>(wb);
  double *dt = a.get<double>(ld);
  EK_HIP_CHECK(hipMemsetAsync(dA, 0, (size_t)ld * ld * 8, s));
  EK_HIP_CHECK(hipMemsetAsync(dV, 0, (size_t)ld * ld * 8, s));
  EK_HIP_CHECK(hipMemsetAsync(dt, 0, (size_t)ld * 8, s));
  EK_HIP_CHECK(hipMemsetAsync(g_ctx.d_info, 0, 4 * sizeof(int), s));
  rc = h2d_matrix(n, n, A, lda, dA, ld, s); if (rc) return rc;
  sy2sb_lower(s, g_ctx.stream2, n, dA, ld, dV, ld, dt, g_ctx.d_info + 2, work);
  EK_HIP_CHECK(hipGetLastError());
  rc = d2h_matrix(n, n, dA, ld, A, lda, s); if (rc) return rc;
  rc = d2h_matrix(n, n, dV, ld, V, ldv, s); if (rc) return rc;
  if (tau) EK_HIP_CHECK(hipMemcpyAsync(tau, dt, (size_t)n * 8, hipMemcpyDeviceToHost, s));
  int f = 0;
  EK_HIP_CHECK(hipMemcpyAsync(&f, g_ctx.d_info + 2, sizeof(int), hipMemcpyDeviceToHost, s));
  EK_HIP_CHECK(hipStreamSynchronize(s));
  if (flag) *flag = f;
  return 0;
}

// Stage 2: the lower band (half bandwidth 64) of A -> d, e; Z (n x ncols, may be null) <- Q2 Z.
__global__ void band_to_matrix_kernel(int n, const double *__restrict__ AB, double *__restrict__ A, int lda) {
  const int c = blockIdx.x;
  for (int d = threadIdx.x; d <= kBandW; d += blockDim.x)
    if (c + d < n) A[(size_t)(c + d) + (size_t)c * lda] = AB[(size_t)d + (size_t)c * kBandLd];
}

// Dense -> band over a team (stage level, for tests): nteam >= 1 rehearses a whole team inside this process (every
// member with its own copy of A -- NaN outside its own strips if EK_HIP_TEAM_POISON=1 --, exchanges by device kernels),
// nteam == 0 makes this process one rank of the attached communicator.  Out: the gathered band in the lower band of A
// (zero elsewhere), the reflectors V and tau of member 0; *mismatch = entries in which the members' bands, V or tau differ.
int ek_hip_debug_sy2sb_team(int n, double *A, int lda, double *V, int ldv, double *tau, int nteam, int *flag,
                            long long *mismatch) {
  if (n < 1 || !A || !V || !tau || lda < n || ldv < n) return -1;
  if (nteam < 0 || nteam > kMaxTeam) return -7;
  int rc = ensure_init(); if (rc) return rc;
  std::lock_guard<std::mutex> lk(g_mu);
  if (nteam == 0 && !g_comm.on) return -7;
  hipStream_t s = g_ctx.stream;
  const int ld = pad_ld(n);
  const int nmem = nteam > 0 ? nteam : 1, P = nteam > 0 ? nteam : g_comm.nranks;
  const size_t wb = sy2sb_dist_work_bytes(n, P), bandb = (size_t)kBandLd * (round_up(n + 1, 128)) * 8;
  const size_t per = 2 * al((size_t)ld * ld * 8) + al(wb) + al(bandb) + al((size_t)ld * 8) + 256;
  void *ws;
  rc = workspace(per * nmem + 512, &ws);
  if (rc) return rc;
  Arena a(ws, g_ctx.ws_bytes);
  unsigned long long *d_cnt = a.get<unsigned long long>(1);
  int *d_flags = a.get<int>(kMaxTeam);
  EK_HIP_CHECK(hipMemsetAsync(d_cnt, 0, 8, s));
  EK_HIP_CHECK(hipMemsetAsync(d_flags, 0, kMaxTeam * sizeof(int), s));
  Sy2sbMember mem[kMaxTeam];
  double *ABs[kMaxTeam];
  for (int m = 0; m < nmem; ++m) {
    double *dA = a.get<double>((size_t)ld * ld), *dV = a.get<double>((size_t)ld * ld);
    char *work = a.get<char>(wb);
    ABs[m] = (double *)a.get<char>(bandb);
    double *dt = a.get<double>(ld);
    EK_HIP_CHECK(hipMemsetAsync(dA, 0, (size_t)ld * ld * 8, s));
    EK_HIP_CHECK(hipMemsetAsync(dV, 0, (size_t)ld * ld * 8, s));
    EK_HIP_CHECK(hipMemsetAsync(dt, 0, (size_t)ld * 8, s));
    rc = h2d_matrix(n, n, A, lda, dA, ld, s); if (rc) return rc;
    mem[m] = Sy2sbMember{dA, ld, dV, ld, dt, d_flags + m, work, nteam > 0 ? m : g_comm.rank};
    const char *poison = getenv("EK_HIP_TEAM_POISON");
    if (poison && poison[0] == '1' && P > 1)
      poison_foreign_strips(s, n, dA, ld, P, mem[m].rank);
  }
  const SytrdExchange x = team_exchange(nteam, 0);
  g_comm.err = 0;
  sy2sb_lower_dist(s, g_ctx.stream2, n, nmem, mem, x);
  for (int m = 0; m < nmem; ++m) pack_band(s, n, mem[m].A, ld, ABs[m]);
  gather_band_strips(s, n, nmem, mem[0].rank, ABs, x);
  EK_HIP_CHECK(hipGetLastError());
  for (int m = 1; m < nmem; ++m) {
    count_mismatch(s, n, n, mem[0].Vall, ld, mem[m].Vall, ld, 0, d_cnt);
    count_mismatch(s, kBandLd, n, ABs[0], kBandLd, ABs[m], kBandLd, 0, d_cnt);
    count_mismatch(s, n, 1, mem[0].tau1, n, mem[m].tau1, n, 0, d_cnt);
  }
  // the band into member 0's matrix (zero elsewhere) and out
  EK_HIP_CHECK(hipMemsetAsync(mem[0].A, 0, (size_t)ld * ld * 8, s));
  hipLaunchKernelGGL(band_to_matrix_kernel, dim3(n), dim3(128), 0, s, n, ABs[0], mem[0].A, ld);
  rc = d2h_matrix(n, n, mem[0].A, ld, A, lda, s); if (rc) return rc;
  rc = d2h_matrix(n, n, mem[0].Vall, ld, V, ldv, s); if (rc) return rc;
  EK_HIP_CHECK(hipMemcpyAsync(tau, mem[0].tau1, (size_t)n * 8, hipMemcpyDeviceToHost, s));
  unsigned long long cnt = 0;
  int hf[kMaxTeam];
  EK_HIP_CHECK(hipMemcpyAsync(&cnt, d_cnt, 8, hipMemcpyDeviceToHost, s));
  EK_HIP_CHECK(hipMemcpyAsync(hf, d_flags, sizeof(hf), hipMemcpyDeviceToHost, s));
  EK_HIP_CHECK(hipStreamSynchronize(s));
  if (mismatch) *mismatch = (long long)cnt;
  int f = 0;
  for (int m = 0; m < nmem; ++m) f |= hf[m];
  if (flag) *flag = f;
  if (g_comm.err) { fprintf(stderr, "[ek_hip] exchange failed: %s\n", comm_error_string()); return -996; }
  return 0;
}

int ek_hip_debug_sb2st(int n, const double *A, int lda, double *d, double *e, double *Z, int ldz, int ncols,
                       int *flag) {
  if (n < 1) return -1;
  if (!A || lda < n) return -3;
  if (!d || (n > 1 && !e)) return -4;
  if (ncols < 0 || (ncols > 0 && (!Z || ldz < n))) return -6;
  int rc = ensure_init(); if (rc) return rc;
  std::lock_guard<std::mutex> lk(g_mu);
  hipStream_t s = g_ctx.stream;
  const int ld = pad_ld(n);
  const size_t wb = sb2st_work_bytes(n);
  void *ws;
  rc = workspace(2 * al((size_t)ld * ld * 8) + al((size_t)ld * (ncols > 0 ? ncols : 1) * 8) + al(wb) +
                 2 * al((size_t)ld * 8), &ws);
  if (rc) return rc;
  Arena a(ws, g_ctx.ws_bytes);
  double *dA = a.get<double>((size_t)ld * ld), *dV2 = a.get<double>((size_t)ld * ld);
  double *dZ = a.get<double>((size_t)ld * (ncols > 0 ? ncols : 1));
  char *work = a.get<char>(wb);
  double *dd = a.get<double>(ld), *de = a.get<double>(ld);
  EK_HIP_CHECK(hipMemsetAsync(dA, 0, (size_t)ld * ld * 8, s));
  EK_HIP_CHECK(hipMemsetAsync(dV2, 0, (size_t)ld * ld * 8, s));
  EK_HIP_CHECK(hipMemsetAsync(dd, 0, 2 * al((size_t)ld * 8), s));
  EK_HIP_CHECK(hipMemsetAsync(g_ctx.d_info, 0, 4 * sizeof(int), s));
  rc = h2d_matrix(n, n, A, lda, dA, ld, s); if (rc) return rc;
  sb2st_lower(s, n, dA, ld, dd, de, dV2, ld, g_ctx.d_info + 2, work);
  if (ncols > 0) {
    rc = h2d_matrix(n, ncols, Z, ldz, dZ, ld, s); if (rc) return rc;
    sb2st_apply_q2(s, n, ncols, dV2, ld, dZ, ld, g_ctx.d_info + 2, work);
    rc = d2h_matrix(n, ncols, dZ, ld, Z, ldz, s); if (rc) return rc;
  }
  EK_HIP_CHECK(hipGetLastError());
  EK_HIP_CHECK(hipMemcpyAsync(d, dd, (size_t)n * 8, hipMemcpyDeviceToHost, s));
  if (n > 1) EK_HIP_CHECK(hipMemcpyAsync(e, de, (size_t)(n - 1) * 8, hipMemcpyDeviceToHost, s));
  int f = 0;
  EK_HIP_CHECK(hipMemcpyAsync(&f, g_ctx.d_info + 2, sizeof(int), hipMemcpyDeviceToHost, s));
  EK_HIP_CHECK(hipStreamSynchronize(s));
  if (flag) *flag = f;
  return 0;
}

int ek_hip_debug_set_two_stage(int min_order) { g_two_stage_min = min_order; return 0; }   // -1: default

// HIP-event brackets around the kernels of the two-stage path bench.py reports a roofline for
// (0 q2_apply_nb_kernel, 1 chase kernel, 2 symm_lower_kernel of every 8th panel); _get after the solves.
int ek_hip_profile_kernels(int enable) {
  int rc = ensure_init(); if (rc) return rc;
  std::lock_guard<std::mutex> lk(g_mu);
  EK_HIP_CHECK(hipStreamSynchronize(g_ctx.stream));
  kprof_enable(enable != 0);
  return 0;
}
int ek_hip_profile_kernels_get(double *seconds, long long *launches) {
  int rc = ensure_init(); if (rc) return rc;
  std::lock_guard<std::mutex> lk(g_mu);
  EK_HIP_CHECK(hipStreamSynchronize(g_ctx.stream));
  kprof_collect(seconds, launches);
  return 0;
}

// counters of the last whole-path solve of this process: out[0] = flops executed by the merge products of the
// divide & conquer (after deflation and column selection), out[1] = 1 if the tridiagonalisation ran in two stages
int ek_hip_debug_last_solve_stats(double *out, int count) {
  std::lock_guard<std::mutex> lk(g_mu);
  for (int i = 0; i < count && i < 8; ++i) out[i] = g_ctx.stats[i];
  return 0;
}

// Timing of the two-stage pieces on a device-generated synthetic matrix of order n:
// seconds[0] dense -> band, [1] band -> tridiagonal, [2] Q2 applied to ncols columns, [3] Q1 applied.
int ek_hip_debug_two_stage_timing(int n, int ncols, int reps, double *seconds, int *flag) {
  if (n < 3 || ncols < 1 || ncols > n) return -1;
  int rc = ensure_init(); if (rc) return rc;
  std::lock_guard<std::mutex> lk(g_mu);
  hipStream_t s = g_ctx.stream;
  const int ld = pad_ld(n);
  const size_t wb1 = sy2sb_work_bytes(n), wb2 = sb2st_work_bytes(n), wb3 = ormtr_work_bytes(n, ncols);
  void *ws;
  rc = workspace(4 * al((size_t)ld * ld * 8) + al(wb1) + al(wb2) + al(wb3) + 3 * al((size_t)ld * 8), &ws);
  if (rc) return rc;
  Arena a(ws, g_ctx.ws_bytes);
  double *dA = a.get<double>((size_t)ld * ld), *dV = a.get<double>((size_t)ld * ld);
  double *dV2 = a.get<double>((size_t)ld * ld), *dZ = a.get<double>((size_t)ld * ld);
  char *w1 = a.get<char>(wb1), *w2 = a.get<char>(wb2), *w3 = a.get<char>(wb3);
  double *dt = a.get<double>(ld), *dd = a.get<double>(ld), *de = a.get<double>(ld);
  hipEvent_t ev[5];
  for (auto &e : ev) EK_HIP_CHECK(hipEventCreate(&e));
  double tot[4] = {0, 0, 0, 0};
  EK_HIP_CHECK(hipMemsetAsync(g_ctx.d_info, 0, 4 * sizeof(int), s));
  for (int r = 0; r < reps; ++r) {
    EK_HIP_CHECK(hipMemsetAsync(dA, 0, (size_t)ld * ld * 8, s));
    EK_HIP_CHECK(hipMemsetAsync(dV, 0, (size_t)ld * ld * 8, s));
    EK_HIP_CHECK(hipMemsetAsync(dV2, 0, (size_t)ld * ld * 8, s));
    EK_HIP_CHECK(hipMemsetAsync(dt, 0, 3 * al((size_t)ld * 8), s));
    synth_matrix(s, n, 1, dA, ld);
    // a DENSE Z, as the eigenvectors of the tridiagonal are: on the identity (mostly zeros for most of the stage) the
    // matrix pipe runs cooler and the Q2 application measures 4 % faster than inside a solve
    synth_matrix(s, n, 7, dZ, ld);
    EK_HIP_CHECK(hipEventRecord(ev[0], s));
    sy2sb_lower(s, g_ctx.stream2, n, dA, ld, dV, ld, dt, g_ctx.d_info + 2, w1);
    EK_HIP_CHECK(hipEventRecord(ev[1], s));
    sb2st_lower(s, n, dA, ld, dd, de, dV2, ld, g_ctx.d_info + 2, w2);
    EK_HIP_CHECK(hipEventRecord(ev[2], s));
    sb2st_apply_q2(s, n, ncols, dV2, ld, dZ, ld, g_ctx.d_info + 2, w2);
    EK_HIP_CHECK(hipEventRecord(ev[3], s));
    ormtr_lower(s, n, ncols, dV, ld, dt, dZ, ld, w3);
    EK_HIP_CHECK(hipEventRecord(ev[4], s));
    EK_HIP_CHECK(hipStreamSynchronize(s));
    for (int q = 0; q < 4; ++q) { float ms = 0.f; EK_HIP_CHECK(hipEventElapsedTime(&ms, ev[q], ev[q + 1])); tot[q] += ms * 1e-3; }
  }
  for (auto &e : ev) (void)hipEventDestroy(e);
  int f = 0;
  EK_HIP_CHECK(hipMemcpy(&f, g_ctx.d_info + 2, sizeof(int), hipMemcpyDeviceToHost));
  if (flag) *flag = f;
  if (seconds) for (int q = 0; q < 4; ++q) seconds[q] = tot[q] / (reps > 0 ? reps : 1);
  return 0;
}

// Tuning hook (not part of the drop-in surface): tridiagonalise a device-generated synthetic
// matrix of order n held with leading dimension ld, `reps` times; seconds[0] = stage time per
// repetition.  Honour EK_SYTRD_MAXCOLS to time only the first panels.
int ek_hip_debug_sytrd(int n, int ld, int reps, double *seconds) {
  int rc = ensure_init(); if (rc) return rc;
  std::lock_guard<std::mutex> lk(g_mu);
  hipStream_t s = g_ctx.stream;
  const int npad = pad_ld(n);
  if (ld < npad) ld = npad;
  const size_t wb = sytrd_work_bytes(n);
  void *ws;
  rc = workspace(al((size_t)ld * npad * 8) + al(wb) + 3 * al((size_t)npad * 8), &ws);
  if (rc) return rc;
  Arena a(ws, g_ctx.ws_bytes);
  double *dA = a.get<double>((size_t)ld * npad);
  char *work = a.get<char>(wb);
  double *dd = a.get<double>(npad), *de = a.get<double>(npad), *dt = a.get<double>(npad);
  hipEvent_t e0, e1;
  EK_HIP_CHECK(hipEventCreate(&e0)); EK_HIP_CHECK(hipEventCreate(&e1));
  double tot = 0.0;
  for (int r = 0; r < reps; ++r) {
    EK_HIP_CHECK(hipMemsetAsync(dA, 0, (size_t)ld * npad * 8, s));
    synth_matrix(s, n, 1, dA, ld);
    EK_HIP_CHECK(hipEventRecord(e0, s));
    sytrd_lower(s, n, dA, ld, dd, de, dt, nullptr, 0, work);
    EK_HIP_CHECK(hipEventRecord(e1, s));
    EK_HIP_CHECK(hipStreamSynchronize(s));
    float ms = 0.f; EK_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    tot += ms * 1e-3;
  }
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  if (seconds) *seconds = tot / (reps > 0 ? reps : 1);
  return 0;
}

// Tuning hook: the distributed tridiagonalisation of a device-generated synthetic matrix.
// nteam >= 1: a whole team rehearsed on this GPU (seconds[0] = time of ALL members' work issued
// back to back, i.e. ~nteam x one rank's compute plus the rehearsal exchange kernels);
// nteam == 0: one rank of the attached communicator.
int ek_hip_debug_sytrd_team(int n, int nteam, int reps, double *seconds) {
  if (n < 1) return -1;
  if (nteam < 0 || nteam > kMaxTeam) return -2;
  int rc = ensure_init(); if (rc) return rc;
  std::lock_guard<std::mutex> lk(g_mu);
  if (nteam == 0 && !g_comm.on) return -995;
  hipStream_t s = g_ctx.stream;
  const int ld = pad_ld(n);
  const int nmem = nteam > 0 ? nteam : 1, P = nteam > 0 ? nteam : g_comm.nranks;
  const size_t wb = sytrd_dist_work_bytes(n, P);
  const size_t per = al((size_t)ld * ld * 8) + al(wb) + 3 * al((size_t)ld * 8);
  void *ws;
  rc = workspace(per * nmem, &ws);
  if (rc) return rc;
  Arena a(ws, g_ctx.ws_bytes);
  SytrdMember mem[kMaxTeam];
  for (int m = 0; m < nmem; ++m) {
    double *dA = a.get<double>((size_t)ld * ld);
    char *work = a.get<char>(wb);
    double *dd = a.get<double>(ld), *de = a.get<double>(ld), *dt = a.get<double>(ld);
    mem[m] = SytrdMember{dA, ld, dd, de, dt, nullptr, 0, work, nteam > 0 ? m : g_comm.rank};
  }
  const SytrdExchange x = team_exchange(nteam, n);
  hipEvent_t e0, e1;
  EK_HIP_CHECK(hipEventCreate(&e0)); EK_HIP_CHECK(hipEventCreate(&e1));
  double tot = 0.0;
  g_comm.err = 0;
  for (int r = 0; r < reps; ++r) {
    for (int m = 0; m < nmem; ++m) {
      EK_HIP_CHECK(hipMemsetAsync(mem[m].A, 0, (size_t)ld * ld * 8, s));
      synth_matrix(s, n, 1, mem[m].A, ld);
    }
    EK_HIP_CHECK(hipEventRecord(e0, s));
    sytrd_lower_dist(s, n, nmem, mem, x);
    EK_HIP_CHECK(hipEventRecord(e1, s));
    EK_HIP_CHECK(hipStreamSynchronize(s));
    float ms = 0.f; EK_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    tot += ms * 1e-3;
  }
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  if (seconds) *seconds = tot / (reps > 0 ? reps : 1);
  return g_comm.err ? -996 : 0;
}

// Tuning hook: the team form of the dense -> band stage on the synthetic matrix; *seconds = the whole team back to back
// on this GPU when nteam >= 1 (divide by nteam for a rank's compute: the wire is not in it)
// lookahead_min: rows from which the team form looks ahead (0: never: every section of a panel back to back; -1: default).
// parts (optional, 4 doubles): [0] whole stage, [1] all panel chains, [2] all "rest of the trailing update" sections,
// [3] first chain + sum over the panels of max(chain p + 1, update p / P), of the LAST repetition, from HIP events around
// them -- with lookahead_min = 0 these are the serial pieces from which tools/team_timing.py models a rank's critical
// path (the chain runs on ONE rank while the others wait for its broadcast; the update is shared)
int ek_hip_debug_sy2sb_team_profile(int n, int nteam, int reps, int lookahead_min, double *seconds, double *parts) {
  if (n < 3) return -1;
  if (nteam < 0 || nteam > kMaxTeam) return -2;
  int rc = ensure_init(); if (rc) return rc;
  std::lock_guard<std::mutex> lk(g_mu);
  if (nteam == 0 && !g_comm.on) return -995;
  hipStream_t s = g_ctx.stream;
  const int ld = pad_ld(n);
  const int nmem = nteam > 0 ? nteam : 1, P = nteam > 0 ? nteam : g_comm.nranks;
  const size_t wb = sy2sb_dist_work_bytes(n, P);
  const size_t per = 2 * al((size_t)ld * ld * 8) + al(wb) + al((size_t)ld * 8) + 256;
  void *ws;
  rc = workspace(per * nmem + 256, &ws);
  if (rc) return rc;
  Arena a(ws, g_ctx.ws_bytes);
  int *d_flags = a.get<int>(kMaxTeam);
  Sy2sbMember mem[kMaxTeam];
  for (int m = 0; m < nmem; ++m) {
    double *dA = a.get<double>((size_t)ld * ld), *dV = a.get<double>((size_t)ld * ld);
    char *work = a.get<char>(wb);
    double *dt = a.get<double>(ld);
    mem[m] = Sy2sbMember{dA, ld, dV, ld, dt, d_flags + m, work, nteam > 0 ? m : g_comm.rank};
  }
  const SytrdExchange x = team_exchange(nteam, 0);
  hipEvent_t e0, e1;
  EK_HIP_CHECK(hipEventCreate(&e0)); EK_HIP_CHECK(hipEventCreate(&e1));
  double tot = 0.0;
  g_comm.err = 0;
  sy2sb_dist_set_lookahead(lookahead_min);
  for (int r = 0; r < reps; ++r) {
    if (parts) sy2sb_dist_profile(true);
    EK_HIP_CHECK(hipMemsetAsync(d_flags, 0, kMaxTeam * sizeof(int), s));
    for (int m = 0; m < nmem; ++m) {
      EK_HIP_CHECK(hipMemsetAsync(mem[m].A, 0, (size_t)ld * ld * 8, s));
      EK_HIP_CHECK(hipMemsetAsync(mem[m].Vall, 0, (size_t)ld * ld * 8, s));
      EK_HIP_CHECK(hipMemsetAsync(mem[m].tau1, 0, (size_t)ld * 8, s));
      synth_matrix(s, n, 1, mem[m].A, ld);
    }
    EK_HIP_CHECK(hipEventRecord(e0, s));
    sy2sb_lower_dist(s, g_ctx.stream2, n, nmem, mem, x);
    EK_HIP_CHECK(hipEventRecord(e1, s));
    EK_HIP_CHECK(hipStreamSynchronize(s));
    float ms = 0.f; EK_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    tot += ms * 1e-3;
    if (parts) { parts[0] = ms * 1e-3; sy2sb_dist_profile_collect(parts + 1, P); sy2sb_dist_profile(false); }
  }
  sy2sb_dist_set_lookahead(-1);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  if (seconds) *seconds = tot / (reps > 0 ? reps : 1);
  return g_comm.err ? -996 : 0;
}

int ek_hip_debug_sy2sb_team_timing(int n, int nteam, int reps, double *seconds) {
  return ek_hip_debug_sy2sb_team_profile(n, nteam, reps, -1, seconds, nullptr);
}

// Tuning hook: Cholesky + reduction to standard form of the synthetic pair, distributed form;
// seconds[0] = potrf, seconds[1] = sygst (whole team back to back when nteam >= 1).
int ek_hip_debug_reduce_team(int n, int nteam, int reps, double *seconds) {
  if (n < 1) return -1;
  if (nteam < 0 || nteam > kMaxTeam) return -2;
  int rc = ensure_init(); if (rc) return rc;
  std::lock_guard<std::mutex> lk(g_mu);
  if (nteam == 0 && !g_comm.on) return -995;
  hipStream_t s = g_ctx.stream;
  const int ld = pad_ld(n), nblk = ceil_div(n, kDiagNB);
  const int nmem = nteam > 0 ? nteam : 1, P = nteam > 0 ? nteam : g_comm.nranks;
  const size_t wb = potrf_dist_work_bytes(n, ld, P), scr = sygst_dist_scratch_doubles(n, ld, P);
  const size_t per = 2 * al((size_t)ld * ld * 8) + al((size_t)nblk * kDiagNB * kDiagNB * 8) + al(wb) +
                     al((size_t)128 * ld * 8) + al(scr * 8) + 256;
  void *ws;
  rc = workspace(per * nmem, &ws);
  if (rc) return rc;
  Arena a(ws, g_ctx.ws_bytes);
  PotrfMember pm[kMaxTeam]; SygstMember sm[kMaxTeam];
  for (int m = 0; m < nmem; ++m) {
    double *dA = a.get<double>((size_t)ld * ld), *dB = a.get<double>((size_t)ld * ld);
    double *dInv = a.get<double>((size_t)nblk * kDiagNB * kDiagNB);
    char *work = a.get<char>(wb);
    double *tw = a.get<double>((size_t)128 * ld), *sc = a.get<double>(scr);
    int *dinfo = a.get<int>(1);
    const int rank = nteam > 0 ? m : g_comm.rank;
    pm[m] = PotrfMember{dB, ld, dInv, dinfo, work, rank};
    sm[m] = SygstMember{dA, ld, dB, ld, dInv, tw, sc, rank};
  }
  const SytrdExchange x = team_exchange(nteam, n);
  hipEvent_t e0, e1, e2;
  EK_HIP_CHECK(hipEventCreate(&e0)); EK_HIP_CHECK(hipEventCreate(&e1)); EK_HIP_CHECK(hipEventCreate(&e2));
  double t1 = 0.0, t2 = 0.0;
  g_comm.err = 0;
  for (int r = 0; r < reps; ++r) {
    for (int m = 0; m < nmem; ++m) {
      EK_HIP_CHECK(hipMemsetAsync(sm[m].A, 0, (size_t)ld * ld * 8, s));
      EK_HIP_CHECK(hipMemsetAsync(pm[m].B, 0, (size_t)ld * ld * 8, s));
      EK_HIP_CHECK(hipMemsetAsync(pm[m].d_info, 0, sizeof(int), s));
      synth_matrix(s, n, 1, sm[m].A, ld);
      synth_matrix(s, n, 2, pm[m].B, ld);
    }
    EK_HIP_CHECK(hipEventRecord(e0, s));
    potrf_lower_dist(s, g_ctx.stream2, n, nmem, pm, x);
    EK_HIP_CHECK(hipEventRecord(e1, s));
    sygst_lower_dist(s, n, nmem, sm, x);
    EK_HIP_CHECK(hipEventRecord(e2, s));
    EK_HIP_CHECK(hipStreamSynchronize(s));
    float ms = 0.f;
    EK_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1)); t1 += ms * 1e-3;
    EK_HIP_CHECK(hipEventElapsedTime(&ms, e1, e2)); t2 += ms * 1e-3;
  }
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipEventDestroy(e2);
  if (seconds) { seconds[0] = t1 / (reps > 0 ? reps : 1); seconds[1] = t2 / (reps > 0 ? reps : 1); }
  return g_comm.err ? -996 : 0;
}

// Team Cholesky: look-ahead on / off (-1: default) and, with profile != 0, HIP events around every owner's chain and every
// rest-of-update section of the NEXT ek_hip_debug_reduce_team rehearsal (run it with the look-ahead OFF: the sections must
// not overlap); _get: parts[0] chains, [1] update sections, [2] first chain + sum over strips of max(next chain, update / P).
int ek_hip_debug_potrf_team_profile(int lookahead, int profile) {
  std::lock_guard<std::mutex> lk(g_mu);
  potrf_dist_set_lookahead(lookahead);
  potrf_dist_profile(profile != 0);
  return 0;
}
int ek_hip_debug_potrf_team_profile_get(int nteam, double *parts) {
  if (!parts) return -1;
  int rc = ensure_init(); if (rc) return rc;
  std::lock_guard<std::mutex> lk(g_mu);
  EK_HIP_CHECK(hipDeviceSynchronize());
  potrf_dist_profile_collect(parts, nteam);
  return 0;
}

// Tuning hook: the first max_cols columns of the tridiagonalisation of the synthetic matrix with the
// matrix, the stage scratch (>= ek_hip_debug_sytrd_work_bytes(n)) and three n-vectors at caller-chosen
// device addresses (placement experiments).  seconds[0] = time of the last of `reps` passes.
unsigned long long ek_hip_debug_sytrd_work_bytes(int n) { return (unsigned long long)sytrd_work_bytes(n); }
int ek_hip_debug_sytrd_at(int n, int max_cols, int reps, double *dA, void *work, double *vecs, double *seconds) {
  if (n < 1 || !dA || !work || !vecs) return -1;
  int rc = ensure_init(); if (rc) return rc;
  std::lock_guard<std::mutex> lk(g_mu);
  hipStream_t s = g_ctx.stream;
  const int ld = pad_ld(n);
  hipEvent_t e0, e1;
  EK_HIP_CHECK(hipEventCreate(&e0)); EK_HIP_CHECK(hipEventCreate(&e1));
  const int old_cols = sytrd_get_max_cols();
  sytrd_set_max_cols(max_cols);
  for (int r = 0; r < reps; ++r) {
    EK_HIP_CHECK(hipMemsetAsync(dA, 0, (size_t)ld * ld * 8, s));
    synth_matrix(s, n, 1, dA, ld);
    EK_HIP_CHECK(hipEventRecord(e0, s));
    sytrd_lower(s, n, dA, ld, vecs, vecs + ld, vecs + 2 * (size_t)ld, nullptr, 0, work);
    EK_HIP_CHECK(hipEventRecord(e1, s));
  }
  sytrd_set_max_cols(old_cols);
  EK_HIP_CHECK(hipStreamSynchronize(s));
  float ms = 0.f; EK_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  if (seconds) *seconds = ms * 1e-3;
  return 0;
}

// Tuning hook: C = alpha op(A) op(B) + beta C on device arrays at caller-chosen addresses, timed.
int ek_hip_debug_gemm_at(int transa, int transb, int m, int n, int k, const double *dA, int lda, const double *dB,
                         int ldb, double beta, double *dC, int ldc, int lower_only, int reps, double *seconds) {
  if (m < 1 || n < 1 || k < 1 || !dA || !dB || !dC) return -1;
  int rc = ensure_init(); if (rc) return rc;
  std::lock_guard<std::mutex> lk(g_mu);
  hipStream_t s = g_ctx.stream;
  hipEvent_t e0, e1;
  EK_HIP_CHECK(hipEventCreate(&e0)); EK_HIP_CHECK(hipEventCreate(&e1));
  gemm(s, transa != 0, transb != 0, m, n, k, -1.0, dA, lda, dB, ldb, beta, dC, ldc, lower_only != 0);   // warm-up
  EK_HIP_CHECK(hipEventRecord(e0, s));
  for (int r = 0; r < reps; ++r) gemm(s, transa != 0, transb != 0, m, n, k, -1.0, dA, lda, dB, ldb, beta, dC, ldc, lower_only != 0);
  EK_HIP_CHECK(hipEventRecord(e1, s));
  EK_HIP_CHECK(hipStreamSynchronize(s));
  float ms = 0.f; EK_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  if (seconds) *seconds = ms * 1e-3 / (reps > 0 ? reps : 1);
  return 0;
}

int ek_hip_debug_sytrd_split(void *alt, int mask) {
  std::lock_guard<std::mutex> lk(g_mu);
  sytrd_debug_split(alt, mask);
  return 0;
}

// Tuning hook: the tridiagonalisation hooks stop after max_cols columns (-1 = all of them).
int ek_hip_debug_set_sytrd_maxcols(int max_cols) {
  std::lock_guard<std::mutex> lk(g_mu);
  sytrd_set_max_cols(max_cols);
  return 0;
}

}  // extern "C"
