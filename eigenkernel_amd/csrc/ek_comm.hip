// ek_comm.hip -- the communicator of the distributed path: what stands where the reference has its
// BLACS context (processes.f90:17-36).  RCCL bound at run time (one rank per GPU over xGMI), the same two exchanges
// through the host's allgatherv hook, and the team's agreements; see include/ek_hip.h.
#include "ek_api_internal.h"
#include <dlfcn.h>

namespace ek {
namespace api {

// ---- RCCL, bound at run time.  The collective of the distributed tridiagonalisation (one
// all-reduce per Householder column) has to be issued from inside the library on the library's
// stream: a host-language collective per column costs more than the column.  dlopen keeps
// libek_hip.so loadable where RCCL is absent and makes it share the copy the host already loaded
// (PyTorch ships its own librccl.so.1).  Replaces the BLACS calls inside PDSYTRD.
struct Rccl {
  void *h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;   // (optional)
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;              // (grids with nprow > 1)
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  int load() {
    if (h) return 0;
    const char *names[] = {getenv("EK_HIP_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *nm : names) {
      if (!nm || !*nm) continue;
      h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
      if (h) break;
    }
    if (!h) { fprintf(stderr, "[ek_hip] cannot load RCCL: %s\n", dlerror()); return -997; }
    GetUniqueId = (decltype(GetUniqueId))dlsym(h, "ncclGetUniqueId");
    CommInitRank = (decltype(CommInitRank))dlsym(h, "ncclCommInitRank");
    CommDestroy = (decltype(CommDestroy))dlsym(h, "ncclCommDestroy");
    AllReduce = (decltype(AllReduce))dlsym(h, "ncclAllReduce");
    GetErrorString = (decltype(GetErrorString))dlsym(h, "ncclGetErrorString");
    Broadcast = (decltype(Broadcast))dlsym(h, "ncclBroadcast");
    GroupStart = (decltype(GroupStart))dlsym(h, "ncclGroupStart");
    GroupEnd = (decltype(GroupEnd))dlsym(h, "ncclGroupEnd");
    AllGather = (decltype(AllGather))dlsym(h, "ncclAllGather");
    Send = (decltype(Send))dlsym(h, "ncclSend");
    Recv = (decltype(Recv))dlsym(h, "ncclRecv");
    if (!GetUniqueId || !CommInitRank || !CommDestroy || !AllReduce || !GetErrorString || !Broadcast ||
        !GroupStart || !GroupEnd) {
      fprintf(stderr, "[ek_hip] RCCL symbols missing\n");
      dlclose(h); h = nullptr; return -997;
    }
    return 0;
  }
};
Rccl g_rccl;
Comm g_comm;

void rccl_allreduce(hipStream_t s, int nmem, double *const *bufs, size_t count, void *) {
  if (nmem != 1 || !g_comm.on) { if (!g_comm.err) g_comm.err = (int)ncclInvalidUsage; return; }
  const ncclResult_t r = g_rccl.AllReduce(bufs[0], bufs[0], count, ncclDouble, ncclSum, g_comm.comm, s);
  if (r != ncclSuccess && !g_comm.err) g_comm.err = (int)r;
}

// all-gather of unequal pieces, in place: one grouped ncclBroadcast per owner
void rccl_allgatherv(hipStream_t s, int nmem, int, double *const *bufs, const size_t *offs,
                     const size_t *counts, int nranks, void *) {
  if (nmem != 1 || !g_comm.on || nranks != g_comm.nranks) { if (!g_comm.err) g_comm.err = (int)ncclInvalidUsage; return; }
  // equal pieces that lie side by side in rank order (a round of P strips of the D&C's compact bases, a round of band
  // strips): that is ncclAllGather's own in-place layout -- one ring collective instead of P broadcasts
  bool side_by_side = g_rccl.AllGather != nullptr && counts[0] > 0;
  for (int q = 1; q < nranks && side_by_side; ++q) side_by_side = counts[q] == counts[0] && offs[q] == offs[0] + (size_t)q * counts[0];
  if (side_by_side) {
    const ncclResult_t ra = g_rccl.AllGather(bufs[0] + offs[g_comm.rank], bufs[0] + offs[0], counts[0], ncclDouble, g_comm.comm, s);
    if (ra != ncclSuccess && !g_comm.err) g_comm.err = (int)ra;
    return;
  }
  ncclResult_t r = g_rccl.GroupStart();
  for (int root = 0; root < nranks && r == ncclSuccess; ++root)
    if (counts[root] > 0)
      r = g_rccl.Broadcast(bufs[0] + offs[root], bufs[0] + offs[root], counts[root], ncclDouble, root, g_comm.comm, s);
  const ncclResult_t r2 = g_rccl.GroupEnd();
  if (r == ncclSuccess) r = r2;
  if (r != ncclSuccess && !g_comm.err) g_comm.err = (int)r;
}


// Orders from which the whole-path call tridiagonalises in two stages (dense -> band on the matrix
// cores, band -> tridiagonal by bulge chasing; ek_sy2sb.hip, ek_sb2st.hip) instead of the one-stage
// Householder reduction.  EK_HIP_TWO_STAGE_MIN overrides (0 = never); ek_hip_debug_set_two_stage too.
// Measured with tools/crossover.py (standard problem, full spectrum, one-stage / two-stage seconds):
// 512: 0.0100 / 0.0074, 1024: 0.0190 / 0.0142, 2048: 0.0388 / 0.0290, 4096: 0.0914 / 0.066,
// 8192: 0.3065 / 0.19 -- the two-stage form is ahead by a quarter and more from 512 on (at the start of
// round 2 the difference below 2048 was a millisecond or two and the crossover stood at 2048).  Below 512 the
// whole-path call keeps the one-stage form, whose by-products (PDSYTRD's reflectors in A) are what a caller
// of the reference finds there; INTEGRATION.md says what A holds after a two-stage solve.
int g_two_stage_min = -1;
int two_stage_min() {
  if (g_two_stage_min >= 0) return g_two_stage_min;
  static int env = -2;
  if (env == -2) { const char *e = getenv("EK_HIP_TWO_STAGE_MIN"); env = e ? atoi(e) : -1; }
  return env >= 0 ? env : 512;
}

// From how many ranks on the Cholesky factor and the reduction to standard form are distributed
// as well (below that their replicated forms are cheaper); EK_HIP_DIST_MIN_RANKS overrides (tests).
int dist_min_ranks() {
  const char *e = getenv("EK_HIP_DIST_MIN_RANKS");
  return e ? atoi(e) : 3;
}

// ---- the same two exchanges through the host's allgatherv hook (ek_hip_set_allgatherv): for
// hosts that have MPI but no RCCL-capable node, and for multi-process tests on one GPU.  Every
// exchange drains the stream and crosses PCIe twice, so this is a compatibility path, not a fast
// one.  The sum is formed on the host in rank order: bit-identical on every rank.
std::vector<double> g_hx_send, g_hx_recv;
void host_allreduce(hipStream_t s, int nmem, double *const *bufs, size_t count, void *) {
  if (nmem != 1 || !g_comm.on || !g_allgatherv) { if (!g_comm.err) g_comm.err = (int)ncclInvalidUsage; return; }
  const int P = g_comm.nranks;
  g_hx_send.resize(count); g_hx_recv.resize(count * P);
  std::vector<long long> counts(P, (long long)count), displs(P);
  for (int r = 0; r < P; ++r) displs[r] = (long long)r * (long long)count;
  bool ok = hipStreamSynchronize(s) == hipSuccess &&
            hipMemcpy(g_hx_send.data(), bufs[0], count * 8, hipMemcpyDeviceToHost) == hipSuccess;
  if (ok) ok = g_allgatherv(g_hx_send.data(), (long long)count, g_hx_recv.data(), counts.data(), displs.data(),
                            g_allgatherv_user) == 0;
  if (ok) {
    for (size_t i = 0; i < count; ++i) {
      double v = 0.0;
      for (int r = 0; r < P; ++r) v += g_hx_recv[(size_t)r * count + i];
      g_hx_send[i] = v;
    }
    ok = hipMemcpy(bufs[0], g_hx_send.data(), count * 8, hipMemcpyHostToDevice) == hipSuccess;
  }
  if (!ok && !g_comm.err) g_comm.err = (int)ncclSystemError;
}
void host_allgatherv(hipStream_t s, int nmem, int, double *const *bufs, const size_t *offs, const size_t *counts,
                     int nranks, void *) {
  if (nmem != 1 || !g_comm.on || !g_allgatherv || nranks != g_comm.nranks) {
    if (!g_comm.err) g_comm.err = (int)ncclInvalidUsage;
    return;
  }
  const int me = g_comm.rank;
  std::vector<long long> cnt(nranks), displs(nranks);
  long long tot = 0;
  for (int r = 0; r < nranks; ++r) { cnt[r] = (long long)counts[r]; displs[r] = tot; tot += cnt[r]; }
  g_hx_send.resize(counts[me] > 0 ? counts[me] : 1); g_hx_recv.resize(tot > 0 ? (size_t)tot : 1);
  bool ok = hipStreamSynchronize(s) == hipSuccess;
  if (ok && counts[me] > 0)
    ok = hipMemcpy(g_hx_send.data(), bufs[0] + offs[me], counts[me] * 8, hipMemcpyDeviceToHost) == hipSuccess;
  if (ok) ok = g_allgatherv(g_hx_send.data(), cnt[me], g_hx_recv.data(), cnt.data(), displs.data(), g_allgatherv_user) == 0;
  for (int r = 0; ok && r < nranks; ++r)
    if (r != me && counts[r] > 0)
      ok = hipMemcpy(bufs[0] + offs[r], g_hx_recv.data() + displs[r], counts[r] * 8, hipMemcpyHostToDevice) == hipSuccess;
  if (!ok && !g_comm.err) g_comm.err = (int)ncclSystemError;
}

// Pairwise exchange (the eigenvector columns' way from the rank that formed them to the cells of its process column, on
// grids with more than one process row: ek_solve.hip).  Where the reference's PDORMTR / PDTRTRS leave Z on the 2-D grid
// by construction (solver_scalapack_all.f90:115, generalized_to_standard.f90:103).
void team_sendrecv(hipStream_t s, int npeers, const int *peers, double *const *send, const size_t *send_counts,
                   double *const *recv, const size_t *recv_counts) {
  if (!g_comm.on) { if (!g_comm.err) g_comm.err = (int)ncclInvalidUsage; return; }
  if (!g_comm.host) {
    if (npeers == 0) return;
    if (!g_rccl.Send || !g_rccl.Recv) { if (!g_comm.err) g_comm.err = (int)ncclInvalidUsage; return; }
    ncclResult_t r = g_rccl.GroupStart();
    for (int i = 0; i < npeers && r == ncclSuccess; ++i) {
      if (send_counts[i] > 0) r = g_rccl.Send(send[i], send_counts[i], ncclDouble, peers[i], g_comm.comm, s);
      if (r == ncclSuccess && recv_counts[i] > 0) r = g_rccl.Recv(recv[i], recv_counts[i], ncclDouble, peers[i], g_comm.comm, s);
    }
    const ncclResult_t r2 = g_rccl.GroupEnd();
    if (r == ncclSuccess) r = r2;
    if (r != ncclSuccess && !g_comm.err) g_comm.err = (int)r;
    return;
  }
  // host communicator: a header round (who sends how much to whom), then one all-gather of everybody's payloads, of which
  // a rank keeps what is addressed to it -- a compatibility path like the other host exchanges
  if (!g_allgatherv) { if (!g_comm.err) g_comm.err = (int)ncclInvalidUsage; return; }
  const int P = g_comm.nranks, me = g_comm.rank, H = 2 * kMaxTeam + 1;
  std::vector<double> hdr(H, 0.0), hall((size_t)H * P);
  hdr[0] = npeers;
  size_t mine = 0;
  for (int i = 0; i < npeers && i < kMaxTeam; ++i) { hdr[1 + 2 * i] = peers[i]; hdr[2 + 2 * i] = (double)send_counts[i]; mine += send_counts[i]; }
  std::vector<long long> cnt(P, H), dsp(P);
  for (int r = 0; r < P; ++r) dsp[r] = (long long)r * H;
  bool ok = npeers <= kMaxTeam && hipStreamSynchronize(s) == hipSuccess;
  ok = g_allgatherv(hdr.data(), H, hall.data(), cnt.data(), dsp.data(), g_allgatherv_user) == 0 && ok;
  long long tot = 0;
  for (int r = 0; r < P; ++r) {
    long long c = 0;
    const int np_r = (int)hall[(size_t)r * H];
    for (int i = 0; i < np_r && i < kMaxTeam; ++i) c += (long long)hall[(size_t)r * H + 2 + 2 * i];
    cnt[r] = c; dsp[r] = tot; tot += c;
  }
  g_hx_send.resize(mine > 0 ? mine : 1); g_hx_recv.resize(tot > 0 ? (size_t)tot : 1);
  size_t o = 0;
  for (int i = 0; ok && i < npeers; ++i) {
    if (send_counts[i] > 0) ok = hipMemcpy(g_hx_send.data() + o, send[i], send_counts[i] * 8, hipMemcpyDeviceToHost) == hipSuccess;
    o += send_counts[i];
  }
  // (every rank makes the second call even after a local failure: the exchange is collective)
  ok = g_allgatherv(g_hx_send.data(), (long long)mine, g_hx_recv.data(), cnt.data(), dsp.data(), g_allgatherv_user) == 0 && ok;
  for (int i = 0; ok && i < npeers; ++i) {
    const int r = peers[i];
    const int np_r = (int)hall[(size_t)r * H];
    long long off = dsp[r];
    bool found = false;
    for (int j = 0; j < np_r && j < kMaxTeam; ++j) {
      const long long c = (long long)hall[(size_t)r * H + 2 + 2 * j];
      if ((int)hall[(size_t)r * H + 1 + 2 * j] == me) {
        found = (size_t)c == recv_counts[i];
        if (found && c > 0) ok = hipMemcpy(recv[i], g_hx_recv.data() + off, (size_t)c * 8, hipMemcpyHostToDevice) == hipSuccess;
        break;
      }
      off += c;
    }
    ok = ok && (found || recv_counts[i] == 0);
  }
  if (!ok && !g_comm.err) g_comm.err = (int)ncclSystemError;
}

// The exchange a team of this process uses: the device-side emulation of a rehearsed team (nteam members in this
// process), the host's allgatherv hook, or RCCL.  (n is kept for the callers' sake: it sized the peer windows of the
// one-stage form, removed in round 4.)
SytrdExchange team_exchange(int nteam, int) {
  SytrdExchange x{nteam > 0 ? nteam : g_comm.nranks, nullptr, nullptr};
  if (nteam > 0) { x.allreduce = sytrd_team_allreduce; x.allgatherv = team_allgatherv; }
  else if (g_comm.host) { x.allreduce = host_allreduce; x.allgatherv = host_allgatherv; }
  else { x.allreduce = rccl_allreduce; x.allgatherv = rccl_allgatherv; }
  return x;
}

// One small all-reduce of a status word over the attached communicator.  A sticky exchange error recorded earlier in
// the call (g_comm.err: the stages' exchanges set it and go on) is neither erased nor kept private: it travels in the
// word's high part, so every rank learns that SOME rank's exchange failed, and the local record survives the vote.
constexpr double kVoteErr = 1048576.0;       // (> kMaxTeam: the low part counts the ranks that raised `local`)
static bool comm_vote(int local, double *sum) {
  const SytrdExchange x = team_exchange(0);
  const int prev = g_comm.err;
  double st = (local ? 1.0 : 0.0) + (prev ? kVoteErr : 0.0);
  bool ok = hipMemcpy(g_ctx.d_status, &st, sizeof(double), hipMemcpyHostToDevice) == hipSuccess;
  double *bufs[1] = {g_ctx.d_status};
  g_comm.err = 0;
  x.allreduce(g_ctx.stream, 1, bufs, 1, x.user);
  ok = ok && hipStreamSynchronize(g_ctx.stream) == hipSuccess && !g_comm.err;
  ok = ok && hipMemcpy(&st, g_ctx.d_status, sizeof(double), hipMemcpyDeviceToHost) == hipSuccess;
  if (prev) g_comm.err = prev;               // (else: whatever the vote's own exchange recorded stays)
  *sum = st;
  return ok;
}

// 1 if `local` is non-zero on ANY rank of the team (the same answer on all of them), else 0; -996 on every rank if the
// vote itself or an earlier exchange of the call failed on any of them
int comm_any(int local) {
  if (!g_comm.on || g_comm.nranks <= 1) return g_comm.err ? -996 : (local ? 1 : 0);
  double st = 0.0;
  if (!comm_vote(local, &st) || st >= kVoteErr) return -996;
  return st != 0.0 ? 1 : 0;
}

// A rank-local failure (allocation, staging copy) in front of a collective part of a call must not leave
// the other ranks waiting in that collective: every rank contributes its status to one small all-reduce
// over the attached communicator and all of them leave together -- the failing rank with its own code,
// the others with -993.  Returns 0 when every rank is fine.  (The word lives in memory allocated at
// initialisation, so the agreement itself needs nothing that could fail locally.)
int comm_agree(int local_rc) {
  if (!g_comm.on || g_comm.nranks <= 1) return local_rc;
  double st = 0.0;
  if (!comm_vote(local_rc, &st)) return local_rc ? local_rc : -996;
  if (st >= kVoteErr) return local_rc ? local_rc : -996;
  if (st != 0.0) return local_rc ? local_rc : -993;
  return 0;
}

const char *comm_error_string() {
  if (g_comm.host || !g_rccl.GetErrorString) return "exchange through the host hook failed";
  return g_rccl.GetErrorString((ncclResult_t)g_comm.err);
}


void comm_teardown() {
  if (g_comm.on && !g_comm.host && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(g_comm.comm);
  g_comm = Comm{};
}

}  // namespace api
}  // namespace ek

using namespace ek;
using namespace ek::api;

extern "C" {

// ---- communicator of the distributed path: one rank per GPU, RCCL over xGMI.  The host
// obtains the 128-byte id on rank 0, broadcasts it with whatever it has (MPI_Bcast in the
// Fortran host, torch.distributed in the tests) and every rank calls ek_hip_comm_init.
int ek_hip_comm_unique_id(void *id, int bytes) {
  if (!id) return -1;
  if (bytes < (int)sizeof(ncclUniqueId)) return -2;
  int rc = ensure_init(); if (rc) return rc;
  std::lock_guard<std::mutex> lk(g_mu);
  rc = g_rccl.load(); if (rc) return rc;
  ncclUniqueId uid;
  const ncclResult_t r = g_rccl.GetUniqueId(&uid);
  if (r != ncclSuccess) { fprintf(stderr, "[ek_hip] ncclGetUniqueId: %s\n", g_rccl.GetErrorString(r)); return -996; }
  memcpy(id, &uid, sizeof(uid));
  return 0;
}

int ek_hip_comm_init(const void *id, int bytes, int nranks, int rank) {
  if (!id) return -1;
  if (bytes < (int)sizeof(ncclUniqueId)) return -2;
  if (nranks < 1 || nranks > kMaxTeam) return -3;
  if (rank < 0 || rank >= nranks) return -4;
  int rc = ensure_init(); if (rc) return rc;
  std::lock_guard<std::mutex> lk(g_mu);
  rc = g_rccl.load(); if (rc) return rc;
  if (g_comm.on && !g_comm.host) (void)g_rccl.CommDestroy(g_comm.comm);
  g_comm = Comm{};
  EK_HIP_CHECK(hipSetDevice(g_ctx.device));
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof(uid));
  const ncclResult_t r = g_rccl.CommInitRank(&g_comm.comm, nranks, uid, rank);
  if (r != ncclSuccess) { fprintf(stderr, "[ek_hip] ncclCommInitRank: %s\n", g_rccl.GetErrorString(r)); return -996; }
  g_comm.on = true; g_comm.nranks = nranks; g_comm.rank = rank; g_comm.err = 0;
  return 0;
}

// The same distributed stages with every exchange routed through the host's allgatherv hook
// (ek_hip_set_allgatherv) instead of RCCL.
int ek_hip_comm_attach_host(int nranks, int rank) {
  if (nranks < 1 || nranks > kMaxTeam) return -1;
  if (rank < 0 || rank >= nranks) return -2;
  int rc = ensure_init(); if (rc) return rc;
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_allgatherv) return -998;
  if (g_comm.on && !g_comm.host) (void)g_rccl.CommDestroy(g_comm.comm);
  g_comm = Comm{};
  g_comm.on = true; g_comm.host = true; g_comm.nranks = nranks; g_comm.rank = rank;
  return 0;
}

int ek_hip_comm_size(void) { return g_comm.on ? g_comm.nranks : 0; }
int ek_hip_comm_rank(void) { return g_comm.on ? g_comm.rank : -1; }

int ek_hip_comm_destroy(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (g_comm.on) {
    if (g_ctx.ready) (void)hipStreamSynchronize(g_ctx.stream);
    if (!g_comm.host) (void)g_rccl.CommDestroy(g_comm.comm);
  }
  g_comm = Comm{};
  return 0;
}

// sum over the ranks of the attached communicator of a device vector, in place (binding check;
// the same call the tridiagonalisation issues once per column)
int ek_hip_comm_allreduce_device(double *dbuf, long long count) {
  if (count < 0) return -2;
  if (count > 0 && !dbuf) return -1;
  int rc = ensure_init(); if (rc) return rc;
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_comm.on) return -995;
  g_comm.err = 0;
  double *bufs[1] = {dbuf};
  if (count > 0) team_exchange(0).allreduce(g_ctx.stream, 1, bufs, (size_t)count, nullptr);
  EK_HIP_CHECK(hipStreamSynchronize(g_ctx.stream));
  return g_comm.err ? -996 : 0;
}


}  // extern "C"
