// ek_api.hip -- the C-ABI of libek_hip.so (declared in include/ek_hip.h).
//
// Host side of the drop-in boundary: validates arguments the way a LAPACK-style routine
// does (info = -k), stages host arrays into padded device work arrays, runs the stage
// kernels on one HIP stream and hands results back.  No numerical work happens on the CPU.
#include "../../include/ek_hip.h"
#include "../../include/ek_hip_debug.h"
#include "ek_common.h"

#include <rccl/rccl.h>   // types only: the library is bound at run time (dlopen), see Rccl below
#include <dlfcn.h>

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

namespace {

using namespace ek;

struct Context {
  bool ready = false;
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t stream2 = nullptr;   // look-ahead work (panel chain of the Cholesky factorisation)
  // cached device workspace (grown on demand, never shrunk until finalize)
  void *ws = nullptr;         // what the stages use (may sit inside a larger allocation, see place_workspace)
  void *ws_alloc = nullptr;   // what hipFree gets
  size_t ws_bytes = 0;
  int *d_info = nullptr;
  double *d_status = nullptr;   // one word for the team's status agreements (comm_agree)
  double *d_stats = nullptr;    // [8] counters of the last whole-path solve ([0] flops the D&C merge products executed)
  double stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};
};
Context g_ctx;
std::mutex g_mu;

int ensure_init() {
  if (g_ctx.ready) return 0;
  int ndev = 0;
  EK_HIP_CHECK(hipGetDeviceCount(&ndev));
  if (ndev <= 0) {
    fprintf(stderr, "[ek_hip] no HIP device visible: this library has no CPU fallback\n");
    return -1000 - (int)hipErrorNoDevice;
  }
  EK_HIP_CHECK(hipSetDevice(g_ctx.device));
  EK_HIP_CHECK(hipStreamCreateWithFlags(&g_ctx.stream, hipStreamNonBlocking));
  {
    // the look-ahead stream carries short latency-bound chains beside a chip-filling GEMM: its
    // workgroups must be dispatched first
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    EK_HIP_CHECK(hipStreamCreateWithPriority(&g_ctx.stream2, hipStreamNonBlocking, hi));
  }
  EK_HIP_CHECK(hipMalloc((void **)&g_ctx.d_info, 64 * sizeof(int)));
  EK_HIP_CHECK(hipMalloc((void **)&g_ctx.d_status, 64));
  EK_HIP_CHECK(hipMalloc((void **)&g_ctx.d_stats, 64));
  EK_HIP_CHECK(hipMemset(g_ctx.d_stats, 0, 64));
  g_ctx.ready = true;
  return 0;
}

int workspace(size_t bytes, void **p) {
  if (bytes > g_ctx.ws_bytes) {
    if (g_ctx.ws_alloc) EK_HIP_CHECK(hipFree(g_ctx.ws_alloc));
    g_ctx.ws = nullptr; g_ctx.ws_alloc = nullptr; g_ctx.ws_bytes = 0;
    EK_HIP_CHECK(hipMalloc(&g_ctx.ws_alloc, bytes));
    g_ctx.ws = g_ctx.ws_alloc;
    g_ctx.ws_bytes = bytes;
  }
  *p = g_ctx.ws;
  return 0;
}

// simple bump allocator over the cached workspace, 256-byte aligned pieces
struct Arena {
  char *base; size_t off = 0, cap;
  Arena(void *p, size_t c) : base((char *)p), cap(c) {}
  template <typename T> T *get(size_t count) {
    size_t bytes = (count * sizeof(T) + 255) & ~(size_t)255;
    T *r = (T *)(base + off);
    off += bytes;
    return r;
  }
};
inline size_t al(size_t bytes) { return (bytes + 255) & ~(size_t)255; }

// Placement of the tridiagonalisation's scratch.  The HBM-bound symv runs 2.6 % faster (4.5 % on the
// largest trailing matrices) or slower depending on where its SCRATCH (x, the panel, the partial
// sums: 64 MB at N = 16384) lies relative to the matrix it streams: device memory comes in two
// "colours" of some physical origin (tools/placement_addr.py: every allocation has one, large blocks
// change it at multi-GiB boundaries), and the launch is fast when matrix and scratch have different
// colours -- whichever they are -- and slow when they share one.  Nothing about an address tells its
// colour, so it is measured: when the workspace has been (re)allocated, the first panel of a
// tridiagonalisation of a synthetic matrix is timed in the matrix's place with the scratch (a) where
// the arena has it and (b) in up to three small separate allocations, until one is clearly faster
// than another; that one is kept for all later solves.  ~30 ms per candidate, once per workspace
// size (it happens in the warm-up solve); EK_HIP_PLACEMENT=0 turns it off.
struct ScratchChoice {
  void *buf = nullptr;            // separately allocated scratch in use (nullptr: the arena's own)
  size_t bytes = 0;
  const void *for_ws = nullptr;   // the workspace allocation this choice was made for
  int for_n = 0;
};
ScratchChoice g_scratch;

void release_scratch_choice() {
  if (g_scratch.buf) (void)hipFree(g_scratch.buf);
  g_scratch = ScratchChoice{};
}

// returns the scratch to use for the tridiagonalisation of order n on matrix wA (arena_work if nothing better)
void *choose_sytrd_scratch(int n, int ld, double *wA, void *arena_work, double *vecs, size_t need) {
  static int enabled = -1;
  if (enabled < 0) { const char *e = getenv("EK_HIP_PLACEMENT"); enabled = e ? atoi(e) : 1; }
  if (!enabled || n < 8192) return arena_work;
  if (g_scratch.for_ws == g_ctx.ws_alloc && g_scratch.for_n == n && g_scratch.bytes >= need)
    return g_scratch.buf ? g_scratch.buf : arena_work;
  release_scratch_choice();
  hipStream_t s = g_ctx.stream;
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { (void)hipGetLastError(); return arena_work; }
  const int old_cols = sytrd_get_max_cols();
  auto probe = [&](void *work) -> float {
    sytrd_set_max_cols(64);
    for (int rep = 0; rep < 2; ++rep) {                                // first pass warms up, second is timed
      (void)hipMemsetAsync(wA, 0, (size_t)ld * ld * 8, s);
      synth_matrix(s, n, 1, wA, ld);
      (void)hipEventRecord(e0, s);
      sytrd_lower(s, n, wA, ld, vecs, vecs + ld, vecs + 2 * (size_t)ld, nullptr, 0, work);
      (void)hipEventRecord(e1, s);
    }
    sytrd_set_max_cols(old_cols);
    float ms = 1e30f;
    if (hipStreamSynchronize(s) == hipSuccess) (void)hipEventElapsedTime(&ms, e0, e1);
    else (void)hipGetLastError();
    return ms;
  };
  constexpr int kCand = 4;
  void *cand[kCand] = {arena_work, nullptr, nullptr, nullptr};
  float t[kCand];
  int ncand = 0, best = 0;
  float tmin = 1e30f, tmax = 0.f;
  char msg[256]; int mlen = 0; msg[0] = 0;
  for (int c = 0; c < kCand; ++c) {
    if (c > 0 && hipMalloc(&cand[c], need) != hipSuccess) { (void)hipGetLastError(); cand[c] = nullptr; break; }
    t[c] = probe(cand[c]);
    ++ncand;
    if (mlen < 240) mlen += snprintf(msg + mlen, sizeof(msg) - mlen, " %.3f", t[c]);
    if (t[c] < tmin) { tmin = t[c]; best = c; }
    if (t[c] > tmax) tmax = t[c];
    if (tmin < 0.98f * tmax) break;                     // both colours seen: the faster one is known
  }
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  for (int c = 1; c < ncand; ++c) if (c != best && cand[c]) (void)hipFree(cand[c]);
  g_scratch.buf = best > 0 ? cand[best] : nullptr;
  g_scratch.bytes = need; g_scratch.for_ws = g_ctx.ws_alloc; g_scratch.for_n = n;
  if (getenv("EK_HIP_PLACEMENT_VERBOSE"))
    fprintf(stderr, "[ek_hip] scratch placement: first-panel probes (ms; first = inside the arena):%s -> %s\n", msg,
            best ? "a separate allocation" : "the arena's own");
  return g_scratch.buf ? g_scratch.buf : arena_work;
}

// device buffers of one host-array call: released on every exit path
struct DevMem {
  std::vector<void *> ptrs;
  ~DevMem() { for (void *p : ptrs) (void)hipFree(p); }
  int alloc(double **p, size_t bytes) {
    EK_HIP_CHECK(hipMalloc((void **)p, bytes > 0 ? bytes : 8));
    ptrs.push_back(*p);
    return 0;
  }
};

// descriptor checks for the 1x1 grid this round implements; returns 0 or the LAPACK-style
// 100*argpos + field code ScaLAPACK uses (-(argpos*100 + field)).
int check_desc(const int *desc, int argpos, int m, int n, int lld_rows = -1) {
  if (lld_rows < 0) lld_rows = m;
  if (!desc) return -argpos;
  if (desc[0] != 1) return -(argpos * 100 + 1);
  if (desc[2] != m) return -(argpos * 100 + 3);
  if (desc[3] != n) return -(argpos * 100 + 4);
  if (desc[4] < 1 || desc[4] != desc[5]) return -(argpos * 100 + 5);
  if (desc[6] != 0) return -(argpos * 100 + 7);
  if (desc[7] != 0) return -(argpos * 100 + 8);
  if (desc[8] < (lld_rows > 1 ? lld_rows : 1)) return -(argpos * 100 + 9);
  return 0;
}

// NUMROC with source process 0: rows/columns of an n-long dimension (blocks nb) owned by `me` of `np`
int numroc0(int n, int nb, int me, int np) {
  const int nblocks = n / nb;
  int num = (nblocks / np) * nb;
  const int extra = nblocks % np;
  if (me < extra) num += nb;
  else if (me == extra) num += n % nb;
  return num;
}

// Owner cell of a process grid for the replicated-input mode (ek_hip_solve_replicated)
struct GridCell { int nb, nprow, npcol, myrow, mycol; };

// Exchange hook for block-cyclically distributed inputs (ek_hip_set_allgatherv)
ek_hip_allgatherv_fn g_allgatherv = nullptr;
void *g_allgatherv_user = nullptr;

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
    if (!GetUniqueId || !CommInitRank || !CommDestroy || !AllReduce || !GetErrorString || !Broadcast ||
        !GroupStart || !GroupEnd) {
      fprintf(stderr, "[ek_hip] RCCL symbols missing\n");
      dlclose(h); h = nullptr; return -997;
    }
    return 0;
  }
};
Rccl g_rccl;
struct Comm {
  bool on = false;
  bool host = false;    // exchanges go through the host's allgatherv hook instead of RCCL
  ncclComm_t comm = nullptr;
  int nranks = 0, rank = 0;
  int err = 0;          // first failing collective since the last check (ncclResult_t)
};
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
  ncclResult_t r = g_rccl.GroupStart();
  for (int root = 0; root < nranks && r == ncclSuccess; ++root)
    if (counts[root] > 0)
      r = g_rccl.Broadcast(bufs[0] + offs[root], bufs[0] + offs[root], counts[root], ncclDouble, root, g_comm.comm, s);
  const ncclResult_t r2 = g_rccl.GroupEnd();
  if (r == ncclSuccess) r = r2;
  if (r != ncclSuccess && !g_comm.err) g_comm.err = (int)r;
}

constexpr int kPotrfRlMin = 1024;

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

// ---- peer windows (ek_hip_comm_peer_enable): the per-column exchange of the tridiagonalisation
// without a collective.  Each rank allocates one receive area in its HBM, exports it with
// hipIpcGetMemHandle, and maps everybody else's; contributions are stored straight into the peers'
// areas by yreduce and announced by stream memory operations.
struct PeerX {
  bool on = false;
  PeerWindow win{};
  unsigned long long seq = 0;
  size_t bytes = 0;
  bool opened[kMaxTeam] = {};
};
PeerX g_peer;
constexpr size_t kPeerFlagBytes = 256;    // kMaxTeam 64-bit flags, padded

void peer_signal(hipStream_t s, unsigned long long seq, void *) {
  const PeerWindow &w = g_peer.win;
  if (w.nranks <= 1) return;
  const hipError_t e = hipStreamWaitValue64(s, w.base[w.me], seq * (unsigned long long)(w.nranks - 1),
                                            hipStreamWaitValueGte, 0xFFFFFFFFFFFFFFFFull);
  if (e != hipSuccess && !g_comm.err) g_comm.err = (int)ncclSystemError;
}

// Releases whatever has been allocated, opened or mapped so far: also called on the failure exits of
// ek_hip_comm_peer_enable, where the windows are not "on" yet.
void peer_teardown() {
  const PeerWindow &w = g_peer.win;
  bool any = g_peer.on || w.done != nullptr;
  for (int r = 0; r < kMaxTeam; ++r) any = any || g_peer.opened[r] || w.base[r] != nullptr;
  if (!any) return;
  if (g_ctx.ready) (void)hipStreamSynchronize(g_ctx.stream);
  for (int r = 0; r < w.nranks && r < kMaxTeam; ++r)
    if (r != w.me && g_peer.opened[r]) (void)hipIpcCloseMemHandle(w.base[r]);
  if (w.me >= 0 && w.me < kMaxTeam && w.base[w.me]) (void)hipFree(w.base[w.me]);
  if (w.done) (void)hipFree(w.done);
  (void)hipGetLastError();
  g_peer = PeerX{};
}

inline int pad_ld(int n);
// n: order of the solve the exchange is for.  The peer windows were sized for ek_hip_comm_peer_enable's
// n_max (slots of 2 * pad(n_max) + 8 doubles in every peer's HBM); a larger order would store past the
// slots in other processes' memory, so it takes the collective exchange instead (n is the same on every
// rank: all ranks decide alike).  n = 0: no window exchange will be issued (Cholesky, reduction).
SytrdExchange team_exchange(int nteam, int n = 0) {
  SytrdExchange x{nteam > 0 ? nteam : g_comm.nranks, nullptr, nullptr};
  const bool fits = 2 * (size_t)pad_ld(n > 0 ? n : 1) + 1 <= g_peer.win.maxcount;
  if (nteam == 0 && g_peer.on && g_peer.win.nranks == g_comm.nranks && fits) x.peer = &g_peer.win;
  if (nteam > 0) { x.allreduce = sytrd_team_allreduce; x.allgatherv = team_allgatherv; }
  else if (g_comm.host) { x.allreduce = host_allreduce; x.allgatherv = host_allgatherv; }
  else { x.allreduce = rccl_allreduce; x.allgatherv = rccl_allgatherv; }
  return x;
}

// A rank-local failure (allocation, staging copy) in front of a collective part of a call must not leave
// the other ranks waiting in that collective: every rank contributes its status to one small all-reduce
// over the attached communicator and all of them leave together -- the failing rank with its own code,
// the others with -993.  Returns 0 when every rank is fine.  (The word lives in memory allocated at
// initialisation, so the agreement itself needs nothing that could fail locally.)
// 1 if `local` is non-zero on ANY rank of the team (the same answer on all of them), else 0; < 0: the exchange failed
int comm_any(int local) {
  if (!g_comm.on || g_comm.nranks <= 1) return local ? 1 : 0;
  const SytrdExchange x = team_exchange(0);
  double st = local ? 1.0 : 0.0;
  bool ok = hipMemcpy(g_ctx.d_status, &st, sizeof(double), hipMemcpyHostToDevice) == hipSuccess;
  double *bufs[1] = {g_ctx.d_status};
  g_comm.err = 0;
  x.allreduce(g_ctx.stream, 1, bufs, 1, x.user);
  ok = ok && hipStreamSynchronize(g_ctx.stream) == hipSuccess && !g_comm.err;
  ok = ok && hipMemcpy(&st, g_ctx.d_status, sizeof(double), hipMemcpyDeviceToHost) == hipSuccess;
  if (!ok) return -996;
  return st != 0.0 ? 1 : 0;
}

int comm_agree(int local_rc) {
  if (!g_comm.on || g_comm.nranks <= 1) return local_rc;
  const SytrdExchange x = team_exchange(0);
  double st = local_rc ? 1.0 : 0.0;
  bool ok = hipMemcpy(g_ctx.d_status, &st, sizeof(double), hipMemcpyHostToDevice) == hipSuccess;
  double *bufs[1] = {g_ctx.d_status};
  g_comm.err = 0;
  x.allreduce(g_ctx.stream, 1, bufs, 1, x.user);
  ok = ok && hipStreamSynchronize(g_ctx.stream) == hipSuccess && !g_comm.err;
  ok = ok && hipMemcpy(&st, g_ctx.d_status, sizeof(double), hipMemcpyDeviceToHost) == hipSuccess;
  if (!ok) return local_rc ? local_rc : -996;
  if (st != 0.0) return local_rc ? local_rc : -993;
  return 0;
}

const char *comm_error_string() {
  if (g_comm.host || !g_rccl.GetErrorString) return "exchange through the host hook failed";
  return g_rccl.GetErrorString((ncclResult_t)g_comm.err);
}

// test aid (EK_HIP_TEAM_POISON=1): NaN into every column of the strips a member does not own, to
// prove that the distributed tridiagonalisation never reads them
__global__ void poison_foreign_strips_kernel(int n, double *A, int lda, int P, int rank) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)n * n) return;
  const int r = (int)(idx % n), c = (int)(idx / n);
  if ((c / 128) % P != rank) A[(size_t)r + (size_t)c * lda] = __longlong_as_double(0x7ff8000000000000ll);
}

__global__ void count_mismatch_kernel(int m, int n, const double *X, int ldx, const double *Y, int ldy,
                                      int lower, unsigned long long *count) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)m * n) return;
  const int r = (int)(idx % m), c = (int)(idx / m);
  if (lower && r < c) return;
  const unsigned long long a = __double_as_longlong(X[(size_t)r + (size_t)c * ldx]);
  const unsigned long long b = __double_as_longlong(Y[(size_t)r + (size_t)c * ldy]);
  if (a != b) atomicAdd(count, 1ull);
}

// local piece (nr x nc, lld) <-> its place in the full matrix; blocks of nb rows are contiguous
template <typename F>
void for_each_local_block(int m, int n, int nb, int pr, int nprow, int pc, int npcol, F f) {
  const int nr = numroc0(m, nb, pr, nprow), nc = numroc0(n, nb, pc, npcol);
  for (int lc = 0; lc < nc; ++lc) {
    const size_t gc = (size_t)((lc / nb) * npcol + pc) * nb + lc % nb;
    for (int lr0 = 0; lr0 < nr; lr0 += nb) {
      const size_t gr0 = (size_t)((lr0 / nb) * nprow + pr) * nb;
      f(lr0, lc, gr0, gc, (nr - lr0 < nb) ? nr - lr0 : nb, nr);
    }
  }
}

// M_full (m x n, ldf) <- all ranks' pieces of a block-cyclic matrix, through the host hook.
int gather_full(int m, int n, const double *M_loc, const int *desc, const GridCell &g, double *M_full,
                int ldf) {
  if (!g_allgatherv) return -998;
  const int nb = desc[4], P = g.nprow * g.npcol;
  std::vector<long long> counts(P), displs(P);
  long long tot = 0;
  for (int r = 0; r < P; ++r) {            // ranks in row-major grid order (processes.f90:23, 'R')
    counts[r] = (long long)numroc0(m, nb, r / g.npcol, g.nprow) * numroc0(n, nb, r % g.npcol, g.npcol);
    displs[r] = tot; tot += counts[r];
  }
  const int me = g.myrow * g.npcol + g.mycol;
  double *send = (double *)malloc((size_t)(counts[me] > 0 ? counts[me] : 1) * 8);
  double *recv = (double *)malloc((size_t)(tot > 0 ? tot : 1) * 8);
  if (!send || !recv) { free(send); free(recv); return -1000 - (int)hipErrorOutOfMemory; }
  const int lld = desc[8];
  for_each_local_block(m, n, nb, g.myrow, g.nprow, g.mycol, g.npcol,
                       [&](int lr0, int lc, size_t, size_t, int len, int nr) {
                         memcpy(send + (size_t)lr0 + (size_t)lc * nr, M_loc + (size_t)lr0 + (size_t)lc * lld,
                                (size_t)len * 8);
                       });
  const int rc = g_allgatherv(send, counts[me], recv, counts.data(), displs.data(), g_allgatherv_user);
  if (rc == 0) {
    for (int r = 0; r < P; ++r) {
      const double *piece = recv + displs[r];
      for_each_local_block(m, n, nb, r / g.npcol, g.nprow, r % g.npcol, g.npcol,
                           [&](int lr0, int lc, size_t gr0, size_t gc, int len, int nr) {
                             memcpy(M_full + gr0 + gc * (size_t)ldf, piece + (size_t)lr0 + (size_t)lc * nr,
                                    (size_t)len * 8);
                           });
    }
  }
  free(send); free(recv);
  return rc == 0 ? 0 : -999;
}

// M_loc <- this cell's piece of M_full
void extract_local(int m, int n, const double *M_full, int ldf, const int *desc, const GridCell &g,
                   double *M_loc) {
  const int lld = desc[8];
  for_each_local_block(m, n, desc[4], g.myrow, g.nprow, g.mycol, g.npcol,
                       [&](int lr0, int lc, size_t gr0, size_t gc, int len, int) {
                         memcpy(M_loc + (size_t)lr0 + (size_t)lc * lld, M_full + gr0 + gc * (size_t)ldf,
                                (size_t)len * 8);
                       });
}

inline int pad_ld(int n) {
  static int extra = -1;
  if (extra < 0) { const char *e = getenv("EK_HIP_LDPAD"); extra = e ? atoi(e) : 0; }
  return round_up(n > 0 ? n : 1, 128) + extra;
}

int h2d_matrix(int m, int n, const double *h, int ldh, double *d, int ldd, hipStream_t s) {
  EK_HIP_CHECK(hipMemcpy2DAsync(d, (size_t)ldd * sizeof(double), h, (size_t)ldh * sizeof(double),
                                (size_t)m * sizeof(double), n, hipMemcpyHostToDevice, s));
  return 0;
}
int d2h_matrix(int m, int n, const double *d, int ldd, double *h, int ldh, hipStream_t s) {
  EK_HIP_CHECK(hipMemcpy2DAsync(h, (size_t)ldh * sizeof(double), d, (size_t)ldd * sizeof(double),
                                (size_t)m * sizeof(double), n, hipMemcpyDeviceToHost, s));
  return 0;
}

int fetch_info(int *info) {
  EK_HIP_CHECK(hipMemcpyAsync(info, g_ctx.d_info, sizeof(int), hipMemcpyDeviceToHost, g_ctx.stream));
  EK_HIP_CHECK(hipStreamSynchronize(g_ctx.stream));
  return 0;
}

}  // namespace

extern "C" {

int ek_hip_version(void) { return 1; }

int ek_hip_init(int device) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (g_ctx.ready && g_ctx.device == device) return 0;
  if (g_ctx.ready) return -1;   // already bound to another device
  g_ctx.device = device;
  return ensure_init();
}

int ek_hip_finalize(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_ctx.ready) return 0;
  (void)hipStreamSynchronize(g_ctx.stream);
  if (g_ctx.ws_alloc) (void)hipFree(g_ctx.ws_alloc);
  g_ctx.ws = nullptr; g_ctx.ws_alloc = nullptr; g_ctx.ws_bytes = 0;
  release_scratch_choice();
  // a communicator and its peer windows do not outlive the library's device state
  peer_teardown();
  if (g_comm.on && !g_comm.host && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(g_comm.comm);
  g_comm = Comm{};
  return 0;
}

const char *ek_hip_stage_name(int stage) {
  static const char *names[EK_HIP_N_STAGES] = {
      "reduce_generalized:pdpotrf", "reduce_generalized:pdsygst",
      "eigen_solver_scalapack_all:pdsytrd", "eigen_solver_scalapack_all:gather1",
      "eigen_solver_scalapack_all:pdstedc", "eigen_solver_scalapack_all:pdormtr",
      "recovery_generalized", "ek_hip:host_device_copies"};
  return (stage >= 0 && stage < EK_HIP_N_STAGES) ? names[stage] : "";
}

int ek_hip_malloc(void **dptr, unsigned long long bytes) {
  int rc = ensure_init(); if (rc) return rc;
  EK_HIP_CHECK(hipMalloc(dptr, bytes));
  return 0;
}
int ek_hip_free(void *dptr) { EK_HIP_CHECK(hipFree(dptr)); return 0; }
int ek_hip_memcpy_h2d(void *dst, const void *src, unsigned long long bytes) {
  int rc = ensure_init(); if (rc) return rc;
  EK_HIP_CHECK(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
  return 0;
}
int ek_hip_memcpy_d2h(void *dst, const void *src, unsigned long long bytes) {
  int rc = ensure_init(); if (rc) return rc;
  EK_HIP_CHECK(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
  return 0;
}
int ek_hip_synchronize(void) {
  int rc = ensure_init(); if (rc) return rc;
  EK_HIP_CHECK(hipDeviceSynchronize());
  return 0;
}

int ek_hip_dgemm(int transa, int transb, int m, int n, int k, double alpha, const double *A,
                 int lda, const double *B, int ldb, double beta, double *C, int ldc,
                 int lower_only) {
  if (m < 0) return -3; if (n < 0) return -4; if (k < 0) return -5;
  const int ar = transa ? k : m, ac = transa ? m : k, br = transb ? n : k, bc = transb ? k : n;
  if (lda < (ar > 1 ? ar : 1)) return -8;
  if (ldb < (br > 1 ? br : 1)) return -10;
  if (ldc < (m > 1 ? m : 1)) return -13;
  int rc = ensure_init(); if (rc) return rc;
  if (m == 0 || n == 0) return 0;
  std::lock_guard<std::mutex> lk(g_mu);
  hipStream_t s = g_ctx.stream;
  const int pa = pad_ld(ar), pb = pad_ld(br), pc = pad_ld(m);
  void *ws;
  rc = workspace(al((size_t)pa * (ac > 0 ? ac : 1) * 8) + al((size_t)pb * (bc > 0 ? bc : 1) * 8) +
                 al((size_t)pc * n * 8), &ws);
  if (rc) return rc;
  Arena ar_(ws, g_ctx.ws_bytes);
  double *dA = ar_.get<double>((size_t)pa * (ac > 0 ? ac : 1));
  double *dB = ar_.get<double>((size_t)pb * (bc > 0 ? bc : 1));
  double *dC = ar_.get<double>((size_t)pc * n);
  if (k > 0) {
    rc = h2d_matrix(ar, ac, A, lda, dA, pa, s); if (rc) return rc;
    rc = h2d_matrix(br, bc, B, ldb, dB, pb, s); if (rc) return rc;
  }
  rc = h2d_matrix(m, n, C, ldc, dC, pc, s); if (rc) return rc;
  gemm(s, transa != 0, transb != 0, m, n, k, alpha, dA, pa, dB, pb, beta, dC, pc, lower_only != 0);
  EK_HIP_CHECK(hipGetLastError());
  rc = d2h_matrix(m, n, dC, pc, C, ldc, s); if (rc) return rc;
  EK_HIP_CHECK(hipStreamSynchronize(s));
  return 0;
}

int ek_hip_potrf(int n, double *B_loc, const int desc_B[9]) {
  if (n < 0) return -1;
  if (!B_loc && n > 0) return -2;
  int rc = check_desc(desc_B, 3, n, n); if (rc) return rc;
  rc = ensure_init(); if (rc) return rc;
  if (n == 0) return 0;
  std::lock_guard<std::mutex> lk(g_mu);
  hipStream_t s = g_ctx.stream;
  const int ld = pad_ld(n), nblk = ceil_div(n, kDiagNB);
  void *ws;
  const size_t rlb = al(potrf_rl_work_bytes(n, ld));
  rc = workspace(al((size_t)ld * n * 8) + al((size_t)nblk * kDiagNB * kDiagNB * 8) +
                 al((size_t)128 * ld * 8) + rlb, &ws);
  if (rc) return rc;
  Arena a(ws, g_ctx.ws_bytes);
  double *dB = a.get<double>((size_t)ld * n);
  double *dInv = a.get<double>((size_t)nblk * kDiagNB * kDiagNB);
  double *work = a.get<double>((size_t)128 * ld);
  char *rlwork = a.get<char>(rlb);
  rc = h2d_matrix(n, n, B_loc, desc_B[8], dB, ld, s); if (rc) return rc;
  EK_HIP_CHECK(hipMemsetAsync(g_ctx.d_info, 0, sizeof(int), s));
  if (n >= kPotrfRlMin) potrf_lower_rl(s, g_ctx.stream2, n, dB, ld, dInv, g_ctx.d_info, rlwork);
  else potrf_lower(s, n, dB, ld, dInv, g_ctx.d_info, work);
  EK_HIP_CHECK(hipGetLastError());
  rc = d2h_matrix(n, n, dB, ld, B_loc, desc_B[8], s); if (rc) return rc;
  int info = 0;
  rc = fetch_info(&info); if (rc) return rc;
  return info;
}

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
  potrf_lower_dist(s, n, nmem, mem, team_exchange(nteam));
  EK_HIP_CHECK(hipGetLastError());
  for (int m = 1; m < nmem; ++m) {
    hipLaunchKernelGGL(count_mismatch_kernel, dim3((unsigned)(((size_t)n * n + 255) / 256)), dim3(256), 0, s, n, n,
                       mem[0].B, ld, mem[m].B, ld, 1, d_cnt);
    hipLaunchKernelGGL(count_mismatch_kernel, dim3((unsigned)(((size_t)nblk * kDiagNB * kDiagNB + 255) / 256)), dim3(256),
                       0, s, nblk * kDiagNB * kDiagNB, 1, mem[0].invdiag, 1, mem[m].invdiag, 1, 0, d_cnt);
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

int ek_hip_sygst(int n, double *A_loc, const int desc_A[9], const double *L_loc,
                 const int desc_B[9], double *scale) {
  if (n < 0) return -1;
  if (!A_loc && n > 0) return -2;
  int rc = check_desc(desc_A, 3, n, n); if (rc) return rc;
  if (!L_loc && n > 0) return -4;
  rc = check_desc(desc_B, 5, n, n); if (rc) return rc;
  rc = ensure_init(); if (rc) return rc;
  if (scale) *scale = 1.0;
  if (n == 0) return 0;
  std::lock_guard<std::mutex> lk(g_mu);
  hipStream_t s = g_ctx.stream;
  const int ld = pad_ld(n), nblk = ceil_div(n, kDiagNB);
  void *ws;
  rc = workspace(2 * al((size_t)ld * n * 8) + al((size_t)nblk * kDiagNB * kDiagNB * 8) +
                 al((size_t)128 * ld * 8) + al(sygst_scratch_doubles(n) * 8), &ws);
  if (rc) return rc;
  Arena a(ws, g_ctx.ws_bytes);
  double *dA = a.get<double>((size_t)ld * n);
  double *dL = a.get<double>((size_t)ld * n);
  double *dInv = a.get<double>((size_t)nblk * kDiagNB * kDiagNB);
  double *work = a.get<double>((size_t)128 * ld);
  double *scr = a.get<double>(sygst_scratch_doubles(n));
  rc = h2d_matrix(n, n, A_loc, desc_A[8], dA, ld, s); if (rc) return rc;
  rc = h2d_matrix(n, n, L_loc, desc_B[8], dL, ld, s); if (rc) return rc;
  trtri_diag_blocks(s, n, dL, ld, dInv);
  sygst_lower(s, n, dA, ld, dL, ld, dInv, work, scr);
  EK_HIP_CHECK(hipGetLastError());
  rc = d2h_matrix(n, n, dA, ld, A_loc, desc_A[8], s); if (rc) return rc;
  EK_HIP_CHECK(hipStreamSynchronize(s));
  return 0;
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

int ek_hip_trtrs(int n, int nrhs, const double *L_loc, const int desc_B[9], double *Z_loc,
                 const int desc_Z[9]) {
  if (n < 0) return -1;
  if (nrhs < 0) return -2;
  if (!L_loc && n > 0) return -3;
  int rc = check_desc(desc_B, 4, n, n); if (rc) return rc;
  if (!Z_loc && n > 0 && nrhs > 0) return -5;
  if (!desc_Z) return -6;
  rc = check_desc(desc_Z, 6, n, desc_Z[3]); if (rc) return rc;
  if (desc_Z[3] < nrhs) return -604;
  rc = ensure_init(); if (rc) return rc;
  if (n == 0 || nrhs == 0) return 0;
  for (int i = 0; i < n; ++i)   // PDTRTRS singularity check: info = i if L(i,i) == 0
    if (L_loc[(size_t)i + (size_t)i * desc_B[8]] == 0.0) return i + 1;
  std::lock_guard<std::mutex> lk(g_mu);
  hipStream_t s = g_ctx.stream;
  const int ld = pad_ld(n), nblk = ceil_div(n, kDiagNB);
  void *ws;
  rc = workspace(al((size_t)ld * n * 8) + al((size_t)ld * nrhs * 8) +
                 al((size_t)nblk * kDiagNB * kDiagNB * 8) + al((size_t)128 * (ld > nrhs ? ld : nrhs) * 8), &ws);
  if (rc) return rc;
  Arena a(ws, g_ctx.ws_bytes);
  double *dL = a.get<double>((size_t)ld * n);
  double *dZ = a.get<double>((size_t)ld * nrhs);
  double *dInv = a.get<double>((size_t)nblk * kDiagNB * kDiagNB);
  double *work = a.get<double>((size_t)128 * (ld > nrhs ? ld : nrhs));
  rc = h2d_matrix(n, n, L_loc, desc_B[8], dL, ld, s); if (rc) return rc;
  rc = h2d_matrix(n, nrhs, Z_loc, desc_Z[8], dZ, ld, s); if (rc) return rc;
  trtri_diag_blocks(s, n, dL, ld, dInv);
  trsm_llt(s, n, nrhs, dL, ld, dInv, dZ, ld, work);
  EK_HIP_CHECK(hipGetLastError());
  rc = d2h_matrix(n, nrhs, dZ, ld, Z_loc, desc_Z[8], s); if (rc) return rc;
  EK_HIP_CHECK(hipStreamSynchronize(s));
  return 0;
}


int ek_hip_sytrd(int n, double *A_loc, const int desc_A[9], double *d, double *e, double *tau) {
  if (n < 0) return -1;
  if (!A_loc && n > 0) return -2;
  int rc = check_desc(desc_A, 3, n, n); if (rc) return rc;
  if (n > 0 && !d) return -4;
  if (n > 1 && !e) return -5;
  if (n > 1 && !tau) return -6;
  rc = ensure_init(); if (rc) return rc;
  if (n == 0) return 0;
  std::lock_guard<std::mutex> lk(g_mu);
  hipStream_t s = g_ctx.stream;
  const int ld = pad_ld(n);
  const size_t wb = sytrd_work_bytes(n);
  void *ws;
  rc = workspace(al((size_t)ld * ld * 8) + al(wb) + 3 * al((size_t)ld * 8), &ws);
  if (rc) return rc;
  Arena a(ws, g_ctx.ws_bytes);
  double *dA = a.get<double>((size_t)ld * ld);
  char *work = a.get<char>(wb);
  double *dd = a.get<double>(ld), *de = a.get<double>(ld), *dt = a.get<double>(ld);
  EK_HIP_CHECK(hipMemsetAsync(dA, 0, (size_t)ld * ld * 8, s));
  EK_HIP_CHECK(hipMemsetAsync(dd, 0, 3 * al((size_t)ld * 8), s));
  rc = h2d_matrix(n, n, A_loc, desc_A[8], dA, ld, s); if (rc) return rc;
  sytrd_lower(s, n, dA, ld, dd, de, dt, nullptr, 0, work);
  EK_HIP_CHECK(hipGetLastError());
  rc = d2h_matrix(n, n, dA, ld, A_loc, desc_A[8], s); if (rc) return rc;
  EK_HIP_CHECK(hipMemcpyAsync(d, dd, (size_t)n * 8, hipMemcpyDeviceToHost, s));
  if (n > 1) {
    EK_HIP_CHECK(hipMemcpyAsync(e, de, (size_t)(n - 1) * 8, hipMemcpyDeviceToHost, s));
    EK_HIP_CHECK(hipMemcpyAsync(tau, dt, (size_t)(n - 1) * 8, hipMemcpyDeviceToHost, s));
  }
  EK_HIP_CHECK(hipStreamSynchronize(s));
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
      hipLaunchKernelGGL(poison_foreign_strips_kernel, dim3((unsigned)(((size_t)n * n + 255) / 256)), dim3(256), 0, s,
                         n, dA, ld, P, mem[m].rank);
  }
  const SytrdExchange x = team_exchange(nteam, n);
  g_comm.err = 0;
  sytrd_lower_dist(s, n, nmem, mem, x);
  EK_HIP_CHECK(hipGetLastError());
  for (int m = 1; m < nmem; ++m) {
    const unsigned nb = (unsigned)(((size_t)n * n + 255) / 256);
    hipLaunchKernelGGL(count_mismatch_kernel, dim3(nb), dim3(256), 0, s, n, n, mem[0].A, ld, mem[m].A, ld, 1, d_cnt);
    hipLaunchKernelGGL(count_mismatch_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, s, n, 1, mem[0].d, n, mem[m].d, n, 0, d_cnt);
    if (n > 1) {
      hipLaunchKernelGGL(count_mismatch_kernel, dim3(ceil_div(n - 1, 256)), dim3(256), 0, s, n - 1, 1, mem[0].e, n, mem[m].e, n, 0, d_cnt);
      hipLaunchKernelGGL(count_mismatch_kernel, dim3(ceil_div(n - 1, 256)), dim3(256), 0, s, n - 1, 1, mem[0].tau, n, mem[m].tau, n, 0, d_cnt);
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
  peer_teardown();
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
  peer_teardown();
  if (g_comm.on && !g_comm.host) (void)g_rccl.CommDestroy(g_comm.comm);
  g_comm = Comm{};
  g_comm.on = true; g_comm.host = true; g_comm.nranks = nranks; g_comm.rank = rank;
  return 0;
}

// Peer windows for the attached communicator (collective call).  n_max = largest matrix order that
// will be solved while they are enabled.  The handles travel through the communicator itself.
int ek_hip_comm_peer_enable(int n_max) {
  if (n_max < 1) return -1;
  int rc = ensure_init(); if (rc) return rc;
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_comm.on) return -995;
  peer_teardown();
  hipStream_t s = g_ctx.stream;
  const int P = g_comm.nranks, me = g_comm.rank;
  PeerWindow &w = g_peer.win;
  w.nranks = P; w.me = me; w.slots_off = kPeerFlagBytes;
  w.maxcount = 2 * (size_t)pad_ld(n_max) + 8;
  w.seq = &g_peer.seq; w.signal = peer_signal; w.user = nullptr;
  g_peer.bytes = kPeerFlagBytes + (size_t)P * 2 * w.maxcount * sizeof(double);
  // A rank whose local step fails keeps taking part in the exchanges below and says so in its
  // status word, so that all ranks give up together (-993) instead of waiting for each other.
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "handle size");
  constexpr int kRec = 9;                    // doubles per rank: 8 = the 64-byte handle, 1 = status
  double rec[kMaxTeam * kRec];
  memset(rec, 0, sizeof(rec));
  bool ok = true;
  char *mine = nullptr;
  ok = ok && hipExtMallocWithFlags((void **)&mine, g_peer.bytes, hipDeviceMallocFinegrained) == hipSuccess;
  ok = ok && hipMemset(mine, 0, g_peer.bytes) == hipSuccess;
  ok = ok && hipMalloc((void **)&w.done, 256) == hipSuccess && hipMemset(w.done, 0, 256) == hipSuccess;
  ok = ok && hipDeviceSynchronize() == hipSuccess;
  w.base[me] = mine;
  if (ok && P > 1) {
    hipIpcMemHandle_t h;
    ok = hipIpcGetMemHandle(&h, mine) == hipSuccess;
    if (ok) memcpy(&rec[me * kRec], &h, sizeof(h));
  }
  rec[me * kRec + 8] = ok ? 0.0 : 1.0;
  (void)hipGetLastError();
  double *dh = nullptr;
  DevMem mem;
  rc = mem.alloc(&dh, sizeof(rec));
  if (rc) { peer_teardown(); return rc; }
  size_t offs[kMaxTeam], counts[kMaxTeam];
  for (int r = 0; r < P; ++r) { offs[r] = (size_t)r * kRec; counts[r] = kRec; }
  double *bufs[1] = {dh};
  const SytrdExchange x = team_exchange(0);
  auto exchange_status = [&]() -> int {     // everyone's record; returns the number of ranks that failed, or < 0
    if (hipMemcpy(dh, rec, sizeof(rec), hipMemcpyHostToDevice) != hipSuccess) return -1;
    g_comm.err = 0;
    if (P > 1) x.allgatherv(s, 1, me, bufs, offs, counts, P, x.user);
    if (hipStreamSynchronize(s) != hipSuccess || g_comm.err) return -1;
    if (hipMemcpy(rec, dh, sizeof(rec), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    int bad = 0;
    for (int r = 0; r < P; ++r) if (rec[r * kRec + 8] != 0.0) ++bad;
    return bad;
  };
  int bad = exchange_status();
  if (bad != 0) { peer_teardown(); return bad < 0 ? -996 : -993; }
  for (int r = 0; r < P && ok; ++r) {
    if (r == me) continue;
    hipIpcMemHandle_t h;
    memcpy(&h, &rec[r * kRec], sizeof(h));
    ok = hipIpcOpenMemHandle((void **)&w.base[r], h, hipIpcMemLazyEnablePeerAccess) == hipSuccess;
    if (ok) g_peer.opened[r] = true;
  }
  (void)hipGetLastError();
  // nobody stores into a peer before every rank has mapped every area -- and has said so
  rec[me * kRec + 8] = ok ? 0.0 : 1.0;
  bad = exchange_status();
  if (bad != 0) { peer_teardown(); return bad < 0 ? -996 : -993; }
  g_peer.seq = 0;
  g_peer.on = true;
  return 0;
}

int ek_hip_comm_peer_disable(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  peer_teardown();
  return 0;
}

int ek_hip_comm_size(void) { return g_comm.on ? g_comm.nranks : 0; }
int ek_hip_comm_rank(void) { return g_comm.on ? g_comm.rank : -1; }

int ek_hip_comm_destroy(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  peer_teardown();
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


int ek_hip_stedc(int n, double *d, double *e, double *Z_loc, const int desc_Z[9]) {
  if (n < 0) return -1;
  if (n > 0 && !d) return -2;
  if (n > 1 && !e) return -3;
  if (n > 0 && !Z_loc) return -4;
  int rc = check_desc(desc_Z, 5, n, n); if (rc) return rc;
  rc = ensure_init(); if (rc) return rc;
  if (n == 0) return 0;
  std::lock_guard<std::mutex> lk(g_mu);
  hipStream_t s = g_ctx.stream;
  const int ld = pad_ld(n);
  const size_t wb = stedc_work_bytes(n);
  void *ws;
  rc = workspace(al((size_t)ld * n * 8) + al(wb) + 3 * al((size_t)ld * 8), &ws);
  if (rc) return rc;
  Arena a(ws, g_ctx.ws_bytes);
  double *dZ = a.get<double>((size_t)ld * n);
  char *work = a.get<char>(wb);
  double *dd = a.get<double>(ld), *de = a.get<double>(ld), *dw = a.get<double>(ld);
  EK_HIP_CHECK(hipMemcpyAsync(dd, d, (size_t)n * 8, hipMemcpyHostToDevice, s));
  EK_HIP_CHECK(hipMemsetAsync(de, 0, (size_t)ld * 8, s));
  if (n > 1) EK_HIP_CHECK(hipMemcpyAsync(de, e, (size_t)(n - 1) * 8, hipMemcpyHostToDevice, s));
  EK_HIP_CHECK(hipMemsetAsync(g_ctx.d_info, 0, 4 * sizeof(int), s));
  stedc(s, n, dd, de, dw, dZ, ld, work, g_ctx.d_info);
  EK_HIP_CHECK(hipGetLastError());
  rc = d2h_matrix(n, n, dZ, ld, Z_loc, desc_Z[8], s); if (rc) return rc;
  EK_HIP_CHECK(hipMemcpyAsync(d, dw, (size_t)n * 8, hipMemcpyDeviceToHost, s));
  int info = 0;
  rc = fetch_info(&info); if (rc) return rc;
  return info;
}

int ek_hip_ormtr(int n, int ncols, const double *A_loc, const int desc_A[9], const double *tau,
                 double *Z_loc, const int desc_Z[9]) {
  if (n < 0) return -1;
  if (ncols < 0) return -2;
  if (n > 0 && !A_loc) return -3;
  int rc = check_desc(desc_A, 4, n, n); if (rc) return rc;
  if (n > 1 && !tau) return -5;
  if (n > 0 && ncols > 0 && !Z_loc) return -6;
  if (!desc_Z) return -7;
  rc = check_desc(desc_Z, 7, n, desc_Z[3]); if (rc) return rc;
  if (desc_Z[3] < ncols) return -704;
  rc = ensure_init(); if (rc) return rc;
  if (n <= 1 || ncols == 0) return 0;
  std::lock_guard<std::mutex> lk(g_mu);
  hipStream_t s = g_ctx.stream;
  const int ld = pad_ld(n);
  const size_t wb = ormtr_work_bytes(n, ncols);
  void *ws;
  rc = workspace(2 * al((size_t)ld * ld * 8) + al((size_t)ld * ncols * 8) + al(wb) + al((size_t)ld * 8), &ws);
  if (rc) return rc;
  Arena a(ws, g_ctx.ws_bytes);
  double *dA = a.get<double>((size_t)ld * ld);
  double *dV = a.get<double>((size_t)ld * ld);
  double *dZ = a.get<double>((size_t)ld * ncols);
  char *work = a.get<char>(wb);
  double *dt = a.get<double>(ld);
  rc = h2d_matrix(n, n, A_loc, desc_A[8], dA, ld, s); if (rc) return rc;
  rc = h2d_matrix(n, ncols, Z_loc, desc_Z[8], dZ, ld, s); if (rc) return rc;
  EK_HIP_CHECK(hipMemsetAsync(dt, 0, (size_t)ld * 8, s));
  EK_HIP_CHECK(hipMemcpyAsync(dt, tau, (size_t)(n - 1) * 8, hipMemcpyHostToDevice, s));
  EK_HIP_CHECK(hipMemsetAsync(dV, 0, (size_t)ld * ld * 8, s));
  build_explicit_v(s, n, dA, ld, dV, ld);
  ormtr_lower(s, n, ncols, dV, ld, dt, dZ, ld, work);
  EK_HIP_CHECK(hipGetLastError());
  rc = d2h_matrix(n, ncols, dZ, ld, Z_loc, desc_Z[8], s); if (rc) return rc;
  EK_HIP_CHECK(hipStreamSynchronize(s));
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
// all-gather of the packed band of a team whose members hold the columns of their own strips (strip S on rank S mod P):
// one exchange per round of P strips
void gather_band_strips(hipStream_t s, int n, int nmem, int rank0, double *const *ABs, const SytrdExchange &x) {
  const int NRB = ceil_div(n, 128), P = x.nranks;
  if (P <= 1) return;
  for (int q = 0; q * P < NRB; ++q) {
    size_t offs[kMaxTeam], counts[kMaxTeam];
    for (int r = 0; r < P; ++r) {
      const int S = q * P + r;
      const int cols = (S < NRB) ? ((n - S * 128 < 128) ? n - S * 128 : 128) : 0;
      offs[r] = (S < NRB) ? (size_t)S * 128 * kBandLd : 0; counts[r] = (size_t)cols * kBandLd;
    }
    x.allgatherv(s, nmem, rank0, ABs, offs, counts, P, x.user);
  }
}

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
      hipLaunchKernelGGL(poison_foreign_strips_kernel, dim3((unsigned)(((size_t)n * n + 255) / 256)), dim3(256), 0, s,
                         n, dA, ld, P, mem[m].rank);
  }
  const SytrdExchange x = team_exchange(nteam, 0);
  g_comm.err = 0;
  sy2sb_lower_dist(s, n, nmem, mem, x);
  for (int m = 0; m < nmem; ++m) pack_band(s, n, mem[m].A, ld, ABs[m]);
  gather_band_strips(s, n, nmem, mem[0].rank, ABs, x);
  EK_HIP_CHECK(hipGetLastError());
  for (int m = 1; m < nmem; ++m) {
    const unsigned nb = (unsigned)(((size_t)n * n + 255) / 256);
    hipLaunchKernelGGL(count_mismatch_kernel, dim3(nb), dim3(256), 0, s, n, n, mem[0].Vall, ld, mem[m].Vall, ld, 0, d_cnt);
    hipLaunchKernelGGL(count_mismatch_kernel, dim3((unsigned)(((size_t)kBandLd * n + 255) / 256)), dim3(256), 0, s, kBandLd, n,
                       ABs[0], kBandLd, ABs[m], kBandLd, 0, d_cnt);
    hipLaunchKernelGGL(count_mismatch_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, s, n, 1, mem[0].tau1, n, mem[m].tau1, n, 0, d_cnt);
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
// (0 q2_apply_kernel, 1 chase_kernel, 2 symm_lower_kernel of every 8th panel); _get after the solves.
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
    set_matrix(s, n, ncols, 0.0, 1.0, dZ, ld);
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
int ek_hip_debug_sy2sb_team_timing(int n, int nteam, int reps, double *seconds) {
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
  for (int r = 0; r < reps; ++r) {
    EK_HIP_CHECK(hipMemsetAsync(d_flags, 0, kMaxTeam * sizeof(int), s));
    for (int m = 0; m < nmem; ++m) {
      EK_HIP_CHECK(hipMemsetAsync(mem[m].A, 0, (size_t)ld * ld * 8, s));
      EK_HIP_CHECK(hipMemsetAsync(mem[m].Vall, 0, (size_t)ld * ld * 8, s));
      EK_HIP_CHECK(hipMemsetAsync(mem[m].tau1, 0, (size_t)ld * 8, s));
      synth_matrix(s, n, 1, mem[m].A, ld);
    }
    EK_HIP_CHECK(hipEventRecord(e0, s));
    sy2sb_lower_dist(s, n, nmem, mem, x);
    EK_HIP_CHECK(hipEventRecord(e1, s));
    EK_HIP_CHECK(hipStreamSynchronize(s));
    float ms = 0.f; EK_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    tot += ms * 1e-3;
  }
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  if (seconds) *seconds = tot / (reps > 0 ? reps : 1);
  return g_comm.err ? -996 : 0;
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
    potrf_lower_dist(s, n, nmem, pm, x);
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

int ek_hip_synth_matrix_device(int n, unsigned long long seed, double *dM, int ldm) {
  if (n < 0) return -1;
  if (ldm < (n > 1 ? n : 1)) return -4;
  int rc = ensure_init(); if (rc) return rc;
  std::lock_guard<std::mutex> lk(g_mu);
  synth_matrix(g_ctx.stream, n, seed, dM, ldm);
  EK_HIP_CHECK(hipGetLastError());
  EK_HIP_CHECK(hipStreamSynchronize(g_ctx.stream));
  return 0;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------
// Whole path on device-resident data.
namespace {

// Staging pipeline of ek_hip_solve on a 1 x 1 grid (host arrays in, host arrays out: solver_main.f90:64-65).  The
// copies run on worker threads with their own non-blocking streams while the main thread issues the stages:
//   in : B, then A (B is needed first: the Cholesky factorisation runs while A is still on its way);
//   out: L as soon as it is final (it leaves during the reduction), the reflectors / band of A after the
//        tridiagonalisation, Z in column slabs as the last stage finishes them, w last.
// A copy of pageable host memory keeps its calling thread busy (the runtime stages it through pinned buffers), which
// is why the copies have threads of their own; a matrix is cut into column pieces so that two threads share it.
struct HostPipe {
  struct Job { double *dev; int ldd; double *host; int ldh; int m, n; hipEvent_t after; int tag; bool to_host; };
  static constexpr int kThreads = 2;         // per direction
  std::mutex mu;
  std::condition_variable cv;
  std::deque<Job> in_q, out_q;
  int pending_in[2] = {0, 0};                // tag 0 = B, 1 = A: pieces not yet in HBM
  int pending_out = 0;
  bool closing = false;
  int err = 0;
  std::vector<std::thread> th;
  hipStream_t cs[2 * kThreads] = {};
  int device = 0;
  int z_slab = 2048;

  int start(int dev) {
    device = dev;
    for (auto &c : cs) EK_HIP_CHECK(hipStreamCreateWithFlags(&c, hipStreamNonBlocking));
    for (int i = 0; i < 2 * kThreads; ++i) th.emplace_back([this, i]() { run(i < kThreads, cs[i]); });
    return 0;
  }
  void run(bool input, hipStream_t c) {
    (void)hipSetDevice(device);
    std::deque<Job> &q = input ? in_q : out_q;
    while (true) {
      Job j;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&]() { return !q.empty() || closing; });
        if (q.empty()) return;
        j = q.front(); q.pop_front();
      }
      hipError_t e = hipSuccess;
      if (j.after) e = hipEventSynchronize(j.after);
      if (e == hipSuccess && j.m > 0 && j.n > 0) {
        if (j.to_host)
          e = hipMemcpy2DAsync(j.host, (size_t)j.ldh * 8, j.dev, (size_t)j.ldd * 8, (size_t)j.m * 8, j.n, hipMemcpyDeviceToHost, c);
        else
          e = hipMemcpy2DAsync(j.dev, (size_t)j.ldd * 8, j.host, (size_t)j.ldh * 8, (size_t)j.m * 8, j.n, hipMemcpyHostToDevice, c);
        if (e == hipSuccess) e = hipStreamSynchronize(c);
      }
      {
        std::lock_guard<std::mutex> lk(mu);
        if (e != hipSuccess && !err) err = -1000 - (int)e;
        if (input) --pending_in[j.tag]; else --pending_out;
      }
      cv.notify_all();
    }
  }
  // host array (m x n, ldh) <-> device image (ldd), cut into column pieces for the threads
  void push(bool to_host, double *dev, int ldd, double *host, int ldh, int m, int n, hipEvent_t after, int tag) {
    const int pieces = (n >= 256) ? 2 * kThreads : 1;
    {
      std::lock_guard<std::mutex> lk(mu);
      for (int p = 0; p < pieces; ++p) {
        const int c0 = (int)((long long)n * p / pieces), c1 = (int)((long long)n * (p + 1) / pieces);
        Job j{dev + (size_t)c0 * ldd, ldd, host + (size_t)c0 * ldh, ldh, m, c1 - c0, after, tag, to_host};
        if (to_host) { out_q.push_back(j); ++pending_out; } else { in_q.push_back(j); ++pending_in[tag]; }
      }
    }
    cv.notify_all();
  }
  int wait_in(int tag) {
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [&]() { return pending_in[tag] == 0; });
    return err;
  }
  int finish() {                              // all copies done; threads joined; streams released
    {
      std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, [&]() { return pending_out == 0 && pending_in[0] == 0 && pending_in[1] == 0; });
      closing = true;
    }
    cv.notify_all();
    for (auto &t : th) t.join();
    th.clear();
    for (auto &c : cs) if (c) { (void)hipStreamDestroy(c); c = nullptr; }
    for (auto &e : evs) (void)hipEventDestroy(e);
    evs.clear();
    return err;
  }
  ~HostPipe() { if (!th.empty()) (void)finish(); }
  // an event recorded on stream s now (the device image is final there)
  std::vector<hipEvent_t> evs;
  hipEvent_t mark(hipStream_t s) {
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
    (void)hipEventRecord(e, s);
    evs.push_back(e);
    return e;
  }
  // what the solve hands over: host destinations of the in-place results
  double *hA = nullptr, *hB = nullptr, *hZ = nullptr; int ldha = 0, ldhb = 0, ldhz = 0;
};

struct StageTimer {      // events are released when the timer goes out of scope, whichever way the call ends
  hipEvent_t ev[EK_HIP_N_STAGES + 1];
  int made = 0;
  bool on = false;
  int init() {
    for (auto &e : ev) { EK_HIP_CHECK(hipEventCreate(&e)); ++made; }
    on = true; return 0;
  }
  void destroy() { for (int i = 0; i < made; ++i) (void)hipEventDestroy(ev[i]); made = 0; on = false; }
  ~StageTimer() { destroy(); }
};

// Runs the path on user device arrays dA, dB, dZ (column-major, any ld >= n) by way of padded
// internal work arrays (ld multiple of 128, zero padding), so the kernels see aligned tiles.
//
// cell == nullptr: dZ receives the first n_vec eigenvectors (n x n_vec).  Otherwise the reduction
// and the tridiagonal eigenproblem are computed as usual (replicated on every rank) and only the
// eigenvector columns this grid cell owns are back-transformed; dZ receives the local
// block-cyclic piece numroc(n, nb, myrow, nprow) x numroc(n_vec, nb, mycol, npcol).
int solve_device_locked(int problem, int n, int n_vec, double *dA, int lda, double *dB, int ldb,
                        double *dw, double *dZ, int ldz, double *stage_seconds, int n_stages,
                        const GridCell *cell = nullptr, HostPipe *pipe = nullptr) {
  hipStream_t s = g_ctx.stream;
  const int ld = pad_ld(n), nblk = ceil_div(n, kDiagNB);
  const int nc_loc = cell ? numroc0(n_vec, cell->nb, cell->mycol, cell->npcol) : n_vec;
  const int nr_loc = cell ? numroc0(n, cell->nb, cell->myrow, cell->nprow) : n;
  // A communicator attached by the host (ek_hip_comm_init) whose size is the grid's: the
  // tridiagonalisation is distributed over the ranks (one RCCL all-reduce per column); the other
  // stages are as in the replicated-input mode.
  const bool dist = cell && g_comm.on && g_comm.nranks == cell->nprow * cell->npcol;
  if (dist && g_comm.rank != cell->myrow * cell->npcol + cell->mycol) return -994;
  const size_t wb_sytrd = dist ? sytrd_dist_work_bytes(n, g_comm.nranks) : sytrd_work_bytes(n),
               wb_stedc = stedc_work_bytes(n), wb_ormtr = ormtr_work_bytes(n, nc_loc, n_vec);
  const size_t mat = al((size_t)ld * ld * 8);
  size_t scratch = wb_sytrd;
  if (wb_stedc > scratch) scratch = wb_stedc;
  if (wb_ormtr > scratch) scratch = wb_ormtr;
  const size_t trsm_work = al((size_t)128 * ld * 8);
  void *ws;
  size_t sygst_dbl = (problem == 1) ? sygst_scratch_doubles(n) : 0;
  if (problem == 1 && dist) {
    const size_t dd = sygst_dist_scratch_doubles(n, ld, g_comm.nranks);
    if (dd > sygst_dbl) sygst_dbl = dd;
  }
  const size_t sygst_scr = al(sygst_dbl * 8);
  // right-looking Cholesky with look-ahead from this order on (below it the recursion is as fast)
  const bool potrf_rl = problem == 1 && n >= kPotrfRlMin;
  size_t potrf_wb = (problem == 1 && dist) ? al(potrf_dist_work_bytes(n, ld, g_comm.nranks)) : 0;
  if (potrf_rl && al(potrf_rl_work_bytes(n, ld)) > potrf_wb) potrf_wb = al(potrf_rl_work_bytes(n, ld));
  int rc = 0;
  // two-stage tridiagonalisation: one more matrix for the reflectors of the bulge chasing, a copy of
  // the reduced matrix for the (rare) fall-back to the one-stage path, and the stages' own scratch
  const int ts_min = two_stage_min();
  const bool two_stage = ts_min > 0 && n >= ts_min && n >= 3;
  const size_t wb_sy2sb = two_stage ? al(dist ? sy2sb_dist_work_bytes(n, g_comm.nranks) : sy2sb_work_bytes(n)) : 0,
               wb_sb2st = two_stage ? al(sb2st_work_bytes(n)) : 0;
  const size_t wb_q1prep = two_stage ? al(ormtr_prep_bytes(n)) : 0;
  const size_t ws_need = 4 * mat + al((size_t)nblk * kDiagNB * kDiagNB * 8) + trsm_work + al(scratch) +
                         4 * al((size_t)ld * 8) + sygst_scr + potrf_wb +
                         (two_stage ? 2 * mat + wb_sy2sb + wb_sb2st + wb_q1prep + al((size_t)ld * 8) : 0);
  rc = workspace(ws_need, &ws);
  if (dist) rc = comm_agree(rc);         // a rank that cannot get its workspace takes the team out with it (-993)
  if (rc) return rc;
  Arena a(ws, g_ctx.ws_bytes);
  double *wA = a.get<double>((size_t)ld * ld);
  double *wB = a.get<double>((size_t)ld * ld);
  double *wZ = a.get<double>((size_t)ld * ld);
  double *wV = a.get<double>((size_t)ld * ld);
  double *dInv = a.get<double>((size_t)nblk * kDiagNB * kDiagNB);
  double *twork = a.get<double>((size_t)128 * ld);
  char *work = a.get<char>(scratch);
  double *dd = a.get<double>(ld), *de = a.get<double>(ld), *dt = a.get<double>(ld), *dwv = a.get<double>(ld);
  double *sscr = (problem == 1) ? a.get<double>(sygst_dbl) : nullptr;
  char *pwork = potrf_wb ? a.get<char>(potrf_wb) : nullptr;
  double *wV2 = two_stage ? a.get<double>((size_t)ld * ld) : nullptr;
  double *wA0 = two_stage ? a.get<double>((size_t)ld * ld) : nullptr;
  char *work_sy2sb = two_stage ? a.get<char>(wb_sy2sb) : nullptr;
  char *work_sb2st = two_stage ? a.get<char>(wb_sb2st) : nullptr;
  char *q1prep = two_stage ? a.get<char>(wb_q1prep) : nullptr;
  double *dt1 = two_stage ? a.get<double>(ld) : nullptr;
  // where the tridiagonalisation keeps x, the panel and its partial sums (probed once per workspace)
  void *sytrd_work = two_stage ? (void *)work : choose_sytrd_scratch(n, ld, wA, work, dd, wb_sytrd);

  StageTimer tm;
  const bool timing = stage_seconds && n_stages > 0;
  if (timing) { rc = tm.init(); if (rc) return rc; }
  int evi = 0;
  auto mark = [&]() { if (timing) (void)hipEventRecord(tm.ev[evi++], s); };

  EK_HIP_CHECK(hipMemsetAsync(g_ctx.d_info, 0, 4 * sizeof(int), s));
  mark();                                                              // 0
  // stage-in: padded, zero-filled work copies.  With a staging pipeline B comes first and A is waited for only
  // after the Cholesky factorisation has been issued (it is still crossing PCIe meanwhile).
  EK_HIP_CHECK(hipMemsetAsync(wV, 0, (size_t)ld * ld * 8, s));
  EK_HIP_CHECK(hipMemsetAsync(dd, 0, 4 * al((size_t)ld * 8), s));
  double sigma = 1.0;
  auto stage_in_B = [&]() -> int {
    if (problem != 1) return 0;
    if (pipe) { const int e = pipe->wait_in(0); if (e) return e; }
    if (ld != n) EK_HIP_CHECK(hipMemsetAsync(wB, 0, (size_t)ld * ld * 8, s));
    copy_matrix(s, n, n, dB, ldb, wB, ld);
    return 0;
  };
  auto stage_in_A = [&]() -> int {
    if (pipe) { const int e = pipe->wait_in(1); if (e) return e; }
    if (ld != n) EK_HIP_CHECK(hipMemsetAsync(wA, 0, (size_t)ld * ld * 8, s));   // (the copy covers all of an unpadded array)
    copy_matrix(s, n, n, dA, lda, wA, ld);
    // Scale A into the safe range when its entries are extreme (as DSYEV / PDSYEV do before
    // DSYTRD): the Householder norms are plain sums of squares.  Eigenvalues scale back linearly.
    double *d_part = (double *)work;   // stage scratch, free until the reduction starts
    maxabs_lower(s, n, wA, ld, d_part);
    double part[256];
    EK_HIP_CHECK(hipMemcpyAsync(part, d_part, sizeof(part), hipMemcpyDeviceToHost, s));
    EK_HIP_CHECK(hipStreamSynchronize(s));
    double anrm = 0.0;
    for (double v : part) if (v > anrm) anrm = v;
    if (!(anrm <= 1.7e308)) return -4;   // NaN / Inf in A: illegal value, as XERBLA
    // the tridiagonalisation forms x^T A x of the unscaled column (|A|^3 n^2): keep cubes in range
    const double rmin = 1e-90, rmax = 1e90;
    if (anrm > 0.0 && anrm < rmin) sigma = rmin / anrm;
    else if (anrm > rmax) sigma = rmax / anrm;
    if (sigma != 1.0) scale_lower(s, n, sigma, wA, ld);
    return 0;
  };
  if (!pipe) { rc = stage_in_A(); if (rc) { tm.destroy(); return rc; } }
  rc = stage_in_B(); if (rc) { tm.destroy(); return rc; }
  mark();                                                              // 1
  g_comm.err = 0;
  if (problem == 1) {
    // right-looking sweep with one panel broadcast per strip: pays from three ranks on
    if (dist && g_comm.nranks >= dist_min_ranks()) {
      const PotrfMember me{wB, ld, dInv, g_ctx.d_info, pwork, g_comm.rank};
      potrf_lower_dist(s, n, 1, &me, team_exchange(0));
    } else if (potrf_rl) {
      potrf_lower_rl(s, g_ctx.stream2, n, wB, ld, dInv, g_ctx.d_info, pwork);
    } else {
      potrf_lower(s, n, wB, ld, dInv, g_ctx.d_info, twork);
    }
  }
  mark();                                                              // 2
  if (pipe) {
    if (problem == 1) {        // L is final: it leaves while the reduction runs
      copy_matrix(s, n, n, wB, ld, dB, ldb);
      pipe->push(true, dB, ldb, pipe->hB, pipe->ldhb, n, n, pipe->mark(s), 0);
    }
    rc = stage_in_A(); if (rc) { tm.destroy(); return rc; }
  }
  if (problem == 1) {
    // sharding the two solves costs 2 n^3 / P flops per rank against 1.0 - 1.57 n^3 replicated
    if (dist && g_comm.nranks >= dist_min_ranks()) {
      const SygstMember me{wA, ld, wB, ld, dInv, twork, sscr, g_comm.rank};
      sygst_lower_dist(s, n, 1, &me, team_exchange(0));
    } else {
      sygst_lower(s, n, wA, ld, wB, ld, dInv, twork, sscr);
    }
  }
  mark();                                                              // 3
  bool two_stage_done = false;
  double rescued_panels = 0.0;
  if (dist && !two_stage) {
    const SytrdMember me{wA, ld, dd, de, dt, wV, ld, sytrd_work, g_comm.rank};
    sytrd_lower_dist(s, n, 1, &me, team_exchange(0, n));
  } else if (two_stage) {
    // dense -> band -> tridiagonal.  The panel factorisation of the first stage is CholeskyQR2 with a
    // device-side check; a matrix it cannot handle (rank-deficient or very ill-conditioned panels,
    // e.g. an input that is already banded) takes the one-stage path from a copy instead.
    EK_HIP_CHECK(hipMemcpyAsync(wA0, wA, (size_t)ld * ld * 8, hipMemcpyDeviceToDevice, s));
    EK_HIP_CHECK(hipMemsetAsync(wV2, 0, (size_t)ld * ld * 8, s));
    EK_HIP_CHECK(hipMemsetAsync(dt1, 0, (size_t)ld * 8, s));
    if (dist) {
      // On a team the first stage is distributed over the 128-wide column strips (strip S on rank S mod P: where
      // the distributed reduction to standard form left the matrix, so nothing is gathered in front of it): per panel
      // one broadcast of [V | T | tau] and one all-reduce of Y (ek_sy2sb.hip).  Then ONE all-gather of the band
      // (65 n doubles); the bulge chasing and the D&C below its top merge run replicated, bit-identical on all ranks.
      const SytrdExchange x = team_exchange(0);
      const Sy2sbMember me{wA, ld, wV, ld, dt1, g_ctx.d_info + 2, work_sy2sb, g_comm.rank};
      sy2sb_lower_dist(s, n, 1, &me, x);
      double *ABs[1] = {sb2st_band(work_sb2st, n)};
      pack_band(s, n, wA, ld, ABs[0]);
      gather_band_strips(s, n, 1, g_comm.rank, ABs, x);
      sb2st_lower(s, n, wA, ld, dd, de, wV2, ld, g_ctx.d_info + 2, work_sb2st, /*band_packed=*/true);
    } else {
      sy2sb_lower(s, g_ctx.stream2, n, wA, ld, wV, ld, dt1, g_ctx.d_info + 2, work_sy2sb);
      sb2st_lower(s, n, wA, ld, dd, de, wV2, ld, g_ctx.d_info + 2, work_sb2st);
    }
    // (the bulge chasing does nothing when the first stage has raised its flag: the band is not valid then)
    int flag = 0;
    EK_HIP_CHECK(hipMemcpyAsync(&flag, g_ctx.d_info + 2, sizeof(int), hipMemcpyDeviceToHost, s));
    EK_HIP_CHECK(hipStreamSynchronize(s));
    rescued_panels = (double)(flag >> 8);      // panels of the first stage that took the Householder rescue
    flag &= 0xff;                              // the low byte says why the two-stage form gave up, if it did
    // a team decides together: a flag that only one rank has raised (an abandoned wait depends on timing, not on the
    // data) must not leave the ranks with eigenvectors of two different decompositions
    if (dist) { flag = comm_any(flag); if (flag < 0) return flag; }
    if (flag == 0) two_stage_done = true;
    else {
      EK_HIP_CHECK(hipMemcpyAsync(wA, wA0, (size_t)ld * ld * 8, hipMemcpyDeviceToDevice, s));
      EK_HIP_CHECK(hipMemsetAsync(wV, 0, (size_t)ld * ld * 8, s));
      EK_HIP_CHECK(hipMemsetAsync(dd, 0, 3 * al((size_t)ld * 8), s));
      if (dist) {     // (the copy holds the matrix in this rank's strips only: the one-stage form over the team)
        const SytrdMember me{wA, ld, dd, de, dt, wV, ld, sytrd_work, g_comm.rank};
        sytrd_lower_dist(s, n, 1, &me, team_exchange(0, n));
      } else {
        sytrd_lower(s, n, wA, ld, dd, de, dt, wV, ld, sytrd_work);
      }
    }
  } else {
    sytrd_lower(s, n, wA, ld, dd, de, dt, wV, ld, sytrd_work);
  }
  if (pipe) {                  // what the call leaves in A (reflectors / band) is final
    copy_matrix(s, n, n, wA, ld, dA, lda);
    pipe->push(true, dA, lda, pipe->hA, pipe->ldha, n, n, pipe->mark(s), 0);
  }
  mark();                                                              // 4
  // eigenvector columns wanted: the first n_vec, or this grid cell's share of them; the D&C
  // forms only those (columns 0..nc_loc-1 of wZ) and the two remaining stages treat the
  // columns of Z independently
  const StedcSelect pick{nc_loc, cell ? cell->nb : (n > 0 ? n : 1), cell ? cell->npcol : 1, cell ? cell->mycol : 0};
  stedc(s, n, dd, de, dwv, wZ, ld, work, g_ctx.d_info + 1, &pick, g_ctx.d_stats);
  mark();                                                              // 5
  double *zc = wZ;
  // with a staging pipeline the LAST stage (the recovery; the back-transformation of a standard problem) runs in column
  // slabs, each leaving for the host while the next is computed (columns of Z are independent there)
  const int zslab = (pipe && two_stage_done && nc_loc > pipe->z_slab) ? pipe->z_slab : nc_loc;
  auto z_out = [&](int c0, int nc) {
    copy_matrix(s, n, nc, wZ + (size_t)c0 * ld, ld, dZ + (size_t)c0 * ldz, ldz);
    pipe->push(true, dZ + (size_t)c0 * ldz, ldz, pipe->hZ + (size_t)c0 * pipe->ldhz, pipe->ldhz, n, nc, pipe->mark(s), 0);
  };
  if (two_stage_done) {
    sb2st_apply_q2(s, n, nc_loc, wV2, ld, zc, ld, g_ctx.d_info + 2, work_sb2st);
    // (the T factors of the block reflectors do not depend on Z; forming them on the second stream beside the
    // bulge chasing was measured: the skinny GEMMs take CUs and issue slots from the latency-bound pipeline,
    // which loses 11 ms to gain 6)
    ormtr_prepare(s, n, wV, ld, dt1, q1prep);
    if (pipe && problem == 0 && zslab < nc_loc) {
      for (int c0 = 0; c0 < nc_loc; c0 += zslab) {
        const int nc = (nc_loc - c0 < zslab) ? nc_loc - c0 : zslab;
        ormtr_apply(s, n, nc, wV, ld, q1prep, zc + (size_t)c0 * ld, ld, work, n_vec);
        z_out(c0, nc);
      }
    } else {
      ormtr_apply(s, n, nc_loc, wV, ld, q1prep, zc, ld, work, n_vec);
    }
  } else {
    ormtr_lower(s, n, nc_loc, wV, ld, dt, zc, ld, work, n_vec);
  }
  mark();                                                              // 6
  if (problem == 1) {
    if (pipe) {
      for (int c0 = 0; c0 < nc_loc; c0 += zslab) {
        const int nc = (nc_loc - c0 < zslab) ? nc_loc - c0 : zslab;
        trsm_llt(s, n, nc, wB, ld, dInv, zc + (size_t)c0 * ld, ld, twork);
        z_out(c0, nc);
      }
    } else {
      trsm_llt(s, n, nc_loc, wB, ld, dInv, zc, ld, twork);
    }
  } else if (pipe && !(two_stage_done && zslab < nc_loc)) {
    z_out(0, nc_loc);
  }
  mark();                                                              // 7
  // stage-out: eigenvalues, eigenvectors, and the in-place results the reference leaves
  // behind (L in B, reflectors in A)
  if (sigma != 1.0) scale_vector(s, n, 1.0 / sigma, dwv);
  EK_HIP_CHECK(hipMemcpyAsync(dw, dwv, (size_t)n * 8, hipMemcpyDeviceToDevice, s));
  if (!pipe) {
    if (cell) gather_block_cyclic(s, nr_loc, nc_loc, zc, ld, cell->nb, cell->nprow, cell->myrow, 1, 0, dZ, ldz);
    else copy_matrix(s, n, n_vec, wZ, ld, dZ, ldz);
    copy_matrix(s, n, n, wA, ld, dA, lda);
    if (problem == 1) copy_matrix(s, n, n, wB, ld, dB, ldb);
  }
  mark();                                                              // 8
  EK_HIP_CHECK(hipGetLastError());
  int info[4] = {0, 0, 0, 0};
  EK_HIP_CHECK(hipMemcpyAsync(info, g_ctx.d_info, sizeof(info), hipMemcpyDeviceToHost, s));
  EK_HIP_CHECK(hipMemcpyAsync(g_ctx.stats, g_ctx.d_stats, sizeof(g_ctx.stats), hipMemcpyDeviceToHost, s));
  EK_HIP_CHECK(hipStreamSynchronize(s));
  g_ctx.stats[1] = two_stage_done ? 1.0 : 0.0;
  g_ctx.stats[2] = rescued_panels;
  info[2] &= 0xff;
  if (timing) {
    float ms[8];
    for (int i = 0; i < 8; ++i) (void)hipEventElapsedTime(&ms[i], tm.ev[i], tm.ev[i + 1]);
    double st[EK_HIP_N_STAGES] = {0};
    st[EK_STAGE_COPY] = (ms[0] + ms[7]) * 1e-3;
    st[EK_STAGE_POTRF] = ms[1] * 1e-3; st[EK_STAGE_SYGST] = ms[2] * 1e-3;
    st[EK_STAGE_SYTRD] = ms[3] * 1e-3; st[EK_STAGE_GATHER] = 0.0;
    st[EK_STAGE_STEDC] = ms[4] * 1e-3; st[EK_STAGE_ORMTR] = ms[5] * 1e-3;
    st[EK_STAGE_TRTRS] = ms[6] * 1e-3;
    for (int i = 0; i < n_stages && i < EK_HIP_N_STAGES; ++i) stage_seconds[i] = st[i];
    tm.destroy();
  }
  if (dist && g_comm.err) {
    fprintf(stderr, "[ek_hip] RCCL all-reduce failed: %s\n", comm_error_string());
    return -996;
  }
  if (two_stage_done) {   // the pipelined back-transformation was abandoned (a bounded wait ran out): on a team, for all ranks
    int bad = info[2] != 0;
    if (dist) bad = comm_any(bad);
    if (bad) return bad < 0 ? bad : -992;
  }
  if (info[0] != 0) return info[0];          // Cholesky: leading minor not positive definite
  if (info[1] != 0) return 100000 + info[1];  // tridiagonal eigensolver did not converge
  return 0;
}

// Replicated host inputs (full A, B on every rank) -> this cell's block-cyclic piece of Z.
int replicated_host_locked(int problem, int n, int n_vec, double *A, int lda, double *B, int ldb, double *w,
                           double *Z_loc, int lldz, const GridCell &cell, double *stage_seconds,
                           int n_stages) {
  hipStream_t s = g_ctx.stream;
  const int nr_loc = numroc0(n, cell.nb, cell.myrow, cell.nprow);
  const int nc_loc = numroc0(n_vec, cell.nb, cell.mycol, cell.npcol);
  double *uA = nullptr, *uB = nullptr, *uZ = nullptr, *uw = nullptr;
  const size_t nn = (size_t)n * n * 8;
  const int ldzl = nr_loc > 1 ? nr_loc : 1;
  auto t0 = std::chrono::steady_clock::now();
  DevMem mem;
  int rc = mem.alloc(&uA, nn);
  if (!rc) rc = mem.alloc(&uZ, (size_t)ldzl * (nc_loc > 0 ? nc_loc : 1) * 8);
  if (!rc) rc = mem.alloc(&uw, (size_t)n * 8);
  if (!rc && problem == 1) rc = mem.alloc(&uB, nn);
  if (rc) return rc;
  rc = h2d_matrix(n, n, A, lda, uA, n, s);
  if (!rc && problem == 1) rc = h2d_matrix(n, n, B, ldb, uB, n, s);
  if (!rc) { hipError_t e = hipStreamSynchronize(s); if (e != hipSuccess) rc = -1000 - (int)e; }
  auto t1 = std::chrono::steady_clock::now();
  int info = rc;
  if (!rc) info = solve_device_locked(problem, n, n_vec, uA, n, uB, n, uw, uZ, ldzl, stage_seconds, n_stages, &cell);
  auto t2 = std::chrono::steady_clock::now();
  if (info > -1000) {
    int rc2 = 0;
    if (nr_loc > 0 && nc_loc > 0) rc2 = d2h_matrix(nr_loc, nc_loc, uZ, ldzl, Z_loc, lldz, s);
    if (!rc2) rc2 = d2h_matrix(n, n, uA, n, A, lda, s);
    if (!rc2 && problem == 1) rc2 = d2h_matrix(n, n, uB, n, B, ldb, s);
    if (!rc2) { hipError_t e = hipMemcpyAsync(w, uw, (size_t)n * 8, hipMemcpyDeviceToHost, s); if (e != hipSuccess) rc2 = -1000 - (int)e; }
    if (!rc2) { hipError_t e = hipStreamSynchronize(s); if (e != hipSuccess) rc2 = -1000 - (int)e; }
    if (rc2 && info == 0) info = rc2;
  }
  auto t3 = std::chrono::steady_clock::now();
  if (stage_seconds && n_stages > EK_STAGE_COPY)
    stage_seconds[EK_STAGE_COPY] += std::chrono::duration<double>(t1 - t0).count() +
                                    std::chrono::duration<double>(t3 - t2).count();
  return info;
}

}  // namespace

extern "C" {

int ek_hip_solve_device(int problem, int n, int n_vec, double *dA, int lda, double *dB, int ldb,
                        double *dw, double *dZ, int ldz, double *stage_seconds, int n_stages) {
  if (problem != 0 && problem != 1) return -1;
  if (n < 0) return -2;
  if (n_vec < 0 || n_vec > n) return -3;
  if (n > 0 && !dA) return -4;
  if (lda < (n > 1 ? n : 1)) return -5;
  if (problem == 1 && n > 0 && !dB) return -6;
  if (problem == 1 && ldb < (n > 1 ? n : 1)) return -7;
  if (n > 0 && !dw) return -8;
  if (n > 0 && !dZ) return -9;
  if (ldz < (n > 1 ? n : 1)) return -10;
  int rc = ensure_init(); if (rc) return rc;
  if (n == 0) return 0;
  std::lock_guard<std::mutex> lk(g_mu);
  return solve_device_locked(problem, n, n_vec, dA, lda, dB, ldb, dw, dZ, ldz, stage_seconds, n_stages);
}

int ek_hip_set_allgatherv(ek_hip_allgatherv_fn fn, void *user) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_allgatherv = fn; g_allgatherv_user = user;
  return 0;
}

// Pure host code (no GPU needed): the exchange step of ek_hip_solve for distributed inputs.
int ek_hip_gather_matrix(int m, int n, const double *M_loc, const int desc[9], int nprow, int npcol,
                         int myrow, int mycol, double *M_full, int ldf) {
  if (m < 0) return -1;
  if (n < 0) return -2;
  if (m > 0 && n > 0 && !M_loc) return -3;
  if (nprow < 1) return -5;
  if (npcol < 1) return -6;
  if (myrow < 0 || myrow >= nprow) return -7;
  if (mycol < 0 || mycol >= npcol) return -8;
  if (!desc) return -4;
  if (desc[4] < 1) return -405;
  int rc = check_desc(desc, 4, m, n, numroc0(m, desc[4], myrow, nprow)); if (rc) return rc;
  if (m > 0 && n > 0 && !M_full) return -9;
  if (ldf < (m > 1 ? m : 1)) return -10;
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_allgatherv) return -998;
  if (m == 0 || n == 0) return 0;
  const GridCell cell{desc[4], nprow, npcol, myrow, mycol};
  return gather_full(m, n, M_loc, desc, cell, M_full, ldf);
}

int ek_hip_solve_device_grid(int problem, int n, int n_vec, double *dA, int lda, double *dB, int ldb,
                             double *dw, double *dZ_loc, int ldz_loc, int nb, int nprow, int npcol,
                             int myrow, int mycol, double *stage_seconds, int n_stages) {
  if (problem != 0 && problem != 1) return -1;
  if (n < 0) return -2;
  if (n_vec < 0 || n_vec > n) return -3;
  if (n > 0 && !dA) return -4;
  if (lda < (n > 1 ? n : 1)) return -5;
  if (problem == 1 && n > 0 && !dB) return -6;
  if (problem == 1 && ldb < (n > 1 ? n : 1)) return -7;
  if (n > 0 && !dw) return -8;
  if (n > 0 && !dZ_loc) return -9;
  if (nb < 1) return -11;
  if (nprow < 1) return -12;
  if (npcol < 1) return -13;
  if (myrow < 0 || myrow >= nprow) return -14;
  if (mycol < 0 || mycol >= npcol) return -15;
  const int nr_loc = numroc0(n, nb, myrow, nprow);
  if (ldz_loc < (nr_loc > 1 ? nr_loc : 1)) return -10;
  int rc = ensure_init(); if (rc) return rc;
  if (n == 0) return 0;
  std::lock_guard<std::mutex> lk(g_mu);
  const GridCell cell{nb, nprow, npcol, myrow, mycol};
  return solve_device_locked(problem, n, n_vec, dA, lda, dB, ldb, dw, dZ_loc, ldz_loc, stage_seconds,
                             n_stages, &cell);
}

int ek_hip_solve_replicated(int problem, int n, int n_vec, double *A, int lda, double *B, int ldb,
                            double *w, double *Z_loc, const int desc_Z[9], int nprow, int npcol,
                            int myrow, int mycol, double *stage_seconds, int n_stages) {
  if (problem != 0 && problem != 1) return -1;
  if (n < 0) return -2;
  if (n_vec < 0 || n_vec > n) return -3;
  if (n > 0 && !A) return -4;
  if (lda < (n > 1 ? n : 1)) return -5;
  if (problem == 1 && n > 0 && !B) return -6;
  if (problem == 1 && ldb < (n > 1 ? n : 1)) return -7;
  if (n > 0 && !w) return -8;
  if (n > 0 && !Z_loc) return -9;
  if (nprow < 1) return -11;
  if (npcol < 1) return -12;
  if (myrow < 0 || myrow >= nprow) return -13;
  if (mycol < 0 || mycol >= npcol) return -14;
  if (!desc_Z) return -10;
  if (desc_Z[4] < 1) return -(10 * 100 + 5);
  int rc = check_desc(desc_Z, 10, n, n, numroc0(n, desc_Z[4], myrow, nprow)); if (rc) return rc;
  rc = ensure_init(); if (rc) return rc;
  if (n == 0) return 0;
  std::lock_guard<std::mutex> lk(g_mu);
  const GridCell cell{desc_Z[4], nprow, npcol, myrow, mycol};
  return replicated_host_locked(problem, n, n_vec, A, lda, B, ldb, w, Z_loc, desc_Z[8], cell, stage_seconds,
                                n_stages);
}

int ek_hip_solve(int problem, int n, int n_vec, double *A_loc, const int desc_A[9], double *B_loc,
                 const int desc_B[9], double *w, double *Z_loc, const int desc_Z[9], int nprow,
                 int npcol, int myrow, int mycol, double *stage_seconds, int n_stages) {
  if (problem != 0 && problem != 1) return -1;
  if (n < 0) return -2;
  if (n_vec < 0 || n_vec > n) return -3;
  if (n > 0 && !A_loc) return -4;
  const bool cell_ok = nprow >= 1 && npcol >= 1 && myrow >= 0 && myrow < nprow;
  auto rows_of = [&](const int *d) {   // local row count the descriptor's lld must cover
    return (d && d[4] >= 1 && cell_ok) ? numroc0(n, d[4], myrow, nprow) : n;
  };
  int rc = check_desc(desc_A, 5, n, n, rows_of(desc_A)); if (rc) return rc;
  if (problem == 1) {
    if (n > 0 && !B_loc) return -6;
    rc = check_desc(desc_B, 7, n, n, rows_of(desc_B)); if (rc) return rc;
  }
  if (n > 0 && !w) return -8;
  if (n > 0 && !Z_loc) return -9;
  rc = check_desc(desc_Z, 10, n, n, rows_of(desc_Z)); if (rc) return rc;
  // grids other than 1x1 need the host's exchange hook (ek_hip_set_allgatherv)
  const bool have_exchange = g_allgatherv || (g_comm.on && nprow > 0 && npcol > 0 && g_comm.nranks == nprow * npcol);
  if (nprow != 1 && !(nprow > 1 && have_exchange)) return -11;
  if (npcol != 1 && !(npcol > 1 && have_exchange)) return -12;
  if (myrow < 0 || myrow >= nprow) return -13;
  if (mycol < 0 || mycol >= npcol) return -14;
  rc = ensure_init(); if (rc) return rc;
  if (n == 0) return 0;
  std::lock_guard<std::mutex> lk(g_mu);
  if (nprow * npcol > 1 && g_comm.on && g_comm.nranks == nprow * npcol) {
    // distributed inputs with a communicator attached: only the local pieces cross PCIe; the full
    // matrices are assembled in HBM by one all-gather per matrix (RCCL over xGMI, or the host hook
    // of a host communicator) and the pieces of the reflectors / of L are cut out on the device
    const GridCell cell{desc_Z[4], nprow, npcol, myrow, mycol};
    if (g_comm.rank != myrow * npcol + mycol) return -994;
    hipStream_t s = g_ctx.stream;
    const int P = nprow * npcol, me = g_comm.rank;
    const SytrdExchange x = team_exchange(0);
    auto t0 = std::chrono::steady_clock::now();
    DevMem mem;
    double *uA = nullptr, *uB = nullptr, *uZ = nullptr, *uw = nullptr, *pk = nullptr;
    const size_t nn = (size_t)n * n;
    const int nrz = numroc0(n, cell.nb, myrow, nprow), ncz = numroc0(n_vec, cell.nb, mycol, npcol);
    const int ldzl = nrz > 1 ? nrz : 1;
    rc = mem.alloc(&uA, nn * 8);
    if (!rc && problem == 1) rc = mem.alloc(&uB, nn * 8);
    if (!rc) rc = mem.alloc(&pk, nn * 8);
    if (!rc) rc = mem.alloc(&uZ, (size_t)ldzl * (ncz > 0 ? ncz : 1) * 8);
    if (!rc) rc = mem.alloc(&uw, (size_t)n * 8);
    rc = comm_agree(rc);                 // nobody enters the all-gathers below unless everybody can
    if (rc) return rc;
    g_comm.err = 0;
    auto assemble = [&](const double *M_loc, const int *desc, double *full) -> int {
      const int nb = desc[4];
      size_t offs[kMaxTeam], counts[kMaxTeam];
      size_t tot = 0;
      for (int r = 0; r < P; ++r) {
        counts[r] = (size_t)numroc0(n, nb, r / npcol, nprow) * numroc0(n, nb, r % npcol, npcol);
        offs[r] = tot; tot += counts[r];
      }
      const int nr = numroc0(n, nb, myrow, nprow), nc = numroc0(n, nb, mycol, npcol);
      if (nr > 0 && nc > 0) { int r2 = h2d_matrix(nr, nc, M_loc, desc[8], pk + offs[me], nr, s); if (r2) return r2; }
      double *bufs[1] = {pk};
      x.allgatherv(s, 1, me, bufs, offs, counts, P, x.user);
      for (int r = 0; r < P; ++r)
        scatter_block_cyclic(s, numroc0(n, nb, r / npcol, nprow), numroc0(n, nb, r % npcol, npcol), pk + offs[r],
                             numroc0(n, nb, r / npcol, nprow) > 1 ? numroc0(n, nb, r / npcol, nprow) : 1, nb, nprow,
                             r / npcol, npcol, r % npcol, full, n);
      return 0;
    };
    auto tg0 = std::chrono::steady_clock::now();
    int info = assemble(A_loc, desc_A, uA);
    if (!info && problem == 1) info = assemble(B_loc, desc_B, uB);
    if (!info) { hipError_t e = hipStreamSynchronize(s); if (e != hipSuccess) info = -1000 - (int)e; }
    if (!info && g_comm.err) info = -996;
    info = comm_agree(info);             // a staging failure on one rank ends the call on all of them
    const double tg = std::chrono::duration<double>(std::chrono::steady_clock::now() - tg0).count();
    auto t1 = std::chrono::steady_clock::now();
    if (!info) info = solve_device_locked(problem, n, n_vec, uA, n, uB, n, uw, uZ, ldzl, stage_seconds, n_stages, &cell);
    auto t2 = std::chrono::steady_clock::now();
    if (info > -1000) {
      int rc2 = 0;
      if (nrz > 0 && ncz > 0) rc2 = d2h_matrix(nrz, ncz, uZ, ldzl, Z_loc, desc_Z[8], s);
      auto cut = [&](const double *full, const int *desc, double *M_loc) -> int {
        const int nb = desc[4], nr = numroc0(n, nb, myrow, nprow), nc = numroc0(n, nb, mycol, npcol);
        if (nr <= 0 || nc <= 0) return 0;
        gather_block_cyclic(s, nr, nc, full, n, nb, nprow, myrow, npcol, mycol, pk, nr);
        return d2h_matrix(nr, nc, pk, nr, M_loc, desc[8], s);
      };
      if (!rc2) rc2 = cut(uA, desc_A, A_loc);
      if (!rc2) { hipError_t e = hipStreamSynchronize(s); if (e != hipSuccess) rc2 = -1000 - (int)e; }   // pk is reused
      if (!rc2 && problem == 1) rc2 = cut(uB, desc_B, B_loc);
      if (!rc2) { hipError_t e = hipMemcpyAsync(w, uw, (size_t)n * 8, hipMemcpyDeviceToHost, s); if (e != hipSuccess) rc2 = -1000 - (int)e; }
      if (!rc2) { hipError_t e = hipStreamSynchronize(s); if (e != hipSuccess) rc2 = -1000 - (int)e; }
      if (rc2 && info == 0) info = rc2;
    }
    auto t3 = std::chrono::steady_clock::now();
    if (stage_seconds && n_stages > EK_STAGE_COPY)
      stage_seconds[EK_STAGE_COPY] += std::chrono::duration<double>(t1 - t0).count() - tg +
                                      std::chrono::duration<double>(t3 - t2).count();
    if (stage_seconds && n_stages > EK_STAGE_GATHER) stage_seconds[EK_STAGE_GATHER] += tg;
    return info;
  }
  if (nprow * npcol > 1) {
    // distributed inputs: assemble the full matrices on every rank through the hook, then
    // proceed as in the replicated-input mode; A_loc / B_loc receive their pieces of the
    // reflectors / of L, as every rank of the reference ends up with
    const GridCell cell{desc_Z[4], nprow, npcol, myrow, mycol};
    double *Af = (double *)malloc((size_t)n * n * 8);
    double *Bf = problem == 1 ? (double *)malloc((size_t)n * n * 8) : nullptr;
    int info = (!Af || (problem == 1 && !Bf)) ? -1000 - (int)hipErrorOutOfMemory : 0;
    auto t0 = std::chrono::steady_clock::now();
    if (!info) info = gather_full(n, n, A_loc, desc_A, cell, Af, n);
    if (!info && problem == 1) info = gather_full(n, n, B_loc, desc_B, cell, Bf, n);
    const double tg = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (!info) {
      info = replicated_host_locked(problem, n, n_vec, Af, n, Bf, n, w, Z_loc, desc_Z[8], cell,
                                    stage_seconds, n_stages);
      if (info > -1000) {
        extract_local(n, n, Af, n, desc_A, cell, A_loc);
        if (problem == 1) extract_local(n, n, Bf, n, desc_B, cell, B_loc);
      }
      if (stage_seconds && n_stages > EK_STAGE_GATHER) stage_seconds[EK_STAGE_GATHER] += tg;
    }
    free(Af); free(Bf);
    return info;
  }
  hipStream_t s = g_ctx.stream;
  // user-side device images (exact n x n); freed before returning: the library keeps nothing
  double *uA = nullptr, *uB = nullptr, *uZ = nullptr, *uw = nullptr;
  const size_t nn = (size_t)n * n * 8;
  auto t0 = std::chrono::steady_clock::now();
  DevMem mem;
  rc = mem.alloc(&uA, nn);
  if (!rc) rc = mem.alloc(&uZ, nn);
  if (!rc) rc = mem.alloc(&uw, (size_t)n * 8);
  if (!rc && problem == 1) rc = mem.alloc(&uB, nn);
  if (rc) return rc;
  int pipe_min = 2048;           // EK_HIP_PIPE_MIN: order from which the host path stages through the pipeline (0: never)
  if (const char *e = getenv("EK_HIP_PIPE_MIN")) pipe_min = atoi(e);
  if (pipe_min > 0 && n >= pipe_min) {
    // staging pipeline: the copies overlap the stages (HostPipe): what remains exposed is B's way in, the rest of A's
    // behind the Cholesky factorisation, and the last slab of Z
    HostPipe pipe;
    pipe.hA = A_loc; pipe.ldha = desc_A[8]; pipe.hB = B_loc; pipe.ldhb = problem == 1 ? desc_B[8] : 0;
    pipe.hZ = Z_loc; pipe.ldhz = desc_Z[8];
    rc = pipe.start(g_ctx.device);
    if (rc) return rc;
    if (problem == 1) pipe.push(false, uB, n, B_loc, desc_B[8], n, n, nullptr, 0);
    pipe.push(false, uA, n, A_loc, desc_A[8], n, n, nullptr, 1);
    double st[EK_HIP_N_STAGES] = {0};
    int info = solve_device_locked(problem, n, n_vec, uA, n, uB, n, uw, uZ, n, st, EK_HIP_N_STAGES, nullptr, &pipe);
    const int rcp = pipe.finish();
    if (info == 0 && rcp) info = rcp;
    if (info > -1000) {
      hipError_t e = hipMemcpyAsync(w, uw, (size_t)n * 8, hipMemcpyDeviceToHost, s);
      if (e == hipSuccess) e = hipStreamSynchronize(s);
      if (e != hipSuccess && info == 0) info = -1000 - (int)e;
    }
    const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (stage_seconds) {
      double dev = 0.0;
      for (int i = 0; i < EK_HIP_N_STAGES; ++i) if (i != EK_STAGE_COPY) dev += st[i];
      st[EK_STAGE_COPY] = wall > dev ? wall - dev : 0.0;     // what the copies add to the stages: their exposed part
      for (int i = 0; i < n_stages && i < EK_HIP_N_STAGES; ++i) stage_seconds[i] = st[i];
    }
    return info;
  }
  rc = h2d_matrix(n, n, A_loc, desc_A[8], uA, n, s);
  if (!rc && problem == 1) rc = h2d_matrix(n, n, B_loc, desc_B[8], uB, n, s);
  if (!rc) { hipError_t e = hipStreamSynchronize(s); if (e != hipSuccess) rc = -1000 - (int)e; }
  auto t1 = std::chrono::steady_clock::now();
  int info = rc;
  if (!rc) info = solve_device_locked(problem, n, n_vec, uA, n, uB, n, uw, uZ, n, stage_seconds, n_stages);
  auto t2 = std::chrono::steady_clock::now();
  if (info >= 0 || info > -1000) {
    // results travel back even when info > 0 so the host can inspect them, as with ScaLAPACK
    int rc2 = d2h_matrix(n, n_vec, uZ, n, Z_loc, desc_Z[8], s);
    if (!rc2) rc2 = d2h_matrix(n, n, uA, n, A_loc, desc_A[8], s);
    if (!rc2 && problem == 1) rc2 = d2h_matrix(n, n, uB, n, B_loc, desc_B[8], s);
    if (!rc2) { hipError_t e = hipMemcpyAsync(w, uw, (size_t)n * 8, hipMemcpyDeviceToHost, s); if (e != hipSuccess) rc2 = -1000 - (int)e; }
    if (!rc2) { hipError_t e = hipStreamSynchronize(s); if (e != hipSuccess) rc2 = -1000 - (int)e; }
    if (rc2 && info == 0) info = rc2;
  }
  auto t3 = std::chrono::steady_clock::now();
  if (stage_seconds && n_stages > EK_STAGE_COPY)
    stage_seconds[EK_STAGE_COPY] += std::chrono::duration<double>(t1 - t0).count() +
                                    std::chrono::duration<double>(t3 - t2).count();
  return info;
}


// ---- acceptance checks on the GPU: device-resident and host-array forms ----
int ek_hip_residual_device(int problem, int n, int n_check, const double *dA, int lda, const double *dB,
                           int ldb, const double *dw, const double *dZ, int ldz, double *a_norm,
                           double *res_ave, double *res_max) {
  if (problem != 0 && problem != 1) return -1;
  if (n < 0) return -2;
  if (n_check < 0 || n_check > n) return -3;
  if (n > 0 && !dA) return -4;
  if (problem == 1 && n > 0 && !dB) return -6;
  int rc = ensure_init(); if (rc) return rc;
  if (n == 0 || n_check == 0) return 0;
  std::lock_guard<std::mutex> lk(g_mu);
  hipStream_t s = g_ctx.stream;
  void *ws;
  rc = workspace(verify_work_bytes(n, n_check) + 256, &ws); if (rc) return rc;
  double *d_res = (double *)((char *)ws + verify_work_bytes(n, n_check));
  residual_norms(s, n, n_check, dA, lda, problem ? dB : nullptr, ldb, dw, dZ, ldz, d_res, ws);
  EK_HIP_CHECK(hipGetLastError());
  double r[3];
  EK_HIP_CHECK(hipMemcpyAsync(r, d_res, sizeof(r), hipMemcpyDeviceToHost, s));
  EK_HIP_CHECK(hipStreamSynchronize(s));
  if (a_norm) *a_norm = r[2];
  if (res_ave) *res_ave = r[0] / r[2] / (double)n_check;      /* verifier.f90:198 */
  if (res_max) *res_max = r[1] / r[2];                        /* verifier.f90:199 */
  return 0;
}

int ek_hip_orthogonality_device(int problem, int n, int index1, int index2, const double *dB, int ldb,
                                const double *dZ, int ldz, double *orthogonality) {
  if (problem != 0 && problem != 1) return -1;
  if (n < 0) return -2;
  if (index1 < 1 || index1 > n) return -3;
  if (index2 < index1 || index2 > n) return -4;
  if (problem == 1 && !dB) return -5;
  int rc = ensure_init(); if (rc) return rc;
  std::lock_guard<std::mutex> lk(g_mu);
  hipStream_t s = g_ctx.stream;
  const int nc = index2 - index1 + 1;
  void *ws;
  rc = workspace(verify_work_bytes(n, nc) + 256, &ws); if (rc) return rc;
  double *d_res = (double *)((char *)ws + verify_work_bytes(n, nc));
  ek::orthogonality(s, n, index1 - 1, nc, problem ? dB : nullptr, ldb, dZ, ldz, d_res, ws);
  EK_HIP_CHECK(hipGetLastError());
  double r[3];
  EK_HIP_CHECK(hipMemcpyAsync(r, d_res, sizeof(r), hipMemcpyDeviceToHost, s));
  EK_HIP_CHECK(hipStreamSynchronize(s));
  if (orthogonality) *orthogonality = r[2];
  return 0;
}

int ek_hip_ipratios_device(int problem, int n, int n_vec, const double *dB, int ldb, const double *dZ,
                           int ldz, double *ipratios_host) {
  if (problem != 0 && problem != 1) return -1;
  if (n < 0) return -2;
  if (n_vec < 0 || n_vec > n) return -3;
  if (problem == 1 && n > 0 && !dB) return -4;
  int rc = ensure_init(); if (rc) return rc;
  if (n == 0 || n_vec == 0) return 0;
  std::lock_guard<std::mutex> lk(g_mu);
  hipStream_t s = g_ctx.stream;
  void *ws;
  rc = workspace(verify_work_bytes(n, n_vec) + al((size_t)n * 8), &ws); if (rc) return rc;
  double *d_ipr = (double *)((char *)ws + verify_work_bytes(n, n_vec));
  ek::ipratios(s, n, n_vec, problem ? dB : nullptr, ldb, dZ, ldz, d_ipr, ws);
  EK_HIP_CHECK(hipGetLastError());
  EK_HIP_CHECK(hipMemcpyAsync(ipratios_host, d_ipr, (size_t)n_vec * 8, hipMemcpyDeviceToHost, s));
  EK_HIP_CHECK(hipStreamSynchronize(s));
  return 0;
}

// Host-array form: what the Fortran host calls in place of eval_residual_norm (verifier.f90:207),
// eval_orthogonality (:333) and get_ipratios (distribute_matrix.f90:18). what: 0 residual
// (out[0..2] = A_norm, res_ave, res_max; uses n_check columns), 1 orthogonality (out[0]; columns
// index1..index2, 1-based), 2 IPR (out[0..n_vec-1]).
int ek_hip_check(int what, int problem, int n, int n_cols, int index1, int index2, const double *A_loc,
                 const int desc_A[9], const double *B_loc, const int desc_B[9], const double *w,
                 const double *Z_loc, const int desc_Z[9], double *out) {
  if (what < 0 || what > 2) return -1;
  if (problem != 0 && problem != 1) return -2;
  if (n < 0) return -3;
  if (n_cols < 0 || n_cols > n) return -4;
  int rc;
  if (what == 0) { if (!A_loc) return -7; rc = check_desc(desc_A, 8, n, n); if (rc) return rc; }
  if (problem == 1) { if (!B_loc) return -9; rc = check_desc(desc_B, 10, n, n); if (rc) return rc; }
  if (!Z_loc) return -12;
  rc = check_desc(desc_Z, 13, n, n); if (rc) return rc;
  if (!out) return -14;
  rc = ensure_init(); if (rc) return rc;
  if (n == 0) return 0;
  double *uA = nullptr, *uB = nullptr, *uZ = nullptr, *uw = nullptr;
  const size_t nn = (size_t)n * n * 8;
  DevMem mem;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    hipStream_t s = g_ctx.stream;
    rc = mem.alloc(&uZ, nn); if (rc) return rc;
    rc = h2d_matrix(n, n, Z_loc, desc_Z[8], uZ, n, s); if (rc) return rc;
    if (what == 0) {
      rc = mem.alloc(&uA, nn); if (rc) return rc;
      rc = mem.alloc(&uw, (size_t)n * 8); if (rc) return rc;
      rc = h2d_matrix(n, n, A_loc, desc_A[8], uA, n, s); if (rc) return rc;
      EK_HIP_CHECK(hipMemcpyAsync(uw, w, (size_t)n * 8, hipMemcpyHostToDevice, s));
    }
    if (problem == 1) {
      rc = mem.alloc(&uB, nn); if (rc) return rc;
      rc = h2d_matrix(n, n, B_loc, desc_B[8], uB, n, s); if (rc) return rc;
    }
    EK_HIP_CHECK(hipStreamSynchronize(s));
  }
  if (what == 0) rc = ek_hip_residual_device(problem, n, n_cols, uA, n, uB, n, uw, uZ, n, &out[0], &out[1], &out[2]);
  else if (what == 1) rc = ek_hip_orthogonality_device(problem, n, index1, index2, uB, n, uZ, n, &out[0]);
  else rc = ek_hip_ipratios_device(problem, n, n_cols, uB, n, uZ, n, out);
  return rc;
}

}  // extern "C"
