// ek_solve.hip -- the whole path behind the reference's solver arms (solver_main.f90:52-99): the driver on
// device-resident data (solve_device_locked), the staging pipeline of the host path, the replicated- and
// distributed-input forms on process grids, and the C entries ek_hip_solve* of include/ek_hip.h.
#include "ek_api_internal.h"

#include <condition_variable>
#include <sched.h>
#include <deque>
#include <memory>
#include <new>
#include <thread>

namespace ek {
namespace api {

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


}  // namespace api
}  // namespace ek

using namespace ek;
using namespace ek::api;

namespace {

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


// Staging pipeline of ek_hip_solve on a 1 x 1 grid (host arrays in, host arrays out: solver_main.f90:64-65).  The
// copies run on worker threads with their own non-blocking streams while the main thread issues the stages:
//   in : B, then A (B is needed first: the Cholesky factorisation runs while A is still on its way);
//   out: L as soon as it is final (it leaves during the reduction), the reflectors / band of A after the
//        tridiagonalisation, Z in column slabs as the last stage finishes them, w last.
// The caller's arrays are pageable and handed to the runtime as they are: on every box met since round 4 it moves them at
// link rate (57 GB/s in).  Round 4's ring of pinned bounce buffers was measured slower under every runtime met (37 - 51
// GB/s in, ~1 GB/s per worker out beside running kernels; profiles/r04_host_path.txt) and is gone (round 5).  The number
// of workers on the way in follows the cores the process may run on, the way out has two (HostPipe::start says why).
// what the last staging pipeline did (ek_hip_debug_last_pipe_stats): [0] bytes in, [1] span of the input transfers (s),
// [2] busy seconds of the input workers, [3..5] the same on the way out, [6] seconds the main thread waited for inputs,
// [7] for the drain at the end, [8] workers per direction, [9] 0 (round 4: directions through a pinned ring), [10] seconds from the
// start of the pipeline to its end, [11] seconds before the first input job started
double g_pipe_stats[12] = {0};
// the pipeline's streams live as long as the process (creating its 14 streams took a call ~25 ms)
struct PipeStreams {
  hipStream_t cs[2 * 8] = {};
  int made = 0;
  int ensure(int n) {
    for (; made < n; ++made) EK_HIP_CHECK(hipStreamCreateWithFlags(&cs[made], hipStreamNonBlocking));
    return 0;
  }
  void release() {          // ek_hip_finalize: they come back with the next call that needs them
    for (int i = 0; i < made; ++i) { (void)hipStreamDestroy(cs[i]); cs[i] = nullptr; }
    made = 0;
  }
};
PipeStreams g_pipe_streams;

int usable_cores() {
  cpu_set_t set;
  CPU_ZERO(&set);
  if (sched_getaffinity(0, sizeof(set), &set) == 0) { const int c = CPU_COUNT(&set); if (c > 0) return c; }
  const unsigned h = std::thread::hardware_concurrency();
  return h > 0 ? (int)h : 1;
}

struct HostPipe {
  // tri: a square diagonal block on the way out of which only the lower triangle may reach the caller's array (m == n)
  struct Job { double *dev; int ldd; double *host; int ldh; int m, n; hipEvent_t after; int tag; bool to_host; bool tri = false; };
  static constexpr int kMaxThreads = 8;
  static constexpr int kTri = 512;           // columns of a diagonal block on the way out (2 MiB of scratch per worker)
  int kThreads = 2;                          // per direction (EK_HIP_PIPE_THREADS; default by the cores the process has)
  int kOutThreads = 2;                       // of them on the way out (see start())
  bool lower_only = true;                    // EK_HIP_PIPE_LOWER=0: whole matrices both ways, as round 3
  std::mutex mu;
  std::condition_variable cv;
  std::deque<Job> in_q, out_q;
  int pending_in[2] = {0, 0};                // tag 0 = B, 1 = A: pieces not yet in HBM
  int pending_out = 0;
  bool closing = false;
  int err = 0;
  std::vector<std::thread> th;
  hipStream_t cs[2 * kMaxThreads] = {};
  int device = 0;
  int z_slab = 2048;
  // EK_HIP_PIPE_TRACE=1: what every copy job and every wait of the main thread took (stderr, at the end of the call)
  bool trace = false;
  std::chrono::steady_clock::time_point t_origin;
  struct TraceRec { double t0, t1; double bytes; int kind; };      // kind 0 = in, 1 = out, 2 = wait_in, 3 = finish
  std::vector<TraceRec> trace_log;
  double now() const { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_origin).count(); }
  void collect_stats() {
    double busy[2] = {0, 0}, bytes[2] = {0, 0}, first[2] = {1e30, 1e30}, last[2] = {0, 0}, win = 0, wdrain = 0;
    for (const auto &r : trace_log) {
      if (r.kind < 2) {
        busy[r.kind] += r.t1 - r.t0; bytes[r.kind] += r.bytes;
        if (r.t0 < first[r.kind]) first[r.kind] = r.t0;
        if (r.t1 > last[r.kind]) last[r.kind] = r.t1;
      } else if (r.kind == 2) win += r.t1 - r.t0;
      else wdrain += r.t1 - r.t0;
    }
    for (int k = 0; k < 2; ++k) {
      g_pipe_stats[3 * k] = bytes[k]; g_pipe_stats[3 * k + 1] = bytes[k] > 0 ? last[k] - first[k] : 0.0; g_pipe_stats[3 * k + 2] = busy[k];
    }
    g_pipe_stats[6] = win; g_pipe_stats[7] = wdrain; g_pipe_stats[8] = 100.0 * kThreads + kOutThreads; g_pipe_stats[9] = 0;
    g_pipe_stats[10] = now(); g_pipe_stats[11] = bytes[0] > 0 ? first[0] : 0.0;
  }
  void report() {
    collect_stats();
    if (!trace) return;
    double busy[2] = {0, 0}, bytes[2] = {0, 0}, first[2] = {1e30, 1e30}, last[2] = {0, 0};
    for (const auto &r : trace_log) {
      if (r.kind < 2) {
        busy[r.kind] += r.t1 - r.t0; bytes[r.kind] += r.bytes;
        if (r.t0 < first[r.kind]) first[r.kind] = r.t0;
        if (r.t1 > last[r.kind]) last[r.kind] = r.t1;
      } else {
        fprintf(stderr, "[pipe] main thread waited %.4f s (%s) at t = %.4f\n", r.t1 - r.t0, r.kind == 2 ? "input" : "drain", r.t0);
      }
    }
    if (getenv("EK_HIP_PIPE_TRACE") && atoi(getenv("EK_HIP_PIPE_TRACE")) >= 2)       // every job
      for (const auto &r : trace_log)
        if (r.kind < 2) fprintf(stderr, "[pipe] job %s %8.1f MB  t = %.4f .. %.4f  (%.1f GB/s)\n", r.kind ? "out" : "in ", r.bytes / 1e6, r.t0, r.t1,
                                r.bytes / 1e9 / (r.t1 - r.t0 > 1e-9 ? r.t1 - r.t0 : 1e-9));
    for (int k = 0; k < 2; ++k)
      if (bytes[k] > 0)
        fprintf(stderr, "[pipe] %s: %.2f GB between t = %.4f and %.4f s (%.1f GB/s over the span), %d threads busy %.3f s in all (%.1f GB/s per busy thread), %s\n",
                k ? "out" : "in", bytes[k] / 1e9, first[k], last[k], bytes[k] / 1e9 / (last[k] - first[k]), k ? kOutThreads : kThreads, busy[k],
                bytes[k] / 1e9 / busy[k], "pageable");
  }

  int start(int dev) {
    device = dev;
    t_origin = std::chrono::steady_clock::now();
    { static int tr = -1; if (tr < 0) { const char *e = getenv("EK_HIP_PIPE_TRACE"); tr = (e && atoi(e) != 0) ? 1 : 0; } trace = tr != 0; }
    static int env_threads = -2;
    { static int lo = -1; if (lo < 0) { const char *e = getenv("EK_HIP_PIPE_LOWER"); lo = (e && atoi(e) == 0) ? 0 : 1; } lower_only = lo != 0; }
    const int cores = usable_cores();
    kThreads = cores >= 16 ? 6 : cores >= 8 ? 4 : 2;      // (both directions are rarely busy at once)
    if (env_threads >= 1 && env_threads <= kMaxThreads) kThreads = env_threads;
    { const int rc = g_pipe_streams.ensure(2 * kThreads); if (rc) return rc; }
    for (int i = 0; i < 2 * kThreads; ++i) cs[i] = g_pipe_streams.cs[i];
    // The way out hands pageable memory to the runtime: two threads saturate the link (30 GB/s
    // each), and MORE than two collapse under the HIP runtime PyTorch bundles (2.10: 1.7 GB/s per thread with three, 0.9
    // with six, beside running kernels; the system's runtime does 10 - 20 with any number) -- which is the runtime that
    // serves a process that imported torch first.  EK_HIP_PIPE_THREADS_OUT overrides.
    static int env_out = -2;
    if (env_out == -2) { const char *e = getenv("EK_HIP_PIPE_THREADS_OUT"); env_out = e ? atoi(e) : -1; }
    kOutThreads = kThreads < 2 ? kThreads : 2;
    if (env_out >= 1 && env_out <= kThreads) kOutThreads = env_out;
    for (int i = 0; i < kThreads + kOutThreads; ++i) th.emplace_back([this, i]() { run(i < kThreads, cs[i]); });
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
      const double tj0 = now();
      if (e == hipSuccess && j.m > 0 && j.n > 0 && j.tri) {
        // the block crosses into a scratch of this thread; the caller's array receives the entries on and below the
        // diagonal only (its strictly upper triangle is neither read nor written: PDPOTRF / PDSYTRD with uplo = 'L')
        std::vector<double> tmp((size_t)j.n * j.n);
        e = hipMemcpy2DAsync(tmp.data(), (size_t)j.n * 8, j.dev, (size_t)j.ldd * 8, (size_t)j.n * 8, j.n, hipMemcpyDeviceToHost, c);
        if (e == hipSuccess) e = hipStreamSynchronize(c);
        if (e == hipSuccess)
          for (int cc = 0; cc < j.n; ++cc)
            memcpy(j.host + (size_t)cc * j.ldh + cc, tmp.data() + (size_t)cc * j.n + cc, (size_t)(j.n - cc) * 8);
      } else if (e == hipSuccess && j.m > 0 && j.n > 0) {
        if (j.to_host)
          e = hipMemcpy2DAsync(j.host, (size_t)j.ldh * 8, j.dev, (size_t)j.ldd * 8, (size_t)j.m * 8, j.n, hipMemcpyDeviceToHost, c);
        else
          e = hipMemcpy2DAsync(j.dev, (size_t)j.ldd * 8, j.host, (size_t)j.ldh * 8, (size_t)j.m * 8, j.n, hipMemcpyHostToDevice, c);
        if (e == hipSuccess) e = hipStreamSynchronize(c);
      }
      {
        std::lock_guard<std::mutex> lk(mu);
        if (e != hipSuccess && !err) err = -1000 - (int)e;
        trace_log.push_back(TraceRec{tj0, now(), (double)j.m * j.n * 8.0, input ? 0 : 1});
        if (input) --pending_in[j.tag]; else --pending_out;
      }
      cv.notify_all();
    }
  }
  // host array (m x n, ldh) <-> device image (ldd), cut into column pieces for the threads (in column order: the
  // first columns of a matrix arrive first)
  // lower: a square matrix of which only the lower triangle is referenced (A, B on the way in; what the call leaves in A,
  // L on the way out: uplo = 'L' throughout the reference, generalized_to_standard.f90:24,37, solver_scalapack_all.f90:59):
  // a piece then carries the rows from its first column down -- half the bytes, the upper triangle of the destination is
  // not touched (as PDPOTRF / PDSYTRD leave it) -- and the pieces are cut to equal areas
  void push(bool to_host, double *dev, int ldd, double *host, int ldh, int m, int n, hipEvent_t after, int tag,
            bool lower = false) {
    const int pieces = (n >= 256) ? 4 * kThreads : 1;
    lower = lower && lower_only && m == n;
    {
      std::lock_guard<std::mutex> lk(mu);
      int c0 = 0;
      for (int p = 0; p < pieces; ++p) {
        int c1 = (int)((long long)n * (p + 1) / pieces);
        if (lower) c1 = (p + 1 == pieces) ? n : (int)((double)n * (1.0 - sqrt(1.0 - (double)(p + 1) / pieces)));
        if (c1 < c0) c1 = c0;
        if (lower && to_host) {
          // on the way out a piece is cut into chunks of at most kTri columns: the rows below a chunk's diagonal block as
          // one 2-D copy, the diagonal block as a `tri` job -- nothing above the diagonal of the caller's array is written
          for (int a = c0; a < c1; a += kTri) {
            const int b = (c1 - a < kTri) ? c1 : a + kTri;
            Job t{dev + (size_t)a * ldd + a, ldd, host + (size_t)a * ldh + a, ldh, b - a, b - a, after, tag, true, true};
            out_q.push_back(t); ++pending_out;
            if (b < m) {
              Job r{dev + (size_t)a * ldd + b, ldd, host + (size_t)a * ldh + b, ldh, m - b, b - a, after, tag, true};
              out_q.push_back(r); ++pending_out;
            }
          }
          c0 = c1;
          continue;
        }
        const int r0 = lower ? (c0 & ~1) : 0;                          // (even: 16-byte aligned rows)
        Job j{dev + (size_t)c0 * ldd + r0, ldd, host + (size_t)c0 * ldh + r0, ldh, m - r0, c1 - c0, after, tag, to_host};
        if (to_host) { out_q.push_back(j); ++pending_out; } else { in_q.push_back(j); ++pending_in[tag]; }
        c0 = c1;
      }
    }
    cv.notify_all();
  }
  int wait_in(int tag) {
    const double t0 = now();
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [&]() { return pending_in[tag] == 0; });
    trace_log.push_back(TraceRec{t0, now(), 0.0, 2});
    return err;
  }
  int finish() {                              // all copies done; threads joined; streams released
    {
      const double t0 = now();
      std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, [&]() { return pending_out == 0 && pending_in[0] == 0 && pending_in[1] == 0; });
      closing = true;
      trace_log.push_back(TraceRec{t0, now(), 0.0, 3});
    }
    cv.notify_all();
    for (auto &t : th) t.join();
    th.clear();
    for (auto &c : cs) c = nullptr;              // (the streams belong to the process-wide pool)
    for (auto &e : evs) (void)hipEventDestroy(e);
    evs.clear();
    report();
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

// Workspace of one whole-path call, laid out by LIFETIME (round 4).  Until round 3 every work array had its own place
// for the whole call (9.5 matrices + scratch at N = 16384, whatever the team size).  What a stage needs lives in one of
//   persistent : L (wB) and the first stage's reflectors (wV) -- operators every rank applies in full to its own
//                eigenvector columns at the end --, the eigenvector columns this call forms (wZ: n x n on one GPU, a grid
//                cell's share n x n/P on a team), the small vectors, the band and the bulge chasing's mail;
//   X0 (one matrix): the matrix A from the stage-in to the end of dense -> band, THEN the bulge chasing's reflectors V2
//                (what the call leaves in A goes back to the caller in between);
//   X1         : the scratch of the reduction to standard form and of the Cholesky factorisation, THEN the divide &
//                conquer's bases (compact for a cell's share: 1.5 n^2 + n n/P; 2 n^2 for the full spectrum on one GPU),
//                THEN the compact-WY records of Q2 (1.55 n^2, made from V2 after the D&C), THEN the scratch of Q1.
// The copy of the reduced matrix for a fall-back to the one-stage reduction is gone: a bulge chasing that abandons a
// bounded wait is repeated from the band it started from (a few MB) instead.
int g_debug_fail_chase = 0;     // test aid (ek_hip_debug_fail_next_chase): pretend the next k bulge chasings abandoned a wait
int g_debug_dc_team = 0;        // rehearsal aid (ek_hip_debug_stedc_team): a grid cell WITHOUT a communicator plays every rank of a
                                // team of that many in the divide & conquer's sharded heights (one GPU: tests, tools/team_timing.py)

struct PathPlan {
  int ld, nblk, zcols;
  bool two_stage, potrf_rl, compact_dc;
  size_t mat, zmat, x0, x1, sygst_dbl, potrf_wb, wb_sytrd, wb_stedc, wb_ormtr, wb_sy2sb, wb_sb2st, wb_rec, wb_q1prep,
         trsm_work, inv256, total;
};
PathPlan plan_path(int problem, int n, int n_vec, int nc_loc, int nranks_dist /* 0: not distributed */,
                   size_t exch_bytes = 0 /* X1's last life: the eigenvector pieces on their way between the cells of a process column */) {
  PathPlan p{};
  const bool dist = nranks_dist > 0;
  p.ld = pad_ld(n); p.nblk = ceil_div(n, kDiagNB);
  const int ts_min = two_stage_min();
  p.two_stage = ts_min > 0 && n >= ts_min && n >= 3;
  p.mat = al((size_t)p.ld * p.ld * 8);
  // the eigenvector work array holds the columns this call forms only (a grid cell's share, or the first n_vec of a
  // *_select arm) where the D&C can keep its bases compact; else it is n x n and doubles as the D&C's scratch
  p.compact_dc = stedc_compact(n, nc_loc);
  p.zcols = p.compact_dc ? round_up(nc_loc > 0 ? nc_loc : 1, 128) : p.ld;
  p.zmat = al((size_t)p.ld * p.zcols * 8);
  p.wb_sytrd = (p.two_stage) ? 0 : (dist ? sytrd_dist_work_bytes(n, nranks_dist) : sytrd_work_bytes(n));
  p.wb_stedc = stedc_work_bytes(n, nc_loc);
  p.wb_ormtr = ormtr_work_bytes(n, nc_loc, n_vec);
  p.trsm_work = al((size_t)256 * p.ld * 8);       // (a leaf of the solves writes its m x 256 result here before it goes back)
  p.inv256 = (problem == 1) ? al((size_t)(n / 256 > 0 ? n / 256 : 1) * 256 * 256 * 8) : 0;   // 256-block inverses of L
  p.sygst_dbl = (problem == 1) ? sygst_scratch_doubles(n) : 0;
  if (problem == 1 && dist) {
    const size_t dd = sygst_dist_scratch_doubles(n, p.ld, nranks_dist);
    if (dd > p.sygst_dbl) p.sygst_dbl = dd;
  }
  // right-looking Cholesky with look-ahead from this order on (below it the recursion is as fast)
  p.potrf_rl = problem == 1 && n >= kPotrfRlMin;
  p.potrf_wb = (problem == 1 && dist) ? al(potrf_dist_work_bytes(n, p.ld, nranks_dist)) : 0;
  if (p.potrf_rl && al(potrf_rl_work_bytes(n, p.ld)) > p.potrf_wb) p.potrf_wb = al(potrf_rl_work_bytes(n, p.ld));
  p.wb_sy2sb = p.two_stage ? al(dist ? sy2sb_dist_work_bytes(n, nranks_dist) : sy2sb_work_bytes(n)) : 0;
  p.wb_sb2st = p.two_stage ? al(sb2st_work_bytes(n, /*with_records=*/false)) : 0;
  p.wb_rec = p.two_stage ? al(sb2st_record_bytes(n)) : 0;
  p.wb_q1prep = p.two_stage ? al(ormtr_prep_bytes(n)) : 0;
  p.x0 = p.mat;
  const size_t phase_a = al(p.sygst_dbl * 8) + p.potrf_wb + al(p.wb_sytrd) + 512;   // [reduction | Cholesky | one-stage scratch]
  p.x1 = phase_a;
  if (al(p.wb_stedc) > p.x1) p.x1 = al(p.wb_stedc);
  if (p.wb_rec > p.x1) p.x1 = p.wb_rec;
  if (al(p.wb_ormtr) > p.x1) p.x1 = al(p.wb_ormtr);
  if (al(exch_bytes) > p.x1) p.x1 = al(exch_bytes);
  p.total = 2 * p.mat + p.zmat + p.x0 + p.x1 + al((size_t)p.nblk * kDiagNB * kDiagNB * 8) + p.trsm_work +
            5 * al((size_t)p.ld * 8) + p.wb_sy2sb + p.wb_sb2st + p.wb_q1prep + p.inv256 + 4096;
  return p;
}

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
  const int nc_out = cell ? numroc0(n_vec, cell->nb, cell->mycol, cell->npcol) : n_vec;    // columns of the piece of Z this call returns
  const int nr_loc = cell ? numroc0(n, cell->nb, cell->myrow, cell->nprow) : n;
  // A communicator attached by the host (ek_hip_comm_init) whose size is the grid's: the Cholesky factorisation, the
  // reduction and the dense -> band stage of the tridiagonalisation are distributed over the ranks (1 x P team, per panel
  // one broadcast and one all-reduce; below the two-stage orders the one-stage form with its per-column exchange); the
  // other stages are as in the replicated-input mode.
  const bool dist = cell && g_comm.on && g_comm.nranks == cell->nprow * cell->npcol;
  if (dist && g_comm.rank != cell->myrow * cell->npcol + cell->mycol) return -994;
  // On a grid with more than one process ROW the cells of a process column would all form the same eigenvector columns
  // (each keeping its rows): with a communicator they split them instead -- cell (i, j) forms the blocks l of its process
  // column's share with l mod nprow = i, which is the block-cyclic share of world rank i * npcol + j among all P ranks
  // -- and exchange row pieces pairwise at the end (team_sendrecv): 1 / P of the back-transformations and of the recovery
  // on every rank whatever the grid's shape, as PDORMTR / PDTRTRS have it on the reference's 2 x 4 grid
  // (solver_scalapack_all.f90:115, generalized_to_standard.f90:103, processes.f90:56-65).
  const bool split_rows = dist && cell->nprow > 1;
  const int nc_loc = split_rows ? numroc0(n_vec, cell->nb, g_comm.rank, g_comm.nranks) : nc_out;   // columns this call FORMS
  size_t exch_bytes = 0;
  if (split_rows) exch_bytes = ((size_t)n * (nc_loc > 0 ? nc_loc : 1) + (size_t)(nr_loc > 0 ? nr_loc : 1) * (nc_out > 0 ? nc_out : 1)) * 8 + 4096;
  const PathPlan pl = plan_path(problem, n, n_vec, nc_loc, dist ? g_comm.nranks : 0, exch_bytes);
  const int ld = pl.ld, nblk = pl.nblk, zcols = pl.zcols;
  const bool two_stage = pl.two_stage, potrf_rl = pl.potrf_rl;
  const size_t wb_sytrd = pl.wb_sytrd;
  int rc = 0;
  void *ws;
  g_comm.err = 0;                        // (sticky from here to the end of the call: the votes below keep it)
  rc = workspace(pl.total, &ws);
  if (dist) rc = comm_agree(rc);         // a rank that cannot get its workspace takes the team out with it (-993)
  if (rc) return rc;
  Arena a(ws, g_ctx.ws_bytes);
  // A caller's device array IS the internal work array when it already has the internal layout (order a multiple of 128,
  // leading dimension = order, 256-byte aligned: the bench's and the host path's arrays at the BASELINE orders): the
  // stage-in / stage-out copies of A, B and Z -- five passes over 2 GiB at N = 16384, 4.4 ms of a 0.86 s solve -- are
  // then not made (EK_HIP_ALIAS=0: always copy).  The plan above keeps its sizes: an upper bound.
  const char *alias_e = getenv("EK_HIP_ALIAS");          // (read per call: the tests switch it)
  const int alias_env = (alias_e && atoi(alias_e) == 0) ? 0 : 1;
  auto alias_ok = [&](const double *p, int ldu) {
    return alias_env && p && n % 128 == 0 && ldu == ld && (((size_t)p) & 255) == 0;
  };
  const bool aliasA = alias_ok(dA, lda), aliasB = problem == 1 && alias_ok(dB, ldb);
  const bool aliasZ = !cell && nc_loc == n && zcols == ld && alias_ok(dZ, ldz);
  double *wB = aliasB ? dB : a.get<double>((size_t)ld * ld);
  double *wV = a.get<double>((size_t)ld * ld);
  double *wZ = aliasZ ? dZ : a.get<double>((size_t)ld * zcols);
  double *x0 = a.get<double>((size_t)ld * ld);
  char *x1 = a.get<char>(pl.x1);
  double *dInv = a.get<double>((size_t)nblk * kDiagNB * kDiagNB);
  double *twork = a.get<double>((size_t)256 * ld);
  double *dd = a.get<double>(ld), *de = a.get<double>(ld), *dt = a.get<double>(ld), *dwv = a.get<double>(ld);
  double *dt1 = a.get<double>(ld);
  char *work_sy2sb = two_stage ? a.get<char>(pl.wb_sy2sb) : nullptr;
  char *work_sb2st = two_stage ? a.get<char>(pl.wb_sb2st) : nullptr;
  char *q1prep = two_stage ? a.get<char>(pl.wb_q1prep) : nullptr;
  double *inv256 = pl.inv256 ? (double *)a.get<char>(pl.inv256) : nullptr;
  struct Inv256Guard { ~Inv256Guard() { trsm_register_inv256(nullptr, nullptr, 0); } } inv256_guard;   // (whichever way the call ends)
  double *wA = aliasA ? dA : x0;         // X0, first life: the matrix (unless the caller's array serves)
  double *wV2 = x0;                      // X0, second life: the reflectors of the bulge chasing
  double *sscr = (problem == 1) ? (double *)x1 : nullptr;                       // X1, phase A
  char *pwork = pl.potrf_wb ? x1 + al(pl.sygst_dbl * 8) : nullptr;
  char *work = x1;                       // X1 as the scratch of the D&C / of the back-transformation
  double *q2rec = (double *)x1;          // X1 as the records of Q2
  // where the one-stage tridiagonalisation keeps x, the panel and its partial sums (probed once per workspace)
  char *sytrd_arena = x1 + al(pl.sygst_dbl * 8) + pl.potrf_wb + 256;
  // (the probe overwrites the matrix it is given: X0, never the caller's array)
  void *sytrd_work = two_stage ? nullptr : choose_sytrd_scratch(n, ld, x0, sytrd_arena, dd, wb_sytrd);

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
  bool band_input = false;     // a standard problem whose matrix is already a band of half width 64: no first stage, Q1 = I
  auto stage_in_B = [&]() -> int {
    if (problem != 1) return 0;
    if (pipe) { const int e = pipe->wait_in(0); if (e) return e; }
    if (aliasB) return 0;
    if (ld != n) EK_HIP_CHECK(hipMemsetAsync(wB, 0, (size_t)ld * ld * 8, s));
    copy_matrix(s, n, n, dB, ldb, wB, ld);
    return 0;
  };
  auto stage_in_A = [&]() -> int {
    if (pipe) { const int e = pipe->wait_in(1); if (e) return e; }
    if (!aliasA) {
      if (ld != n) EK_HIP_CHECK(hipMemsetAsync(wA, 0, (size_t)ld * ld * 8, s));   // (the copy covers all of an unpadded array)
      copy_matrix(s, n, n, dA, lda, wA, ld);
    }
    // Scale A into the safe range when its entries are extreme (as DSYEV / PDSYEV do before
    // DSYTRD): the Householder norms are plain sums of squares.  Eigenvalues scale back linearly.
    double *d_part = (double *)work;   // stage scratch, free until the reduction starts
    maxabs_lower(s, n, wA, ld, d_part, kBandW);
    double part[512];
    EK_HIP_CHECK(hipMemcpyAsync(part, d_part, sizeof(part), hipMemcpyDeviceToHost, s));
    EK_HIP_CHECK(hipStreamSynchronize(s));
    double anrm = 0.0, offband = 0.0;
    for (int q = 0; q < 256; ++q) { if (part[q] > anrm) anrm = part[q]; if (!(part[256 + q] <= offband)) offband = part[256 + q]; }
    // the reference's own inputs are sparse Hamiltonians, often banded: a matrix with nothing below its 64th subdiagonal
    // IS the output of the dense -> band stage (EK_HIP_BAND_INPUT=0 turns the short cut off)
    const char *band_env = getenv("EK_HIP_BAND_INPUT");
    band_input = !(band_env && atoi(band_env) == 0) && problem == 0 && !dist && offband == 0.0;
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
  if (problem == 1) {
    // right-looking sweep with one panel broadcast per strip: pays from three ranks on
    if (dist && g_comm.nranks >= dist_min_ranks()) {
      const PotrfMember me{wB, ld, dInv, g_ctx.d_info, pwork, g_comm.rank};
      potrf_lower_dist(s, g_ctx.stream2, n, 1, &me, team_exchange(0));
    } else if (potrf_rl) {
      potrf_lower_rl(s, g_ctx.stream2, n, wB, ld, dInv, g_ctx.d_info, pwork);
    } else {
      potrf_lower(s, n, wB, ld, dInv, g_ctx.d_info, twork);
    }
  }
  if (problem == 1 && n >= 256) {      // leaves of 256 for the solves of the reduction and of the recovery (ek_chol.hip)
    trtri256_blocks(s, n, wB, ld, dInv, inv256, twork);
    trsm_register_inv256(dInv, inv256, n);
  }
  mark();                                                              // 2
  if (pipe) {
    // (L is final, but its way out waits until A is in: the output workers' memcpys would share the host's memory
    // bandwidth with the input that the next stage is waiting for -- A came in at 24 GB/s beside them, B alone at 51)
    rc = stage_in_A(); if (rc) { tm.destroy(); return rc; }
    if (problem == 1) {        // L leaves while the reduction runs
      if (!aliasB) copy_matrix(s, n, n, wB, ld, dB, ldb);
      pipe->push(true, dB, ldb, pipe->hB, pipe->ldhb, n, n, pipe->mark(s), 0, /*lower=*/true);
    }
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
  // what the call leaves in A -- PDSYTRD's reflectors; after a two-stage reduction the band and the first stage's R factors
  //  -- goes back to the caller as soon as it is final: its place (X0) is needed again
  auto a_out = [&]() {
    if (!aliasA) copy_matrix(s, n, n, wA, ld, dA, lda);
    if (pipe) pipe->push(true, dA, lda, pipe->hA, pipe->ldha, n, n, pipe->mark(s), 0, /*lower=*/true);
  };
  if (dist && !two_stage) {
    const SytrdMember me{wA, ld, dd, de, dt, wV, ld, sytrd_work, g_comm.rank};
    sytrd_lower_dist(s, n, 1, &me, team_exchange(0, n));
    a_out();
  } else if (two_stage) {
    // dense -> band -> tridiagonal.  The panel factorisation of the first stage is CholeskyQR2 with a device-side check;
    // a panel it cannot handle (rank deficient, cond > 1e7: e.g. an input that is nearly banded) is factored by
    // Householder reflections inside the stage (per-panel rescue, ek_sy2sb.hip): the stage never gives up.
    EK_HIP_CHECK(hipMemsetAsync(dt1, 0, (size_t)ld * 8, s));
    double *AB0 = sb2st_band(work_sb2st, n);
    if (dist) {
      // On a team the first stage is distributed over the 128-wide column strips (strip S on rank S mod P: where
      // the distributed reduction to standard form left the matrix, so nothing is gathered in front of it): per panel
      // one broadcast of [V | T | tau] and one all-reduce of Y, the next panel's chain and broadcast on the second stream
      // beside the rest of the update (ek_sy2sb.hip).  Then ONE all-gather of the band (65 n doubles); the bulge chasing
      // and the D&C below its top merge run replicated, bit-identical on all ranks.
      const SytrdExchange x = team_exchange(0);
      const Sy2sbMember me{wA, ld, wV, ld, dt1, g_ctx.d_info + 2, work_sy2sb, g_comm.rank};
      sy2sb_lower_dist(s, g_ctx.stream2, n, 1, &me, x);
      double *ABs[1] = {AB0};
      pack_band(s, n, wA, ld, AB0);
      gather_band_strips(s, n, 1, g_comm.rank, ABs, x);
    } else {
      if (!band_input) sy2sb_lower(s, g_ctx.stream2, n, wA, ld, wV, ld, dt1, g_ctx.d_info + 2, work_sy2sb);
      pack_band(s, n, wA, ld, AB0);
    }
    a_out();                               // X0 changes hands: the matrix leaves, the reflectors of the chase move in
    EK_HIP_CHECK(hipMemsetAsync(wV2, 0, (size_t)ld * ld * 8, s));
    sb2st_lower(s, n, nullptr, 0, dd, de, wV2, ld, g_ctx.d_info + 2, work_sb2st, /*band_packed=*/true);
    // The low byte of the flag says that the bulge chasing abandoned a bounded wait (a matter of timing, not of the
    // data; nothing has raised it yet).  The stage is then repeated from the band it started from -- kept beside the
    // working copy -- with the older kernel alone, twice at most.  A team decides together: a flag that only one rank
    // has raised must not leave the ranks with eigenvectors of two different decompositions.
    for (int attempt = 0; ; ++attempt) {
      int flag = 0;
      EK_HIP_CHECK(hipMemcpyAsync(&flag, g_ctx.d_info + 2, sizeof(int), hipMemcpyDeviceToHost, s));
      EK_HIP_CHECK(hipStreamSynchronize(s));
      rescued_panels = (double)(flag >> 8);      // panels of the first stage that took the Householder rescue
      if (g_debug_fail_chase > 0) { --g_debug_fail_chase; flag |= 4; }
      int low = flag & 0xff;
      if (dist) { low = comm_any(low); if (low < 0) return low; }
      if (low == 0) { two_stage_done = true; break; }
      if (attempt == 2) return -992;
      flag &= ~0xff;
      EK_HIP_CHECK(hipMemcpyAsync(g_ctx.d_info + 2, &flag, sizeof(int), hipMemcpyHostToDevice, s));
      EK_HIP_CHECK(hipStreamSynchronize(s));
      EK_HIP_CHECK(hipMemsetAsync(wV2, 0, (size_t)ld * ld * 8, s));
      sb2st_lower(s, n, nullptr, 0, dd, de, wV2, ld, g_ctx.d_info + 2, work_sb2st, /*band_packed=*/true, /*chase_mode=*/1);
    }
  } else {
    sytrd_lower(s, n, wA, ld, dd, de, dt, wV, ld, sytrd_work);
    a_out();
  }
  mark();                                                              // 4
  // eigenvector columns wanted: the first n_vec, or this grid cell's share of them; the D&C
  // forms only those (columns 0..nc_loc-1 of wZ) and the two remaining stages treat the
  // columns of Z independently
  const StedcSelect pick{nc_loc, cell ? cell->nb : (n > 0 ? n : 1), split_rows ? g_comm.nranks : (cell ? cell->npcol : 1),
                         split_rows ? g_comm.rank : (cell ? cell->mycol : 0)};
  // On a team the heights right below the top merge are sharded as well (strips of the compact bases, one all-gather
  // round per P strips: ek_stedc.hip); the top merge forms this cell's columns only, as in the replicated-input mode.
  SytrdExchange dcx{};
  StedcTeam dct{0, 0, nullptr, 0};
  if (dist && g_comm.nranks >= 2) {
    // (the team form needs the compact bases, and EVERY rank must take the same decision or the all-gathers never meet:
    // a team of two on a ragged order can leave one rank with more than half of the columns -- then nobody shards)
    const int P = g_comm.nranks, nbz = cell->nb, Pc = split_rows ? P : cell->npcol;
    bool all_compact = true;
    for (int r = 0; r < Pc; ++r) all_compact = all_compact && stedc_compact(n, numroc0(n_vec, nbz, r, Pc));
    dcx = team_exchange(0);
    dct = StedcTeam{P, g_comm.rank, &dcx, all_compact ? stedc_team_levels(n, P) : 0};
  }
  else if (cell && !dist && g_debug_dc_team >= 2) dct = StedcTeam{g_debug_dc_team, -1, nullptr, stedc_team_levels(n, g_debug_dc_team)};
  stedc(s, n, dd, de, dwv, wZ, ld, work, g_ctx.d_info + 1, &pick, g_ctx.d_stats, nullptr, dct.levels > 0 ? &dct : nullptr);
  mark();                                                              // 5
  double *zc = wZ;
  // with a staging pipeline the LAST stage (the recovery; the back-transformation of a standard problem) runs in column
  // slabs, each leaving for the host while the next is computed (columns of Z are independent there)
  const int zslab = (pipe && two_stage_done && nc_loc > pipe->z_slab) ? pipe->z_slab : nc_loc;
  // (the last slabs are narrower: what remains exposed at the end of the call is the last slab's way out)
  auto slab_width = [&](int c0) {
    const int left = nc_loc - c0;
    if (zslab >= nc_loc) return left;
    if (left > 2 * zslab) return zslab;
    if (left > zslab) return zslab / 2 < left ? zslab / 2 : left;
    return (left > 512) ? (left + 1) / 2 : left;
  };
  auto z_out = [&](int c0, int nc) {
    if (!aliasZ) copy_matrix(s, n, nc, wZ + (size_t)c0 * ld, ld, dZ + (size_t)c0 * ldz, ldz);
    pipe->push(true, dZ + (size_t)c0 * ldz, ldz, pipe->hZ + (size_t)c0 * pipe->ldhz, pipe->ldhz, n, nc, pipe->mark(s), 0);
  };
  if (two_stage_done) {
    // (the records of Q2 are made from V2 here, after the D&C whose scratch they take over)
    sb2st_apply_q2(s, n, nc_loc, wV2, ld, zc, ld, g_ctx.d_info + 2, work_sb2st, q2rec);
    // (the T factors of the block reflectors do not depend on Z; forming them on the second stream beside the
    // bulge chasing was measured: the skinny GEMMs take CUs and issue slots from the latency-bound pipeline,
    // which loses 11 ms to gain 6)
    const bool q1 = !band_input;               // (a band on entry: the first stage did nothing, Q1 = I)
    if (q1) ormtr_prepare(s, n, wV, ld, dt1, q1prep);
    if (pipe && problem == 0 && zslab < nc_loc) {
      for (int c0 = 0, nc = 0; c0 < nc_loc; c0 += nc) {
        nc = slab_width(c0);
        if (q1) ormtr_apply(s, n, nc, wV, ld, q1prep, zc + (size_t)c0 * ld, ld, work, n_vec);
        z_out(c0, nc);
      }
    } else if (q1) {
      ormtr_apply(s, n, nc_loc, wV, ld, q1prep, zc, ld, work, n_vec);
    }
  } else {
    ormtr_lower(s, n, nc_loc, wV, ld, dt, zc, ld, work, n_vec);
  }
  mark();                                                              // 6
  if (problem == 1) {
    if (pipe) {
      for (int c0 = 0, nc = 0; c0 < nc_loc; c0 += nc) {
        nc = slab_width(c0);
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
    if (split_rows) {
      // every cell of this process column receives its rows of the columns the others formed: per peer one packed piece
      // each way (X1, free by now), then the pieces' columns take their places in the cell's block-cyclic piece
      const int nb = cell->nb, R = cell->nprow, C = cell->npcol, P = g_comm.nranks;
      double *buf = (double *)x1;
      int peers[kMaxTeam]; double *sp[kMaxTeam], *rp[kMaxTeam]; size_t sc[kMaxTeam], rcn[kMaxTeam];
      int np = 0;
      double *own = nullptr;
      for (int i = 0; i < R; ++i) {               // what goes out (and this cell's own rows of its own columns)
        const int nr_i = numroc0(n, nb, i, R);
        const size_t cnt = (size_t)nr_i * nc_loc;
        if (cnt > 0) gather_block_cyclic(s, nr_i, nc_loc, zc, ld, nb, R, i, 1, 0, buf, nr_i > 1 ? nr_i : 1);
        if (i == cell->myrow) own = buf;
        else { peers[np] = i * C + cell->mycol; sp[np] = buf; sc[np] = cnt; ++np; }
        buf += cnt;
      }
      for (int i = 0, q = 0; i < R; ++i) {        // what comes in
        if (i == cell->myrow) continue;
        rcn[q] = (size_t)nr_loc * numroc0(n_vec, nb, i * C + cell->mycol, P);
        rp[q] = buf; buf += rcn[q]; ++q;
      }
      team_sendrecv(s, np, peers, sp, sc, rp, rcn);
      const int ldp = nr_loc > 1 ? nr_loc : 1;
      for (int i = 0, q = 0; i < R; ++i) {
        const int nc_i = numroc0(n_vec, nb, i * C + cell->mycol, P);
        const double *src = (i == cell->myrow) ? own : rp[q++];
        scatter_block_cyclic(s, nr_loc, nc_i, src, ldp, nb, 1, 0, R, i, dZ, ldz);
      }
    } else if (cell) gather_block_cyclic(s, nr_loc, nc_loc, zc, ld, cell->nb, cell->nprow, cell->myrow, 1, 0, dZ, ldz);
    else if (!aliasZ) copy_matrix(s, n, n_vec, wZ, ld, dZ, ldz);
    if (problem == 1 && !aliasB) copy_matrix(s, n, n, wB, ld, dB, ldb);
  }
  mark();                                                              // 8
  EK_HIP_CHECK(hipGetLastError());
  int info[4] = {0, 0, 0, 0};
  EK_HIP_CHECK(hipMemcpyAsync(info, g_ctx.d_info, sizeof(info), hipMemcpyDeviceToHost, s));
  EK_HIP_CHECK(hipMemcpyAsync(g_ctx.stats, g_ctx.d_stats, sizeof(g_ctx.stats), hipMemcpyDeviceToHost, s));
  EK_HIP_CHECK(hipStreamSynchronize(s));
  g_ctx.stats[1] = two_stage_done ? 1.0 : 0.0;
  g_ctx.stats[2] = rescued_panels;
  g_ctx.stats[3] = (two_stage_done && band_input) ? 1.0 : 0.0;      // the band short cut was taken
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
  // the pipelined back-transformation was abandoned (a bounded wait ran out): on a team, for all ranks -- and an exchange
  // that failed on ANY rank (its sticky record travels in the same vote) ends the call on all of them with -996
  int bad = two_stage_done && info[2] != 0;
  if (dist) {
    bad = comm_any(bad);
    if (bad < 0) {
      if (g_comm.err) fprintf(stderr, "[ek_hip] exchange failed: %s\n", comm_error_string());
      return -996;
    }
  }
  if (bad) return -992;
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

namespace ek { namespace api { void release_pipe_streams() { g_pipe_streams.release(); } } }

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

// Pure host arithmetic (no GPU needed): bytes of workspace one whole-path call asks for -- on one GPU (nranks <= 1: all
// n_vec eigenvector columns) or as rank 0 of a 1 x nranks team with a communicator attached (its share of the columns,
// 64-wide blocks dealt round robin).  parts (optional, 6 entries): matrix bytes (one padded n x n array), persistent
// operators (L, the first stage's reflectors), eigenvector columns, X0, X1, the rest.
unsigned long long ek_hip_debug_workspace_bytes(int problem, int n, int n_vec, int nranks, unsigned long long *parts) {
  if (n < 1 || n_vec < 0 || n_vec > n) return 0;
  const int P = nranks > 1 ? nranks : 1;
  const int nc_loc = P > 1 ? numroc0(n_vec, 64, 0, P) : n_vec;
  const PathPlan p = plan_path(problem, n, n_vec, nc_loc, P > 1 ? P : 0);
  if (parts) {
    parts[0] = p.mat; parts[1] = 2 * p.mat; parts[2] = p.zmat; parts[3] = p.x0; parts[4] = p.x1;
    parts[5] = p.total - (2 * p.mat + p.zmat + p.x0 + p.x1);
  }
  return p.total;
}

// Test aid: the whole-path call treats its next `times` bulge chasings as if they had abandoned a bounded wait (the
// repetition from the saved band with the older kernel, and -992 after the third failure, are otherwise unreachable).
int ek_hip_debug_last_pipe_stats(double *out, int count) {
  std::lock_guard<std::mutex> lk(g_mu);
  for (int i = 0; i < count && i < 12; ++i) out[i] = g_pipe_stats[i];
  return 0;
}

// Rehearsal of the divide & conquer's team form on one GPU: while nranks >= 2, a grid cell solved WITHOUT a communicator
// (ek_hip_solve_device_grid, replicated inputs) forms the sharded heights as a team of nranks would, rank after rank (no
// exchange).  levels: heights below the top merge that are sharded (-1: the library's default for the order); profile != 0:
// HIP events around the call and every rank's sections, read by ek_hip_debug_stedc_team_get ([0] the D&C, [1] all ranks'
// sections, [2] the longest rank's per height: a rank of a real team computes for [0] - [1] + [2] seconds).
int ek_hip_debug_stedc_team(int nranks, int levels, int profile) {
  if (nranks < 0 || nranks > kMaxTeam) return -1;
  std::lock_guard<std::mutex> lk(g_mu);
  g_debug_dc_team = nranks;
  stedc_team_set_levels(levels);
  stedc_team_profile(profile != 0);
  return 0;
}
int ek_hip_debug_stedc_team_get(double *seconds) {
  if (!seconds) return -1;
  int rc = ensure_init(); if (rc) return rc;
  std::lock_guard<std::mutex> lk(g_mu);
  EK_HIP_CHECK(hipStreamSynchronize(g_ctx.stream));
  stedc_team_profile_collect(seconds);
  return 0;
}

int ek_hip_debug_fail_next_chase(int times) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_debug_fail_chase = times > 0 ? times : 0;
  return 0;
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
    g_comm.err = 0;
    rc = comm_agree(rc);                 // nobody enters the all-gathers below unless everybody can
    if (rc) return rc;
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
  // user-side device images (exact n x n) of A, B, Z and w: one allocation that the library keeps between calls like its
  // workspace (6 GiB of hipMalloc + hipFree per call were 0.05 s of a 1.0 s call at N = 16384) and releases in
  // ek_hip_finalize; the caller's pointers are borrowed for the duration of the call only, as before
  double *uA = nullptr, *uB = nullptr, *uZ = nullptr, *uw = nullptr;
  const size_t nn = (size_t)n * n * 8;
  auto t0 = std::chrono::steady_clock::now();
  {
    const size_t nnal = al(nn), need = (problem == 1 ? 3 : 2) * nnal + al((size_t)n * 8);
    rc = user_images(need, (void **)&uA);
    if (rc) return rc;
    uZ = (double *)((char *)uA + nnal);
    uw = (double *)((char *)uZ + nnal);
    if (problem == 1) uB = (double *)((char *)uw + al((size_t)n * 8));
  }
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
    // (only the lower triangles travel: the device images' upper triangles are zeros, never garbage)
    auto zero_first = [&](double *u) -> int {
      if (!pipe.lower_only) return 0;
      EK_HIP_CHECK(hipMemsetAsync(u, 0, nn, s));
      EK_HIP_CHECK(hipStreamSynchronize(s));
      return 0;
    };
    if (problem == 1) { rc = zero_first(uB); if (rc) return rc; pipe.push(false, uB, n, B_loc, desc_B[8], n, n, nullptr, 0, /*lower=*/true); }
    rc = zero_first(uA); if (rc) return rc;
    pipe.push(false, uA, n, A_loc, desc_A[8], n, n, nullptr, 1, /*lower=*/true);
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
    // A and B go back as uplo = 'L' arrays, in chunks of at most kTri columns as the pipeline's way out cuts them: the rows
    // below a chunk's diagonal block straight into the caller's array (one 2-D copy), the diagonal block through a scratch
    // of kTri x kTri doubles (2 MiB whatever the order -- EK_HIP_PIPE_MIN=0 sends N = 16384 down this path too), of which
    // the caller's array receives the entries on and below the diagonal only.  No allocation that can throw across the C ABI.
    constexpr int kTri = HostPipe::kTri;
    std::unique_ptr<double[]> tri(new (std::nothrow) double[(size_t)kTri * kTri]);
    auto d2h_lower = [&](const double *u, double *M_loc, int ldm) -> int {
      if (!tri) return -1000 - (int)hipErrorOutOfMemory;
      for (int a = 0; a < n; a += kTri) {
        const int b = (n - a < kTri) ? n : a + kTri, wdt = b - a;
        int r = d2h_matrix(wdt, wdt, u + (size_t)a * n + a, n, tri.get(), wdt, s);
        if (!r && b < n) r = d2h_matrix(n - b, wdt, u + (size_t)a * n + b, n, M_loc + (size_t)a * ldm + b, ldm, s);
        if (!r) { hipError_t e = hipStreamSynchronize(s); if (e != hipSuccess) r = -1000 - (int)e; }
        if (r) return r;
        for (int c = 0; c < wdt; ++c)
          memcpy(M_loc + (size_t)(a + c) * ldm + a + c, tri.get() + (size_t)c * wdt + c, (size_t)(wdt - c) * 8);
      }
      return 0;
    };
    if (!rc2) rc2 = d2h_lower(uA, A_loc, desc_A[8]);
    if (!rc2 && problem == 1) rc2 = d2h_lower(uB, B_loc, desc_B[8]);
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


}  // extern "C"
