// ek_api_internal.h -- what the translation units behind the C-ABI of libek_hip.so share (not installed):
//   ek_api.hip    boundary: context, memory, the stage-level entries (one per ScaLAPACK call), acceptance checks
//   ek_comm.hip   the communicator that stands where the reference has its BLACS context: RCCL (bound at run time),
//                 the host-hook exchange, the team's agreements
//   ek_solve.hip  the whole-path driver (solve_device_locked), the staging pipeline of the host path, ek_hip_solve*
//   ek_debug.hip  tuning / profiling / rehearsal hooks of include/ek_hip_debug.h and the *_team entries
#pragma once
#include "../../include/ek_hip.h"
#include "../../include/ek_hip_debug.h"
#include "ek_common.h"

#include <rccl/rccl.h>   // types only: the library is bound at run time (dlopen), see ek_comm.hip

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

namespace ek {
namespace api {

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
extern Context g_ctx;
extern std::mutex g_mu;

int ensure_init();
int workspace(size_t bytes, void **p);

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

void release_scratch_choice();
int user_images(size_t bytes, void **p);   // device images of host arrays, kept between calls (ek_api.hip)
void release_user_images();
void release_pipe_streams();       // the staging pipeline's copy streams (ek_solve.hip)
void *choose_sytrd_scratch(int n, int ld, double *wA, void *arena_work, double *vecs, size_t need);

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

int check_desc(const int *desc, int argpos, int m, int n, int lld_rows = -1);
int numroc0(int n, int nb, int me, int np);
// Owner cell of a process grid for the replicated-input mode (ek_hip_solve_replicated)
struct GridCell { int nb, nprow, npcol, myrow, mycol; };

// Exchange hook for block-cyclically distributed inputs (ek_hip_set_allgatherv)
extern ek_hip_allgatherv_fn g_allgatherv;
extern void *g_allgatherv_user;

int pad_ld(int n);
int h2d_matrix(int m, int n, const double *h, int ldh, double *d, int ldd, hipStream_t s);
int d2h_matrix(int m, int n, const double *d, int ldd, double *h, int ldh, hipStream_t s);
int fetch_info(int *info);
// test aids: NaN into the strips a member does not own; entries in which two arrays differ (bitwise)
void poison_foreign_strips(hipStream_t s, int n, double *A, int lda, int P, int rank);
void count_mismatch(hipStream_t s, int m, int n, const double *X, int ldx, const double *Y, int ldy, int lower,
                    unsigned long long *count);

// ---- communicator (ek_comm.hip)
struct Comm {
  bool on = false;
  bool host = false;    // exchanges go through the host's allgatherv hook instead of RCCL
  ncclComm_t comm = nullptr;
  int nranks = 0, rank = 0;
  int err = 0;          // first failing collective since the last check (ncclResult_t)
};
extern Comm g_comm;
constexpr int kPotrfRlMin = 1024;
extern int g_two_stage_min;
int two_stage_min();
int dist_min_ranks();
void comm_teardown();     // the communicator itself (ek_hip_finalize)
SytrdExchange team_exchange(int nteam, int n = 0);
// Pairwise exchange over the attached communicator (RCCL: grouped ncclSend / ncclRecv; host communicator: two rounds of
// the allgatherv hook): this rank sends send[i] (send_counts[i] doubles, device) to world rank peers[i] and receives
// recv_counts[i] doubles from it into recv[i]; stream-ordered, collective over the communicator (a rank with no peer
// calls it with npeers = 0).  Failures are recorded in g_comm.err like those of every other exchange.
void team_sendrecv(hipStream_t s, int npeers, const int *peers, double *const *send, const size_t *send_counts,
                   double *const *recv, const size_t *recv_counts);
int comm_any(int local);
int comm_agree(int local_rc);
const char *comm_error_string();

// ---- whole path (ek_solve.hip)
void gather_band_strips(hipStream_t s, int n, int nmem, int rank0, double *const *ABs, const SytrdExchange &x);

}  // namespace api
}  // namespace ek
