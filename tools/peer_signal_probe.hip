// tools/peer_signal_probe.hip -- feasibility probe (not product): can a stream wait on / write to a
// 64-bit flag that lives in another process's device memory (IPC-mapped), i.e. can the per-column
// exchange of the distributed tridiagonalisation be signalled by the command processors instead of
// by a collective's kernel?
//
//   hipcc --offload-arch=gfx950 -O2 -o build/peer_signal_probe tools/peer_signal_probe.hip
//   timeout -k 5 60 build/peer_signal_probe
//
// Two processes (fork BEFORE any HIP call) on device 0.  Each allocates one region of the tested
// kind (signal memory / fine-grained / plain), exports it with hipIpcGetMemHandle, maps the
// peer's, and they play ping-pong: write my sequence number into the PEER's flag with
// hipStreamWriteValue64, wait for the peer's number in MY flag with hipStreamWaitValue64, a small
// kernel in between that reads a payload the peer stored into my region before raising the flag.
// Every wait is guarded: the host polls the stream with a deadline and gives up (exit code 3).
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sys/wait.h>
#include <unistd.h>

#define CK(x)                                                                                  \
  do {                                                                                         \
    hipError_t e_ = (x);                                                                       \
    if (e_ != hipSuccess) {                                                                    \
      fprintf(stderr, "[%d] %s -> %s\n", g_me, #x, hipGetErrorString(e_));                      \
      return 2;                                                                                \
    }                                                                                          \
  } while (0)

static int g_me = 0;

__global__ void push_kernel(double *peer_payload, double v, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) peer_payload[i] = v + i;
}
__global__ void check_kernel(const double *my_payload, double v, int n, int *bad) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && my_payload[i] != v + i) atomicAdd(bad, 1);
}

static bool wait_stream(hipStream_t s, double seconds) {
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    hipError_t e = hipStreamQuery(s);
    if (e == hipSuccess) return true;
    if (e != hipErrorNotReady) return false;
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > seconds) return false;
    usleep(200);
  }
}

static int run(int me, int rd, int wr, int kind, int rounds) {
  g_me = me;
  CK(hipSetDevice(0));
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  const int n = 4096;
  const size_t bytes = 4096 + (size_t)n * sizeof(double);   // [flag | ... | payload]
  char *mine = nullptr;
  if (kind == 0) CK(hipExtMallocWithFlags((void **)&mine, bytes, hipMallocSignalMemory));
  else if (kind == 1) CK(hipExtMallocWithFlags((void **)&mine, bytes, hipDeviceMallocFinegrained));
  else if (kind == 2) CK(hipExtMallocWithFlags((void **)&mine, bytes, hipDeviceMallocUncached));
  else CK(hipMalloc((void **)&mine, bytes));
  CK(hipMemset(mine, 0, bytes));
  CK(hipDeviceSynchronize());
  hipIpcMemHandle_t hm, hp;
  CK(hipIpcGetMemHandle(&hm, mine));
  if (write(wr, &hm, sizeof(hm)) != (ssize_t)sizeof(hm)) return 2;
  if (read(rd, &hp, sizeof(hp)) != (ssize_t)sizeof(hp)) return 2;
  char *peer = nullptr;
  CK(hipIpcOpenMemHandle((void **)&peer, hp, hipIpcMemLazyEnablePeerAccess));
  int *bad = nullptr;
  CK(hipMalloc((void **)&bad, sizeof(int)));
  CK(hipMemset(bad, 0, sizeof(int)));
  // both sides ready
  char c = 1;
  if (write(wr, &c, 1) != 1 || read(rd, &c, 1) != 1) return 2;
  // (b) the exchange as the library does it: ONE handshake per round, payload slots alternating
  // with the parity of the round; host enqueue time measured apart from completion
  {
    const int n2 = 2048;
    double *my_slots = (double *)(mine + 4096), *peer_slots = (double *)(peer + 4096);
    const auto tb0 = std::chrono::steady_clock::now();
    for (int r = 1; r <= rounds; ++r) {
      const int par = r & 1;
      hipLaunchKernelGGL(push_kernel, dim3(n2 / 256), dim3(256), 0, s, peer_slots + par * n2, (double)(7000 * r + me), n2);
      CK(hipStreamWriteValue64(s, peer + 16, (uint64_t)r, 0));
      CK(hipStreamWaitValue64(s, mine + 16, (uint64_t)r, hipStreamWaitValueGte, 0xFFFFFFFFFFFFFFFFull));
      hipLaunchKernelGGL(check_kernel, dim3(n2 / 256), dim3(256), 0, s, (const double *)(my_slots + par * n2),
                         (double)(7000 * r + (1 - me)), n2, bad);
    }
    const double t_enq = std::chrono::duration<double>(std::chrono::steady_clock::now() - tb0).count();
    if (!wait_stream(s, 20.0)) { fprintf(stderr, "[%d] kind %d: one-handshake loop stuck\n", me, kind); _exit(3); }
    const double t_all = std::chrono::duration<double>(std::chrono::steady_clock::now() - tb0).count();
    int hb = -1;
    CK(hipMemcpy(&hb, bad, sizeof(int), hipMemcpyDeviceToHost));
    printf("[%d] kind %d one handshake per round (2 kernels, 1 write, 1 wait): %.2f us per round, host enqueue %.2f us per round, %d bad\n",
           me, kind, 1e6 * t_all / rounds, 1e6 * t_enq / rounds, hb);
    // (c) is the batched form accepted for these addresses?
    hipStreamBatchMemOpParams ops[2];
    memset(ops, 0, sizeof(ops));
    ops[0].operation = hipStreamMemOpWriteValue64;
    ops[0].writeValue.address = (hipDeviceptr_t)(peer + 24);
    ops[0].writeValue.value64 = 1;
    ops[1].operation = hipStreamMemOpWaitValue64;
    ops[1].waitValue.address = (hipDeviceptr_t)(mine + 24);
    ops[1].waitValue.value64 = 1;
    ops[1].waitValue.flags = hipStreamWaitValueGte;
    hipError_t eb = hipStreamBatchMemOp(s, 2, ops, 0);
    printf("[%d] kind %d hipStreamBatchMemOp(write peer, wait mine): %s\n", me, kind, hipGetErrorString(eb));
    (void)hipGetLastError();
    if (eb == hipSuccess && !wait_stream(s, 10.0)) { fprintf(stderr, "[%d] batch stuck\n", me); _exit(3); }
    fflush(stdout);
  }
  const auto t0 = std::chrono::steady_clock::now();
  for (int r = 1; r <= rounds; ++r) {
    // my payload for this round into the peer's region, then raise my number in the peer's flag
    hipLaunchKernelGGL(push_kernel, dim3(n / 256), dim3(256), 0, s, (double *)(peer + 4096), (double)(1000 * r + me), n);
    CK(hipStreamWriteValue64(s, peer, (uint64_t)r, 0));
    // wait for the peer's number in my flag, then check what it stored
    CK(hipStreamWaitValue64(s, mine, (uint64_t)r, hipStreamWaitValueGte, 0xFFFFFFFFFFFFFFFFull));
    hipLaunchKernelGGL(check_kernel, dim3(n / 256), dim3(256), 0, s, (const double *)(mine + 4096),
                       (double)(1000 * r + (1 - me)), n, bad);
    // the peer may overwrite my payload for round r+1 only after I have checked round r: one more
    // handshake through a second flag word
    CK(hipStreamWriteValue64(s, peer + 8, (uint64_t)r, 0));
    CK(hipStreamWaitValue64(s, mine + 8, (uint64_t)r, hipStreamWaitValueGte, 0xFFFFFFFFFFFFFFFFull));
    if ((r & 63) == 0 && !wait_stream(s, 10.0)) { fprintf(stderr, "[%d] kind %d: stuck at round %d\n", me, kind, r); _exit(3); }
  }
  if (!wait_stream(s, 10.0)) { fprintf(stderr, "[%d] kind %d: stuck at the end\n", me, kind); _exit(3); }
  const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  int hbad = -1;
  CK(hipMemcpy(&hbad, bad, sizeof(int), hipMemcpyDeviceToHost));
  printf("[%d] kind %d (%s): %d rounds ok, %d bad values, %.2f us per round (2 handshakes, 2 kernels)\n", me, kind,
         kind == 0 ? "signal memory" : kind == 1 ? "fine-grained" : kind == 2 ? "uncached" : "plain hipMalloc", rounds,
         hbad, 1e6 * dt / rounds);
  fflush(stdout);
  CK(hipIpcCloseMemHandle(peer));
  return hbad == 0 ? 0 : 4;
}

int main(int argc, char **argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 2000;
  int rc_all = 0;
  for (int kind = 0; kind < 4; ++kind) {
    int p2c[2], c2p[2];
    if (pipe(p2c) || pipe(c2p)) return 1;
    fflush(stdout);
    const pid_t pid = fork();   // before any HIP call in this process (run() is only called in children)
    if (pid == 0) {
      const pid_t pid2 = fork();
      if (pid2 == 0) _exit(run(1, p2c[0], c2p[1], kind, rounds));
      const int rc0 = run(0, c2p[0], p2c[1], kind, rounds);
      int st = 0;
      // give the other side a moment, then make sure it is gone
      for (int i = 0; i < 100 && waitpid(pid2, &st, WNOHANG) == 0; ++i) usleep(100000);
      kill(pid2, SIGKILL);
      _exit(rc0 ? rc0 : (WIFEXITED(st) ? WEXITSTATUS(st) : 5));
    }
    int st = 0;
    waitpid(pid, &st, 0);
    const int rc = WIFEXITED(st) ? WEXITSTATUS(st) : 9;
    printf("kind %d: exit %d\n", kind, rc);
    fflush(stdout);
    if (rc) rc_all = rc;
    close(p2c[0]); close(p2c[1]); close(c2p[0]); close(c2p[1]);
  }
  return 0 * rc_all;
}
