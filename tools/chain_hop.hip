// chain_hop.hip -- what the hand-off cycle of a position-owned bulge chase costs on MI355X.
// K workgroups (one band position each) in a chain.  Per "sweep" s workgroup k
//   * receives 65 doubles from k-1 (the reflector: a self-flagging mailbox line polled by wave 0),
//   * receives 65 doubles from k+1 that belong to sweep s-1 (the late numbers: polled by the last wave),
//   * passes two workgroup barriers (the partial sums), sends 65 doubles forward to k+1 and 65 back to
//     k-1, passes two more barriers (left application, shift).
// The sweep rate of the chain is bounded by the cycle forward hop + backward hop + what lies between;
// the printout is microseconds per sweep.  Every spin is bounded.
// Build: hipcc -O3 --offload-arch=gfx950 -o chain_hop chain_hop.hip ; run: ./chain_hop [K] [sweeps] [xcdmap 0|1|2] [pad] [plain]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

constexpr int MAILW = 72;
constexpr unsigned long long kEmpty = 0x7ff8dead0000beefull;
constexpr unsigned kSpin = 1u << 22;

__device__ __forceinline__ double ld_sc1(const double *p) {
  return __longlong_as_double((long long)__hip_atomic_load((const unsigned long long *)p, __ATOMIC_RELAXED,
                                                           __HIP_MEMORY_SCOPE_AGENT));
}
__device__ int g_plain;   // 1: plain stores (the line stays in the XCD's L2: only right when writer and reader share an XCD)
__device__ __forceinline__ void st_sc1(double *p, double v) {
  if (g_plain)
    __hip_atomic_store((unsigned long long *)p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_WORKGROUP);
  else
    __hip_atomic_store((unsigned long long *)p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool empty(double v) { return (unsigned long long)__double_as_longlong(v) == kEmpty; }

__global__ void init_kernel(double *m, int count) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) m[i] = __longlong_as_double((long long)kEmpty);
}

// fwd: [K][4][MAILW] lines written by k-1 for k; bwd: [K][4][MAILW] lines written by k+1 for k
__global__ __launch_bounds__(512) void chain_kernel(int K, int sweeps, int xcdmap, int pad, double *fwd, double *bwd,
                                                    unsigned *fail, double *out) {
  __shared__ double s_v[2][64];
  __shared__ int s_ok;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  // position of this workgroup: neighbours on one XCD (blocks b and b + 8 share one) or dealt round-robin
  int k = blockIdx.x;
  if (xcdmap == 1) { const int per = (K + 7) / 8; k = (blockIdx.x & 7) * per + (blockIdx.x >> 3); }
  if (xcdmap == 2) { if (blockIdx.x & 7) return; k = blockIdx.x >> 3; }   // every position on ONE XCD
  if (k >= K) return;
  if (t == 0) s_ok = 1;
  __syncthreads();
  double acc = 0.0;
  for (int s = 0; s < sweeps; ++s) {
    double *fin = fwd + ((size_t)k * 4 + (s & 3)) * MAILW;
    double *bin = bwd + ((size_t)k * 4 + ((s - 1) & 3)) * MAILW;
    if (wave == 0 && k > 0) {
      double a = ld_sc1(fin + lane), b = (lane == 0) ? ld_sc1(fin + 64) : 0.0;
      unsigned spins = 0;
      while (__any(empty(a) || (lane == 0 && empty(b)))) {
        if ((++spins & 63u) == 0u && (spins > kSpin || __hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
          if (lane == 0) { s_ok = 0; __hip_atomic_store(fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
          break;
        }
        __builtin_amdgcn_s_sleep(1);
        if (empty(a)) a = ld_sc1(fin + lane);
        if (lane == 0 && empty(b)) b = ld_sc1(fin + 64);
      }
      s_v[s & 1][lane] = a + b;
      const double e = __longlong_as_double((long long)kEmpty);
      st_sc1(fin + lane, e);
      if (lane == 0) st_sc1(fin + 64, e);
    }
    if (wave == 7 && s > 0 && k + 1 < K) {
      double a = ld_sc1(bin + lane), b = (lane == 0) ? ld_sc1(bin + 64) : 0.0;
      unsigned spins = 0;
      while (__any(empty(a) || (lane == 0 && empty(b)))) {
        if ((++spins & 63u) == 0u && (spins > kSpin || __hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
          if (lane == 0) { s_ok = 0; __hip_atomic_store(fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
          break;
        }
        __builtin_amdgcn_s_sleep(1);
        if (empty(a)) a = ld_sc1(bin + lane);
        if (lane == 0 && empty(b)) b = ld_sc1(bin + 64);
      }
      acc += a + b;
      const double e = __longlong_as_double((long long)kEmpty);
      st_sc1(bin + lane, e);
      if (lane == 0) st_sc1(bin + 64, e);
    }
    __syncthreads();                                        // #1
    if (!s_ok) return;
    acc += s_v[s & 1][lane];
    for (int i = 0; i < pad; ++i) acc = acc * 1.0000001 + 1e-9;   // stand-in for the partial sums
    __syncthreads();                                        // #2
    if (wave == 0) {
      if (k + 1 < K) {
        double *fo = fwd + ((size_t)(k + 1) * 4 + (s & 3)) * MAILW;
        st_sc1(fo + lane, acc + lane);
        if (lane == 0) st_sc1(fo + 64, acc);
      }
      if (k > 0) {
        double *bo = bwd + ((size_t)(k - 1) * 4 + (s & 3)) * MAILW;
        st_sc1(bo + lane, acc - lane);
        if (lane == 0) st_sc1(bo + 64, acc);
      }
    }
    for (int i = 0; i < pad; ++i) acc = acc * 1.0000001 + 1e-9;
    __syncthreads();                                        // #3
    for (int i = 0; i < pad; ++i) acc = acc * 1.0000001 + 1e-9;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                        // #4
  }
  out[(size_t)k * 512 + t] = acc;
}

__global__ void dpp_probe(double *x) {   // what wave_shl:1 / wave_shr:1 do on this chip (lane i <- lane i +- 1?)
  const double v = (double)threadIdx.x;
  const int lo = __builtin_amdgcn_update_dpp(-1, __double2loint(v), 0x130, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(-1, __double2hiint(v), 0x130, 0xf, 0xf, false);
  x[threadIdx.x] = __hiloint2double(hi, lo);
  const int lo2 = __builtin_amdgcn_update_dpp(-1, __double2loint(v), 0x138, 0xf, 0xf, false);
  const int hi2 = __builtin_amdgcn_update_dpp(-1, __double2hiint(v), 0x138, 0xf, 0xf, false);
  x[64 + threadIdx.x] = __hiloint2double(hi2, lo2);
}

int main(int argc, char **argv) {
  const int K = argc > 1 ? atoi(argv[1]) : 256, sweeps = argc > 2 ? atoi(argv[2]) : 4096;
  const int xcdmap = argc > 3 ? atoi(argv[3]) : 0, pad = argc > 4 ? atoi(argv[4]) : 0;
  double *fwd, *bwd, *out; unsigned *fail;
  const int count = K * 4 * MAILW;
  hipMalloc(&fwd, count * 8); hipMalloc(&bwd, count * 8); hipMalloc(&out, (size_t)K * 512 * 8 + 8 * 512 * 8); hipMalloc(&fail, 256);
  {
    hipLaunchKernelGGL(dpp_probe, dim3(1), dim3(64), 0, 0, out);
    double h[128]; hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    printf("wave_shl:1 lanes 0,1,15,16,31,32,62,63 <- %g %g %g %g %g %g %g %g\n", h[0], h[1], h[15], h[16], h[31], h[32], h[62], h[63]);
    printf("wave_shr:1 lanes 0,1,15,16,31,32,62,63 <- %g %g %g %g %g %g %g %g\n", h[64], h[65], h[79], h[80], h[95], h[96], h[126], h[127]);
  }
  const int plain = argc > 5 ? atoi(argv[5]) : 0;
  hipMemcpyToSymbol(HIP_SYMBOL(g_plain), &plain, sizeof(int));
  const int grid = xcdmap == 2 ? 8 * K : xcdmap ? ((K + 7) / 8) * 8 : K;
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(init_kernel, dim3((count + 255) / 256), dim3(256), 0, 0, fwd, count);
    hipLaunchKernelGGL(init_kernel, dim3((count + 255) / 256), dim3(256), 0, 0, bwd, count);
    hipMemset(fail, 0, 256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(chain_kernel, dim3(grid), dim3(512), 0, 0, K, sweeps, xcdmap, pad, fwd, bwd, fail, out);
    hipEventRecord(e1, 0);
    hipError_t e = hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    unsigned hf = 0; hipMemcpy(&hf, fail, 4, hipMemcpyDeviceToHost);
    printf("K=%d sweeps=%d xcdmap=%d pad=%d plain=%d: %.3f ms -> %.3f us per sweep (fail=%u err=%d)\n", K, sweeps, xcdmap, pad, plain, ms,
           1e3 * ms / (sweeps + K), hf, (int)e);
  }
  return 0;
}
