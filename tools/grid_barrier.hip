// grid_barrier.hip -- cost of a software grid-wide barrier on MI355X (one workgroup per CU).
// Build: hipcc -O3 --offload-arch=gfx950 -o grid_barrier grid_barrier.hip ; run: ./grid_barrier [nwg] [iters]
// Every spin loop is bounded: if a barrier is not passed within ~2^22 polls all workgroups bail out.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ __launch_bounds__(256) void barrier_kernel(unsigned *cnt, unsigned *fail, int iters, int hier,
                                                      unsigned *xcnt) {
  extern __shared__ double lds[];   // sized to force one workgroup per CU
  const unsigned nwg = gridDim.x;
  if (threadIdx.x == 0) lds[0] = 0.0;
  for (int it = 0; it < iters; ++it) {
    __syncthreads();
    if (threadIdx.x == 0) {
      __threadfence();
      bool ok = true;
      if (!hier) {
        __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = (unsigned)(it + 1) * nwg;
        unsigned spins = 0;
        while (__hip_atomic_load(cnt, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
          if (++spins > (1u << 22) || __hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { ok = false; break; }
        }
      } else {
        // two levels: 8 groups by blockIdx % 8 (the XCD a workgroup is dispatched to), then one global counter
        const unsigned g = blockIdx.x & 7, gsz = (nwg + 7 - g) / 8;
        const unsigned old = __hip_atomic_fetch_add(&xcnt[g * 32], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1 == (unsigned)(it + 1) * gsz)
          __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = (unsigned)(it + 1) * (nwg < 8 ? nwg : 8);
        unsigned spins = 0;
        while (__hip_atomic_load(cnt, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
          if (++spins > (1u << 22) || __hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { ok = false; break; }
        }
      }
      if (!ok) __hip_atomic_store(fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (__hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
  }
}

int main(int argc, char **argv) {
  const int nwg = argc > 1 ? atoi(argv[1]) : 256, iters = argc > 2 ? atoi(argv[2]) : 2000;
  unsigned *d; hipMalloc(&d, 4096 * 4);
  hipFuncSetAttribute((const void *)barrier_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  int nb = 0;
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, barrier_kernel, 256, 100 * 1024);
  hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
  printf("CUs %d, resident workgroups per CU %d\n", pr.multiProcessorCount, nb);
  if (nwg > nb * pr.multiProcessorCount) { printf("grid does not fit\n"); return 1; }
  for (int hier = 0; hier < 2; ++hier) {
    for (int rep = 0; rep < 2; ++rep) {
      hipMemset(d, 0, 4096 * 4);
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(barrier_kernel, dim3(nwg), dim3(256), 100 * 1024, 0, d, d + 1024, iters, hier, d + 2048);
      hipEventRecord(e1, 0);
      hipError_t e = hipDeviceSynchronize();
      float ms = 0; hipEventElapsedTime(&ms, e0, e1);
      unsigned h[1025]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
      printf("%s barrier, %d workgroups, %d iterations: %.3f ms -> %.2f us per barrier (fail=%u, err=%d)\n",
             hier ? "two-level" : "flat", nwg, iters, ms, 1e3 * ms / iters, h[1024], (int)e);
    }
  }
  return 0;
}
