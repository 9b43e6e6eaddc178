// fp64 MFMA peak micro-benchmark: back-to-back v_mfma_f64_16x16x4_f64 on registers,
// 4 independent accumulators per wave, W waves per SIMD.  Prints achieved TFLOP/s.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k(double *out, int iters) {
  d4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
  double x = threadIdx.x * 1e-3, y = 1.0 - x;
  for (int i = 0; i < iters; ++i) {
    a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, x, a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, y, a3, 0, 0, 0);
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
}
int main() {
  double *d; hipMalloc(&d, 8 * 256 * 4096);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wg = 256; wg <= 2048; wg *= 2) {
    const int iters = 20000;
    k<<<wg, 256>>>(d, 1000);
    hipDeviceSynchronize();
    hipEventRecord(e0); k<<<wg, 256>>>(d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double fl = (double)wg * 4 /*waves*/ * iters * 4 * 2048.0;
    printf("wgs=%d (%.1f waves/SIMD): %.1f TFLOP/s (%.2f ms)\n", wg, wg * 4 / 1024.0, fl / ms * 1e-9, ms);
  }
  return 0;
}
