// fp64 MFMA ceiling on MI355X: back-to-back v_mfma_f64_16x16x4_f64 on registers with NACC independent
// accumulators per wave and W waves per SIMD, plus the shader clock the loop actually ran at
// (s_memtime cycles / wall time).  Build: hipcc -O3 --offload-arch=gfx950 -o mfma_peak mfma_peak.hip
// The datasheet figure (78.6 TFLOP/s) assumes 2.4 GHz and one MFMA issued every 64 cycles per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k(double *out, long long *cyc, int iters) {
  d4 a[NACC];
#pragma unroll
  for (int q = 0; q < NACC; ++q) a[q] = (d4){0, 0, 0, 0};
  double x = threadIdx.x * 1e-3, y = 1.0 - x;
  const long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int q = 0; q < NACC; ++q) a[q] = __builtin_amdgcn_mfma_f64_16x16x4f64((q & 1) ? x : y, (q & 2) ? x : y, a[q], 0, 0, 0);
  }
  const long long t1 = clock64();
  double s = 0.0;
#pragma unroll
  for (int q = 0; q < NACC; ++q) s += a[q][q & 3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <int NACC>
void run(double *d, long long *c) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wg = 256; wg <= 2048; wg *= 2) {
    const int iters = 40000 / NACC * 4;
    k<NACC><<<wg, 256>>>(d, c, 1000);
    hipDeviceSynchronize();
    hipEventRecord(e0); k<NACC><<<wg, 256>>>(d, c, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long cy; hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost);
    const double fl = (double)wg * 4 * iters * NACC * 2048.0;
    printf("acc/wave=%2d  waves/SIMD=%.0f : %6.1f TFLOP/s  (%.2f ms, %.0f cycles per MFMA per wave, clock %.2f GHz)\n", NACC,
           wg * 4 / 1024.0, fl / ms * 1e-9, ms, (double)cy / ((double)iters * NACC), (double)cy / (ms * 1e6));
  }
}
int main() {
  double *d; hipMalloc(&d, 8 * 256 * 4096);
  long long *c; hipMalloc(&c, 64);
  run<1>(d, c); run<2>(d, c); run<4>(d, c); run<8>(d, c); run<16>(d, c);
  return 0;
}
