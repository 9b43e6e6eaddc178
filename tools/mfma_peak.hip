// fp64 MFMA ceiling on MI355X: back-to-back v_mfma_f64_16x16x4_f64 on registers with NACC independent
// accumulators per wave and W waves per SIMD, plus the shader clock the loop actually ran at
// (s_memtime cycles / wall time).  Build: hipcc -O3 --offload-arch=gfx950 -o mfma_peak mfma_peak.hip
// The datasheet figure (78.6 TFLOP/s) assumes 2.4 GHz and one MFMA issued every 64 cycles per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

// FILL: number of independent integer VALU instructions issued after every MFMA (what a real GEMM does
// between its matrix instructions: LDS reads, address arithmetic)
template <int NACC, int FILL = 0>
__global__ __launch_bounds__(256) void k(double *out, long long *cyc, int iters) {
  unsigned dummy = threadIdx.x;
  d4 a[NACC];
#pragma unroll
  for (int q = 0; q < NACC; ++q) a[q] = (d4){0, 0, 0, 0};
  // distinct operand registers per accumulator (as a GEMM has), so that no two instructions in flight
  // read the same source registers
  double xs[NACC], ys[NACC];
#pragma unroll
  for (int q = 0; q < NACC; ++q) { xs[q] = threadIdx.x * 1e-3 + q; ys[q] = 1.0 - xs[q] * 0.5; }
  const long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int q = 0; q < NACC; ++q) {
      a[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(xs[q], ys[q], a[q], 0, 0, 0);
#pragma unroll
      for (int f = 0; f < FILL; ++f) asm volatile("v_add_u32 %0, %0, 1" : "+v"(dummy));
    }
  }
  const long long t1 = clock64();
  double s = 0.0;
#pragma unroll
  for (int q = 0; q < NACC; ++q) s += a[q][q & 3];
  out[blockIdx.x * 256 + threadIdx.x] = s + dummy;
  if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <int NACC, int FILL = 0>
void run(double *d, long long *c) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wg = 256; wg <= 2048; wg *= 2) {
    const int iters = 40000 / NACC * 4;
    k<NACC, FILL><<<wg, 256>>>(d, c, 1000);
    hipDeviceSynchronize();
    hipEventRecord(e0); k<NACC, FILL><<<wg, 256>>>(d, c, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long cy; hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost);
    const double fl = (double)wg * 4 * iters * NACC * 2048.0;
    printf("acc/wave=%2d fill=%d  waves/SIMD=%.0f : %6.1f TFLOP/s  (%.2f ms, %.0f cycles per MFMA per wave, clock %.2f GHz)\n", NACC, FILL,
           wg * 4 / 1024.0, fl / ms * 1e-9, ms, (double)cy / ((double)iters * NACC), (double)cy / (ms * 1e6));
  }
}
int main() {
  double *d; hipMalloc(&d, 8 * 256 * 4096);
  long long *c; hipMalloc(&c, 64);
  run<1>(d, c); run<2>(d, c); run<4>(d, c); run<8>(d, c); run<16>(d, c);
  run<8, 2>(d, c); run<8, 6>(d, c); run<8, 12>(d, c);
  return 0;
}
