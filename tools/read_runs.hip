// read_runs.hip -- read bandwidth of MI355X for runs of R bytes at a stride of S bytes (tools, not product): the access pattern
// of a column-major tile walker (R = 8 x tile rows per column, S = 8 x leading dimension).  Every workgroup of 256 threads
// streams its share of `cols` runs with 16-byte loads, eight in flight per thread; prints GB/s per R.
// Build: hipcc -O3 --offload-arch=gfx950 -o read_runs read_runs.hip ; run: ./read_runs [stride_bytes] [GiB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double double2_t __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256, 2) void read_kernel(const double *__restrict__ base, size_t stride_d, int run_d, long long nruns,
                                                      int tile_rows_d, double *out) {
  // runs are visited tile by tile: `per` threads share one run (run_d / 2 pairs), 256 / per runs per pass
  const int per = run_d / 2 < 256 ? run_d / 2 : 256;            // threads per run (pairs of doubles)
  const int rpp = 256 / per;                                    // runs per pass of the workgroup
  const int t = threadIdx.x, lane = t % per, sub = t / per;
  double2_t acc = (double2_t){0.0, 0.0};
  for (long long r0 = (long long)blockIdx.x * rpp * 8; r0 < nruns; r0 += (long long)gridDim.x * rpp * 8) {
    double2_t v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const long long r = r0 + (long long)i * rpp + sub;
      // run r: column r % 4096 of "tile row" r / 4096 (tile_rows_d doubles further down the matrix)
      const double *p = base + (size_t)(r % 4096) * stride_d + (size_t)(r / 4096) * tile_rows_d + 2 * lane;
      v[i] = (r < nruns) ? *reinterpret_cast<const double2_t *>(p) : (double2_t){0.0, 0.0};
      if (per < run_d / 2) {                                     // long runs: the rest of the run
        for (int q = per; q < run_d / 2; q += per) v[i] += *reinterpret_cast<const double2_t *>(p + 2 * q);
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) acc += v[i];
  }
  if (acc.x + acc.y == 1.2345e300) out[0] = acc.x;
}
int main(int argc, char **argv) {
  const size_t stride = argc > 1 ? (size_t)atoll(argv[1]) : 131072;
  const double gib = argc > 2 ? atof(argv[2]) : 8.0;
  const size_t bytes = (size_t)(gib * (1 << 30));
  double *buf, *out;
  hipMalloc(&buf, bytes + (1 << 20)); hipMalloc(&out, 64);
  hipMemset(buf, 0, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int run = 256; run <= 65536; run *= 2) {
    // a "matrix" of 4096 columns at the given stride; tile rows of `run` bytes stacked down the columns
    const int run_d = run / 8;
    long long tiles = (long long)(stride / run);                  // runs per column that fit the stride
    if (tiles < 1) break;
    long long nruns = 4096 * tiles;
    if ((size_t)nruns * run > bytes) nruns = (long long)(bytes / run);
    if (4096 * stride > bytes) { printf("buffer too small for 4096 columns at this stride\n"); break; }
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(read_kernel, dim3(512), dim3(256), 0, 0, buf, stride / 8, run_d, nruns, run_d, out);
      hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("run %6d B at stride %zu B: %8.1f GB/s (%lld runs, %.2f GB)\n", run, stride, (double)nruns * run / ms / 1e6, nruns, (double)nruns * run / 1e9);
  }
  return 0;
}
