// ek_util.hip -- small data-movement kernels (copies, fills, symmetrisation, column gather).
// All are plain HBM streaming kernels: one thread per element, consecutive lanes on
// consecutive rows of the column-major arrays so every wave access is a 512-byte run.
#include "ek_common.h"

#include <vector>

namespace ek {
namespace {

__global__ void copy_matrix_kernel(int m, int n, const double *__restrict__ src, int lds,
                                   double *__restrict__ dst, int ldd) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  for (int j = blockIdx.y; j < n; j += gridDim.y)
    dst[(size_t)i + (size_t)j * ldd] = src[(size_t)i + (size_t)j * lds];
}

__global__ void set_matrix_kernel(int m, int n, double offdiag, double diag, double *A, int lda) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  for (int j = blockIdx.y; j < n; j += gridDim.y)
    A[(size_t)i + (size_t)j * lda] = (i == j) ? diag : offdiag;
}

// upper <- lower^T through a 32x33 LDS tile so both the read and the write are coalesced
__global__ void symmetrize_kernel(int n, double *A, int lda) {
  __shared__ double tile[32][33];
  const int bi = blockIdx.x, bj = blockIdx.y;   // tile row / col, only bi >= bj do work
  if (bi < bj) return;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  for (int c = ty; c < 32; c += 8) {
    const int i = bi * 32 + tx, j = bj * 32 + c;
    tile[c][tx] = (i < n && j < n) ? A[(size_t)i + (size_t)j * lda] : 0.0;
  }
  __syncthreads();
  for (int c = ty; c < 32; c += 8) {
    // A(bj*32 + tx, bi*32 + c) <- A(bi*32 + c, bj*32 + tx) = tile[tx][c]
    const int i = bj * 32 + tx, j = bi * 32 + c;
    if (i < n && j < n && i < j) A[(size_t)i + (size_t)j * lda] = tile[tx][c];
  }
}

// dst(i, c) = src(r0 + c, i): 32x33 LDS tile, coalesced on both sides
__global__ void transpose_rows_kernel(int n, int m, const double *__restrict__ src, int lds, int r0,
                                      double *__restrict__ dst, int ldd) {
  __shared__ double tile[32][33];
  const int bi = blockIdx.x, bc = blockIdx.y;   // tile along i (0..n), along c (0..m)
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int k = ty; k < 32; k += 8) {
    const int c = bc * 32 + tx, i = bi * 32 + k;            // read src(r0 + c, i): lanes along c (rows)
    tile[k][tx] = (c < m && i < n) ? src[(size_t)(r0 + c) + (size_t)i * lds] : 0.0;
  }
  __syncthreads();
  for (int k = ty; k < 32; k += 8) {
    const int i = bi * 32 + tx, c = bc * 32 + k;            // write dst(i, c): lanes along i (rows)
    if (i < n && c < m) dst[(size_t)i + (size_t)c * ldd] = tile[tx][k];
  }
}

__global__ void gather_columns_kernel(int m, int n, const double *__restrict__ src, int lds,
                                      const int *__restrict__ perm, double *__restrict__ dst,
                                      int ldd) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  for (int j = blockIdx.y; j < n; j += gridDim.y)
    dst[(size_t)i + (size_t)j * ldd] = src[(size_t)i + (size_t)perm[j] * lds];
}

// dst(lr, lc) = src(global row of local row lr, global column of local column lc) for the
// block-cyclic owner (me_r, me_c) of a pr x pc grid, square blocks nb, source process (0,0)
// (the map of distribute_matrix.f90:128-138 / ScaLAPACK INDXL2G).
__global__ void bc_gather_kernel(int mr, int nc, const double *__restrict__ src, int lds, int nb,
                                 int pr, int me_r, int pc, int me_c, double *__restrict__ dst,
                                 int ldd) {
  const int lr = blockIdx.x * blockDim.x + threadIdx.x;
  if (lr >= mr) return;
  const size_t gr = (size_t)((lr / nb) * pr + me_r) * nb + lr % nb;
  for (int lc = blockIdx.y; lc < nc; lc += gridDim.y) {
    const size_t gc = (size_t)((lc / nb) * pc + me_c) * nb + lc % nb;
    dst[(size_t)lr + (size_t)lc * ldd] = src[gr + gc * lds];
  }
}

__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x) {
  x += 0x9E3779B97F4A7C15ULL;
  unsigned long long z = x;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}

// M(i,j) = M(j,i) = u(seed, max(i,j), min(i,j)) / sqrt(n) + 2 [i == j]   
__global__ void synth_kernel(int n, unsigned long long seed, double inv, double *M, int ldm) {
#pragma clang fp contract(off)   // product and sum round separately, as in the CPU generator
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  for (int j = blockIdx.y; j < n; j += gridDim.y) {
    const unsigned long long hi = i > j ? i : j, lo = i > j ? j : i;
    const unsigned long long r = splitmix64((seed << 40) + hi * (unsigned long long)n + lo);
    const double u = (double)(r >> 11) * (1.0 / 4503599627370496.0) - 1.0;
    const double prod = u * inv;
    M[(size_t)i + (size_t)j * ldm] = prod + (i == j ? 2.0 : 0.0);
  }
}

// out[0] = max |A(i,j)| over the lower triangle (one workgroup, grid-stride; tiny vs the solve)
// (partial[gridDim.x + blockIdx.x]: the same over the entries more than `bw` below the diagonal -- all zero for a matrix
// that is already a band of that half width)
__global__ __launch_bounds__(1024) void maxabs_lower_kernel(int n, const double *__restrict__ A, int lda,
                                                            double *__restrict__ partial, int bw) {
  __shared__ double red[1024];
  double m = 0.0, mo = 0.0;
  for (int j = blockIdx.x; j < n; j += gridDim.x)
    for (int i = j + threadIdx.x; i < n; i += 1024) {
      const double v = fabs(A[(size_t)i + (size_t)j * lda]);
      const double w = (v <= 1.7e308) ? v : INFINITY;    // NaN and Inf both surface as Inf
      m = fmax(m, w);
      if (i - j > bw) mo = fmax(mo, w);
    }
  for (int pass = 0; pass < 2; ++pass) {
    red[threadIdx.x] = pass ? mo : m;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + o]); __syncthreads(); }
    if (threadIdx.x == 0) partial[pass * gridDim.x + blockIdx.x] = red[0];
    __syncthreads();
  }
}

__global__ void scale_lower_kernel(int n, double alpha, double *A, int lda) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  for (int j = blockIdx.y; j < n && j <= i; j += gridDim.y) A[(size_t)i + (size_t)j * lda] *= alpha;
}

__global__ void scale_vector_kernel(int n, double alpha, double *x) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] *= alpha;
}

inline dim3 grid2d(int m, int n) {
  return dim3(ceil_div(m, 256), n < 4096 ? (n > 0 ? n : 1) : 4096);
}

}  // namespace

void copy_matrix(hipStream_t s, int m, int n, const double *src, int lds, double *dst, int ldd) {
  if (m <= 0 || n <= 0) return;
  hipLaunchKernelGGL(copy_matrix_kernel, grid2d(m, n), dim3(256), 0, s, m, n, src, lds, dst, ldd);
}
void set_matrix(hipStream_t s, int m, int n, double offdiag, double diag, double *A, int lda) {
  if (m <= 0 || n <= 0) return;
  hipLaunchKernelGGL(set_matrix_kernel, grid2d(m, n), dim3(256), 0, s, m, n, offdiag, diag, A, lda);
}
void symmetrize_lower(hipStream_t s, int n, double *A, int lda) {
  if (n <= 0) return;
  const int t = ceil_div(n, 32);
  hipLaunchKernelGGL(symmetrize_kernel, dim3(t, t), dim3(256), 0, s, n, A, lda);
}
void transpose_rows(hipStream_t s, int n, int m, const double *src, int lds, int r0, double *dst, int ldd) {
  if (n <= 0 || m <= 0) return;
  hipLaunchKernelGGL(transpose_rows_kernel, dim3(ceil_div(n, 32), ceil_div(m, 32)), dim3(256), 0, s, n, m, src,
                     lds, r0, dst, ldd);
}

void gather_columns(hipStream_t s, int m, int n, const double *src, int lds, const int *perm,
                    double *dst, int ldd) {
  if (m <= 0 || n <= 0) return;
  hipLaunchKernelGGL(gather_columns_kernel, grid2d(m, n), dim3(256), 0, s, m, n, src, lds, perm,
                     dst, ldd);
}

// the inverse map: dst(global) <- src(local piece of owner (me_r, me_c))
__global__ void bc_scatter_kernel(int mr, int nc, const double *__restrict__ src, int lds, int nb,
                                  int pr, int me_r, int pc, int me_c, double *__restrict__ dst, int ldd) {
  const int lr = blockIdx.x * blockDim.x + threadIdx.x;
  if (lr >= mr) return;
  const size_t gr = (size_t)((lr / nb) * pr + me_r) * nb + lr % nb;
  for (int lc = blockIdx.y; lc < nc; lc += gridDim.y) {
    const size_t gc = (size_t)((lc / nb) * pc + me_c) * nb + lc % nb;
    dst[gr + gc * ldd] = src[(size_t)lr + (size_t)lc * lds];
  }
}

void scatter_block_cyclic(hipStream_t s, int mr, int nc, const double *src, int lds, int nb, int pr,
                          int me_r, int pc, int me_c, double *dst, int ldd) {
  if (mr <= 0 || nc <= 0) return;
  hipLaunchKernelGGL(bc_scatter_kernel, grid2d(mr, nc), dim3(256), 0, s, mr, nc, src, lds, nb, pr, me_r,
                     pc, me_c, dst, ldd);
}

void gather_block_cyclic(hipStream_t s, int mr, int nc, const double *src, int lds, int nb, int pr,
                         int me_r, int pc, int me_c, double *dst, int ldd) {
  if (mr <= 0 || nc <= 0) return;
  hipLaunchKernelGGL(bc_gather_kernel, grid2d(mr, nc), dim3(256), 0, s, mr, nc, src, lds, nb, pr, me_r,
                     pc, me_c, dst, ldd);
}

void synth_matrix(hipStream_t s, int n, unsigned long long seed, double *M, int ldm) {
  if (n <= 0) return;
  hipLaunchKernelGGL(synth_kernel, grid2d(n, n), dim3(256), 0, s, n, seed, 1.0 / sqrt((double)n), M, ldm);
}

// partial: >= 512 doubles (device); the caller reduces the 256 partial maxima on the host ([256 ..): the maxima over
// the entries more than `bw` below the diagonal)
void maxabs_lower(hipStream_t s, int n, const double *A, int lda, double *partial, int bw) {
  hipLaunchKernelGGL(maxabs_lower_kernel, dim3(256), dim3(1024), 0, s, n, A, lda, partial, bw);
}
void scale_lower(hipStream_t s, int n, double alpha, double *A, int lda) {
  if (n <= 0) return;
  hipLaunchKernelGGL(scale_lower_kernel, grid2d(n, n), dim3(256), 0, s, n, alpha, A, lda);
}
void scale_vector(hipStream_t s, int n, double alpha, double *x) {
  if (n <= 0) return;
  hipLaunchKernelGGL(scale_vector_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, s, n, alpha, x);
}

// ------------------------------------------------------------------------------ kernel timing
namespace {
struct KProf {
  bool on = false;
  std::vector<hipEvent_t> ev;
  std::vector<int> ids;          // id of every recorded (begin, end) pair
  size_t used = 0;
  double sec[kProfCount] = {0, 0, 0, 0};
  long long cnt[kProfCount] = {0, 0, 0, 0};
} g_kprof;
}  // namespace

void kprof_enable(bool on) {
  g_kprof.on = on; g_kprof.used = 0; g_kprof.ids.clear();
  for (int i = 0; i < kProfCount; ++i) { g_kprof.sec[i] = 0.0; g_kprof.cnt[i] = 0; }
}
bool kprof_enabled() { return g_kprof.on; }
void kprof_begin(hipStream_t s, int id) {
  if (!g_kprof.on) return;
  if (g_kprof.used + 2 > g_kprof.ev.size()) {
    const size_t old = g_kprof.ev.size();
    g_kprof.ev.resize(old + 1024);
    for (size_t q = old; q < g_kprof.ev.size(); ++q) (void)hipEventCreate(&g_kprof.ev[q]);
  }
  g_kprof.ids.push_back(id);
  (void)hipEventRecord(g_kprof.ev[g_kprof.used], s);
}
void kprof_end(hipStream_t s, int id) {
  (void)id;
  if (!g_kprof.on) return;
  (void)hipEventRecord(g_kprof.ev[g_kprof.used + 1], s);
  g_kprof.used += 2;
}
void kprof_collect(double *seconds, long long *launches) {
  for (size_t q = 0; q + 1 < g_kprof.used; q += 2) {
    float ms = 0.f;
    const int id = g_kprof.ids[q / 2];
    if (hipEventElapsedTime(&ms, g_kprof.ev[q], g_kprof.ev[q + 1]) == hipSuccess) { g_kprof.sec[id] += ms * 1e-3; g_kprof.cnt[id] += 1; }
  }
  g_kprof.used = 0; g_kprof.ids.clear();
  for (int i = 0; i < kProfCount; ++i) {
    if (seconds) seconds[i] = g_kprof.sec[i];
    if (launches) launches[i] = g_kprof.cnt[i];
  }
}

}  // namespace ek
