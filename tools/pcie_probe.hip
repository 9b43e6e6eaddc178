// tools/pcie_probe.hip -- what the host side of a GPU box delivers to the staging pipeline of ek_hip_solve (ek_solve.hip
// HostPipe): DMA rates of pinned and pageable memory in both directions, and the CPU's memcpy rate between pageable and
// pinned memory for 1..12 threads.   hipcc -O2 --offload-arch=gfx950 -o tools/pcie_probe tools/pcie_probe.hip -lpthread
#include <hip/hip_runtime.h>
#include <sched.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
  const size_t N = (size_t)1 << 30;                     // 1 GiB
  char *pin = nullptr, *pin2 = nullptr, *dev = nullptr;
  char *page = (char *)malloc(N), *page2 = (char *)malloc(N);
  if (hipHostMalloc((void **)&pin, N, hipHostMallocDefault) != hipSuccess) return 1;
  if (hipHostMalloc((void **)&pin2, N, hipHostMallocNonCoherent) != hipSuccess) pin2 = nullptr;
  if (hipMalloc((void **)&dev, N) != hipSuccess) return 1;
  memset(page, 1, N); memset(page2, 2, N); memset(pin, 3, N); if (pin2) memset(pin2, 4, N);
  hipStream_t s; (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  auto dma = [&](const char *what, void *dst, const void *src, hipMemcpyKind k) {
    (void)hipMemcpyAsync(dst, src, N, k, s); (void)hipStreamSynchronize(s);
    double best = 1e30;
    for (int r = 0; r < 3; ++r) { const double t0 = now(); (void)hipMemcpyAsync(dst, src, N, k, s); (void)hipStreamSynchronize(s); const double t = now() - t0; if (t < best) best = t; }
    printf("%-44s %6.1f GB/s\n", what, N / best / 1e9);
  };
  dma("DMA pinned (default)   -> device", dev, pin, hipMemcpyHostToDevice);
  dma("DMA device -> pinned (default)", pin, dev, hipMemcpyDeviceToHost);
  if (pin2) { dma("DMA pinned (non-coherent) -> device", dev, pin2, hipMemcpyHostToDevice); dma("DMA device -> pinned (non-coherent)", pin2, dev, hipMemcpyDeviceToHost); }
  dma("hipMemcpy pageable -> device", dev, page, hipMemcpyHostToDevice);
  dma("hipMemcpy device -> pageable", page, dev, hipMemcpyDeviceToHost);
  auto cpu = [&](const char *what, char *dst, const char *src) {
    for (int k : {1, 2, 4, 6, 8, 12}) {
      double best = 1e30;
      for (int r = 0; r < 2; ++r) {
        std::vector<std::thread> th;
        const double t0 = now();
        for (int i = 0; i < k; ++i) th.emplace_back([=]() { const size_t c = N / k; memcpy(dst + c * i, src + c * i, c); });
        for (auto &t : th) t.join();
        const double t = now() - t0; if (t < best) best = t;
      }
      printf("%-36s %2d threads %6.1f GB/s\n", what, k, N / best / 1e9);
    }
  };
  cpu("memcpy pageable -> pinned (default)", pin, page);
  cpu("memcpy pinned (default) -> pageable", page, pin);
  if (pin2) { cpu("memcpy pageable -> pinned (non-coh.)", pin2, page); cpu("memcpy pinned (non-coh.) -> pageable", page, pin2); }
  cpu("memcpy pageable -> pageable", page2, page);
  {
    hipStream_t s2; (void)hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    char *dev2 = nullptr; (void)hipMalloc((void **)&dev2, N);
    const double t0 = now();
    (void)hipMemcpyAsync(dev, pin, N, hipMemcpyHostToDevice, s);
    (void)hipMemcpyAsync(pin2 ? pin2 : page2, dev2, N, hipMemcpyDeviceToHost, s2);
    (void)hipStreamSynchronize(s); (void)hipStreamSynchronize(s2);
    printf("%-44s %6.1f GB/s each way\n", "DMA pinned both directions at once", N / (now() - t0) / 1e9);
  }
  cpu_set_t set; CPU_ZERO(&set);
  if (sched_getaffinity(0, sizeof(set), &set) == 0) printf("cores this process may run on: %d (hardware_concurrency %u)\n", CPU_COUNT(&set), std::thread::hardware_concurrency());
  return 0;
}
