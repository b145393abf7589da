// pcie_probe.hip -- what the host->device link of this box delivers, to judge the host CSR ingress against
// (csrc/ingress.hpp): pinned H2D on 1/2/4 streams, chunk sizes, and pageable->pinned memcpy with T threads.
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  const size_t total = (size_t)1280 << 20;  // 1.25 GiB like the 100^3 matrix
  char *pin = nullptr, *dev = nullptr;
  CK(hipHostMalloc((void **)&pin, total, hipHostMallocDefault));
  CK(hipMalloc((void **)&dev, total));
  memset(pin, 1, total);
  std::vector<char> page(total, 2);
  hipStream_t st[4];
  for (int i = 0; i < 4; ++i) CK(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking));
  for (int ns : {1, 2, 4})
    for (size_t chunk : {(size_t)4 << 20, (size_t)12 << 20, (size_t)64 << 20, total}) {
      for (int rep = 0; rep < 2; ++rep) {
        CK(hipDeviceSynchronize());
        const double t0 = now();
        size_t k = 0;
        for (size_t o = 0; o < total; o += chunk, ++k)
          CK(hipMemcpyAsync(dev + o, pin + o, std::min(chunk, total - o), hipMemcpyHostToDevice, st[k % ns]));
        CK(hipDeviceSynchronize());
        const double t = now() - t0;
        if (rep) printf("pinned H2D  streams %d  chunk %4zu MiB : %6.2f ms  %6.2f GB/s\n", ns, chunk >> 20, t * 1e3, total / t * 1e-9);
      }
    }
  {
    CK(hipDeviceSynchronize());
    const double t0 = now();
    CK(hipMemcpy(dev, page.data(), total, hipMemcpyHostToDevice));
    const double t = now() - t0;
    printf("pageable hipMemcpy (what round 2 did): %6.2f ms  %6.2f GB/s\n", t * 1e3, total / t * 1e-9);
  }
  for (int T : {1, 2, 4, 8, 12, 16}) {
    for (int rep = 0; rep < 2; ++rep) {
      const double t0 = now();
      std::vector<std::thread> th;
      for (int t = 0; t < T; ++t)
        th.emplace_back([&, t] {
          const size_t lo = total / T * t, hi = t == T - 1 ? total : total / T * (t + 1);
          memcpy(pin + lo, page.data() + lo, hi - lo);
        });
      for (auto &x : th) x.join();
      const double t = now() - t0;
      if (rep) printf("pageable -> pinned memcpy, %2d threads : %6.2f ms  %6.2f GB/s\n", T, t * 1e3, total / t * 1e-9);
    }
  }
  // the link while staging threads copy pageable -> pinned next to it (what the ingress pipeline does)
  for (int T : {0, 4, 8}) {
    std::vector<char> page2(total, 3);
    char *pin2 = nullptr;
    CK(hipHostMalloc((void **)&pin2, total, hipHostMallocDefault));
    std::atomic<int> stop(0);
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t)
      th.emplace_back([&, t] {
        const size_t lo = total / T * t, hi = t == T - 1 ? total : total / T * (t + 1);
        while (!stop.load()) memcpy(pin2 + lo, page2.data() + lo, hi - lo);
      });
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipDeviceSynchronize());
      const double t0 = now();
      for (size_t o = 0; o < total; o += (size_t)48 << 20) CK(hipMemcpyAsync(dev + o, pin + o, std::min((size_t)48 << 20, total - o), hipMemcpyHostToDevice, st[0]));
      CK(hipDeviceSynchronize());
      const double t = now() - t0;
      if (rep) printf("pinned H2D 48 MiB copies beside %d memcpy threads: %6.2f ms  %6.2f GB/s\n", T, t * 1e3, total / t * 1e-9);
    }
    stop.store(1);
    for (auto &x : th) x.join();
    CK(hipHostFree(pin2));
  }
  printf("hardware_concurrency %u\n", std::thread::hardware_concurrency());
  return 0;
}
