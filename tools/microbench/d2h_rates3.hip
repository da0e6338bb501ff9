// Third look at the record path: can the caller's (pre-faulted, huge-page) buffer be pinned in place cheaply enough that the
// records DMA straight into it, instead of going through the runtime's staging copy (51-55 GB/s, CPU bound)?
// build: hipcc -O3 --offload-arch=gfx950 -o d2h_rates3.bin d2h_rates3.hip -lpthread
#include <hip/hip_runtime.h>
#include <sys/mman.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <typename F>
static void par(size_t n, int threads, F f) {
  std::vector<std::thread> th;
  const size_t per = (n / threads + 4095) & ~size_t(4095);
  for (int t = 0; t < threads; ++t) {
    const size_t lo = std::min(n, per * t), hi = std::min(n, per * (t + 1));
    if (hi > lo) th.emplace_back([=] { f(lo, hi); });
  }
  for (auto& t : th) t.join();
}

int main() {
  const size_t N = size_t(2) << 30;
  char* dev; CK(hipMalloc(&dev, N)); CK(hipMemset(dev, 1, N));
  hipStream_t s; CK(hipStreamCreate(&s));
  for (int huge = 0; huge < 2; ++huge) {
    for (int rep = 0; rep < 2; ++rep) {
      char* host = (char*)mmap(nullptr, N, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
      if (huge) madvise(host, N, MADV_HUGEPAGE);
      double t0 = now();
      par(N, 8, [=](size_t lo, size_t hi) { for (size_t p = lo; p < hi; p += 4096) host[p] = 0; });
      double t1 = now();
      CK(hipHostRegister(host, N, hipHostRegisterDefault));
      double t2 = now();
      CK(hipMemcpyAsync(host, dev, N, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s));
      double t3 = now();
      CK(hipHostUnregister(host));
      double t4 = now();
      printf("whole buffer, %s, fresh mapping #%d: touch %.1f GB/s, register %.3f s (%.1f GB/s), D2H %.1f GB/s, unregister %.3f s -> %.1f GB/s overall\n",
             huge ? "huge pages" : "4 KiB     ", rep, N / (t1 - t0) / 1e9, t2 - t1, N / (t2 - t1) / 1e9, N / (t3 - t2) / 1e9, t4 - t3, N / (t4 - t0) / 1e9);
      munmap(host, N);
    }
  }
  // chunked: touch + register chunk k + 1 on a helper thread while chunk k is copied
  for (size_t chunk : {size_t(208) << 20, size_t(64) << 20}) {
    char* host = (char*)mmap(nullptr, N, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    madvise(host, N, MADV_HUGEPAGE);
    const size_t n_chunks = (N + chunk - 1) / chunk;
    std::vector<int> ready(n_chunks, 0);
    double t_reg = 0;
    double t0 = now();
    std::thread helper([&] {
      for (size_t k = 0; k < n_chunks; ++k) {
        const size_t off = k * chunk, len = std::min(chunk, N - off);
        par(len, 8, [=](size_t lo, size_t hi) { for (size_t p = lo; p < hi; p += 4096) host[off + p] = 0; });
        double a = now();
        CK(hipHostRegister(host + off, len, hipHostRegisterDefault));
        t_reg += now() - a;
        __atomic_store_n(&ready[k], 1, __ATOMIC_RELEASE);
      }
    });
    for (size_t k = 0; k < n_chunks; ++k) {
      while (!__atomic_load_n(&ready[k], __ATOMIC_ACQUIRE)) std::this_thread::yield();
      const size_t off = k * chunk, len = std::min(chunk, N - off);
      CK(hipMemcpyAsync(host + off, dev + off, len, hipMemcpyDeviceToHost, s));
    }
    CK(hipStreamSynchronize(s));
    double t1 = now();
    helper.join();
    for (size_t k = 0; k < n_chunks; ++k) CK(hipHostUnregister(host + k * chunk));
    double t2 = now();
    printf("chunks of %3zu MiB, touched + registered one ahead: %.1f GB/s to the last byte, %.1f GB/s with the unregistering (register %.3f s in all)\n",
           chunk >> 20, N / (t1 - t0) / 1e9, N / (t2 - t0) / 1e9, t_reg);
    munmap(host, N);
  }
  return 0;
}
