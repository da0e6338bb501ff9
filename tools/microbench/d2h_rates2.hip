// Second look at the record path (see d2h_rates.hip): the destination's first touch is what costs (10-15 GB/s however the bytes
// arrive, against 51-57 GB/s into touched or pinned memory).  What makes the first touch faster, and where does the time of the
// chunked pipeline go?
// build: hipcc -O3 --offload-arch=gfx950 -o d2h_rates2.bin d2h_rates2.hip -lpthread
#include <hip/hip_runtime.h>
#include <sys/mman.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <typename F>
static void par(size_t n, int threads, F f) {   // f(lo, hi) on page-aligned slices
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
  {
    FILE* f = fopen("/sys/kernel/mm/transparent_hugepage/enabled", "r");
    char buf[128] = "?";
    if (f) { if (!fgets(buf, sizeof buf, f)) buf[0] = 0; fclose(f); }
    printf("transparent_hugepage/enabled: %s", buf);
  }
  char* dev; CK(hipMalloc(&dev, N)); CK(hipMemset(dev, 1, N));
  hipStream_t s; CK(hipStreamCreate(&s));
  char* pin; CK(hipHostMalloc(&pin, N, hipHostMallocDefault));
  CK(hipMemcpyAsync(pin, dev, N, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s));
  // ---- first touch of a fresh anonymous mapping --------------------------------------------------------------------------
  for (int huge = 0; huge < 2; ++huge) {
    for (int th : {1, 4, 8, 16}) {
      char* host = (char*)mmap(nullptr, N, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
      if (huge) madvise(host, N, MADV_HUGEPAGE);
      double t0 = now();
      par(N, th, [=](size_t lo, size_t hi) { for (size_t p = lo; p < hi; p += 4096) host[p] = 1; });   // one write per page
      double t1 = now();
      par(N, th, [=](size_t lo, size_t hi) { memcpy(host + lo, pin + lo, hi - lo); });
      double t2 = now();
      printf("first touch %s, %2d threads: fault %6.1f GB/s, then memcpy pinned -> it %6.1f GB/s\n", huge ? "MADV_HUGEPAGE" : "4 KiB pages  ", th,
             N / (t1 - t0) / 1e9, N / (t2 - t1) / 1e9);
      munmap(host, N);
    }
  }
  for (int th : {1, 4, 16}) {
    char* host = (char*)mmap(nullptr, N, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    int rc_all = 0;
    double t0 = now();
    par(N, th, [&](size_t lo, size_t hi) { if (madvise(host + lo, hi - lo, MADV_POPULATE_WRITE) != 0) rc_all = 1; });
    double t1 = now();
    printf("MADV_POPULATE_WRITE, %2d threads: %6.1f GB/s%s\n", th, N / (t1 - t0) / 1e9, rc_all ? " (madvise failed)" : "");
    munmap(host, N);
  }
  // ---- hipMemcpy into a destination prepared in different ways -----------------------------------------------------------
  for (int mode = 0; mode < 3; ++mode) {
    char* host = (char*)mmap(nullptr, N, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    double t0 = now();
    if (mode >= 1) madvise(host, N, MADV_HUGEPAGE);
    if (mode == 2) par(N, 16, [=](size_t lo, size_t hi) { for (size_t p = lo; p < hi; p += 4096) host[p] = 0; });
    double t1 = now();
    CK(hipMemcpy(host, dev, N, hipMemcpyDeviceToHost));
    double t2 = now();
    printf("hipMemcpy D2H into fresh mapping, %s: prepare %.3f s, copy %6.1f GB/s, overall %6.1f GB/s\n",
           mode == 0 ? "as it is             " : mode == 1 ? "MADV_HUGEPAGE        " : "MADV_HUGEPAGE + touch", t1 - t0, N / (t2 - t1) / 1e9, N / (t2 - t0) / 1e9);
    munmap(host, N);
  }
  // ---- the chunked pipeline, phases timed ------------------------------------------------------------------------------------
  for (size_t chunk : {size_t(64) << 20, size_t(256) << 20}) {
    char* host = (char*)mmap(nullptr, N, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    madvise(host, N, MADV_HUGEPAGE);
    memset(host, 0, N);
    hipEvent_t ev[2]; CK(hipEventCreate(&ev[0])); CK(hipEventCreate(&ev[1]));
    const size_t n_chunks = (N + chunk - 1) / chunk;
    double t_issue = 0, t_wait = 0, t_copy = 0;
    double t0 = now();
    for (size_t k = 0; k <= n_chunks; ++k) {
      double a = now();
      if (k < n_chunks) {
        const size_t off = k * chunk, len = std::min(chunk, N - off);
        CK(hipMemcpyAsync(pin + (k & 1) * chunk, dev + off, len, hipMemcpyDeviceToHost, s));
        CK(hipEventRecord(ev[k & 1], s));
      }
      double b = now();
      t_issue += b - a;
      if (k > 0) {
        const size_t off = (k - 1) * chunk, len = std::min(chunk, N - off);
        CK(hipEventSynchronize(ev[(k - 1) & 1]));
        double c = now();
        t_wait += c - b;
        const char* src = pin + ((k - 1) & 1) * chunk;
        par(len, 8, [=](size_t lo, size_t hi) { memcpy(host + off + lo, src + lo, hi - lo); });
        t_copy += now() - c;
      }
    }
    double t1 = now();
    printf("pipeline chunk %3zu MiB into touched huge pages: %6.1f GB/s (issue %.3f s, wait for D2H %.3f s, copy out %.3f s of %.3f s)\n", chunk >> 20,
           N / (t1 - t0) / 1e9, t_issue, t_wait, t_copy, t1 - t0);
    munmap(host, N);
  }
  return 0;
}
