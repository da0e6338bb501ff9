// What bounds the record path (sart_trace_records: 208-byte Axion records into CALLER memory that is pageable - a Nim seq, a
// numpy array): rates of the ways 2 GiB can travel from the device into a pageable host buffer.
//   1 hipMemcpy D2H straight into the pageable buffer (what round 2 did)
//   2 D2H into a pinned staging buffer (the PCIe rate itself)
//   3 memcpy pinned -> pageable with 1 .. 16 host threads (first touch and second pass)
//   4 hipHostRegister of the pageable buffer + D2H straight into it + unregister
//   5 the pipeline: chunks D2H into two pinned buffers, host threads copy chunk k out while chunk k + 1 is in flight
// build: hipcc -O3 --offload-arch=gfx950 -o d2h_rates.bin d2h_rates.hip -lpthread
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

static void par_copy(char* dst, const char* src, size_t n, int threads) {
  std::vector<std::thread> th;
  const size_t per = (n / threads + 4095) & ~size_t(4095);
  for (int t = 0; t < threads; ++t) {
    const size_t lo = std::min(n, per * t), hi = std::min(n, per * (t + 1));
    if (hi > lo) th.emplace_back([=] { memcpy(dst + lo, src + lo, hi - lo); });
  }
  for (auto& t : th) t.join();
}

int main() {
  const size_t N = size_t(2) << 30;
  char* dev; CK(hipMalloc(&dev, N)); CK(hipMemset(dev, 1, N));
  hipStream_t s; CK(hipStreamCreate(&s));
  printf("threads available: %u\n", std::thread::hardware_concurrency());
  {
    char* host = (char*)malloc(N);
    double t0 = now(); CK(hipMemcpy(host, dev, N, hipMemcpyDeviceToHost)); double t1 = now();
    printf("1 hipMemcpy -> fresh pageable:      %6.1f GB/s\n", N / (t1 - t0) / 1e9);
    t0 = now(); CK(hipMemcpy(host, dev, N, hipMemcpyDeviceToHost)); t1 = now();
    printf("1 hipMemcpy -> touched pageable:    %6.1f GB/s\n", N / (t1 - t0) / 1e9);
    free(host);
  }
  char* pin; CK(hipHostMalloc(&pin, N, hipHostMallocDefault));
  {
    CK(hipMemcpyAsync(pin, dev, N, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s));
    double t0 = now(); CK(hipMemcpyAsync(pin, dev, N, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); double t1 = now();
    printf("2 D2H -> pinned:                    %6.1f GB/s\n", N / (t1 - t0) / 1e9);
  }
  for (int th : {1, 2, 4, 8, 12, 16}) {
    char* host = (char*)malloc(N);
    double t0 = now(); par_copy(host, pin, N, th); double t1 = now();
    double t2 = now(); par_copy(host, pin, N, th); double t3 = now();
    printf("3 memcpy pinned -> pageable, %2d thr: %6.1f GB/s first touch, %6.1f GB/s touched\n", th, N / (t1 - t0) / 1e9, N / (t3 - t2) / 1e9);
    free(host);
  }
  {
    char* host = (char*)malloc(N);
    double t0 = now(); CK(hipHostRegister(host, N, hipHostRegisterDefault)); double t1 = now();
    CK(hipMemcpyAsync(host, dev, N, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); double t2 = now();
    CK(hipHostUnregister(host)); double t3 = now();
    printf("4 register fresh %.3f s (%5.1f GB/s), D2H %5.1f GB/s, unregister %.3f s -> overall %5.1f GB/s\n", t1 - t0, N / (t1 - t0) / 1e9,
           N / (t2 - t1) / 1e9, t3 - t2, N / (t3 - t0) / 1e9);
    t0 = now(); CK(hipHostRegister(host, N, hipHostRegisterDefault)); t1 = now();
    CK(hipMemcpyAsync(host, dev, N, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); t2 = now();
    CK(hipHostUnregister(host)); t3 = now();
    printf("4 register touched %.3f s (%5.1f GB/s), D2H %5.1f GB/s, unregister %.3f s -> overall %5.1f GB/s\n", t1 - t0, N / (t1 - t0) / 1e9,
           N / (t2 - t1) / 1e9, t3 - t2, N / (t3 - t0) / 1e9);
    free(host);
  }
  for (size_t chunk : {size_t(64) << 20, size_t(208) << 20}) {
    for (int th : {4, 8, 12}) {
      for (int touched = 0; touched < 2; ++touched) {
        char* host = (char*)malloc(N);
        if (touched) memset(host, 0, N);
        hipEvent_t ev[2]; CK(hipEventCreate(&ev[0])); CK(hipEventCreate(&ev[1]));
        const size_t n_chunks = (N + chunk - 1) / chunk;
        double t0 = now();
        for (size_t k = 0; k <= n_chunks; ++k) {
          if (k < n_chunks) {
            const size_t off = k * chunk, len = std::min(chunk, N - off);
            CK(hipMemcpyAsync(pin + (k & 1) * chunk, dev + off, len, hipMemcpyDeviceToHost, s));
            CK(hipEventRecord(ev[k & 1], s));
          }
          if (k > 0) {   // chunk k - 1 has landed: copy it out while chunk k is in flight
            const size_t off = (k - 1) * chunk, len = std::min(chunk, N - off);
            CK(hipEventSynchronize(ev[(k - 1) & 1]));
            par_copy(host + off, pin + ((k - 1) & 1) * chunk, len, th);
          }
        }
        double t1 = now();
        printf("5 pipeline chunk %3zu MiB, %2d threads, %s pageable: %6.1f GB/s\n", chunk >> 20, th, touched ? "touched" : "fresh  ", N / (t1 - t0) / 1e9);
        free(host);
      }
    }
  }
  return 0;
}
