// Throughput of scattered no-return atomics on gfx950: every lane adds to a pseudo-random element of a 65536-element image
// (the focal-plane image of the ray kernel), 256 workgroups x 1024 threads.  f64 / u64 / f32 / u32 adds, and f64 into 8 replicas.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int KIND>
__global__ __launch_bounds__(1024) void atomics_kernel(void* buf, int iters, uint32_t mask, uint32_t rep_mask) {
  uint32_t x = (blockIdx.x * 1024u + threadIdx.x) * 2654435761u + 12345u;
  const uint32_t rep = ((blockIdx.x * 16u + (threadIdx.x >> 6)) & rep_mask) * 65536u;
  for (int i = 0; i < iters; ++i) {
    x = x * 1664525u + 1013904223u;
    const uint32_t p = ((x >> 8) & mask) + rep;
    if (KIND == 0) unsafeAtomicAdd((double*)buf + p, 1.0);
    else if (KIND == 1) __hip_atomic_fetch_add((unsigned long long*)buf + p, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if (KIND == 2) unsafeAtomicAdd((float*)buf + p, 1.0f);
    else if (KIND == 3) __hip_atomic_fetch_add((uint32_t*)buf + p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if (KIND == 4) ((volatile double*)buf)[p] = 1.0;   // plain scattered store
    else if (KIND == 5) __hip_atomic_fetch_add((unsigned long long*)buf + p, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
}

template <int KIND>
void run(const char* name, uint32_t rep_mask) {
  void* buf; hipMalloc(&buf, 128ull * 65536 * 8); hipMemset(buf, 0, 128ull * 65536 * 8);
  const int iters = 256;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(atomics_kernel<KIND>, dim3(256), dim3(1024), 0, 0, buf, 8, 65535u, rep_mask);
  hipEventRecord(e0);
  hipLaunchKernelGGL(atomics_kernel<KIND>, dim3(256), dim3(1024), 0, 0, buf, iters, 65535u, rep_mask);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double n = 256.0 * 1024 * iters;
  printf("%-34s replicas %3u: %8.3f ms  %.3g lane-ops/s  (%.1f cycles per lane-op per CU at 2.1 GHz)\n", name, rep_mask + 1, ms, n / (ms * 1e-3),
         ms * 1e-3 * 2.1e9 / (n / 256.0));
  hipFree(buf);
}

int main() {
  for (uint32_t r : {0u, 7u, 63u}) {
    run<0>("f64 add (global_atomic_add_f64)", r);
    run<1>("u64 add agent scope", r);
    run<5>("u64 add workgroup scope", r);
    run<2>("f32 add", r);
    run<3>("u32 add agent scope", r);
    run<4>("plain 8-byte store", r);
  }
  return 0;
}
