// Issue-rate microbenchmark for the instruction classes of the ray kernel on gfx950: independent chains per lane,
// 4 waves / SIMD (1024-thread workgroups, one per CU), cycles per wave-instruction from s_memtime.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int KIND>
__global__ __launch_bounds__(1024) void rate_kernel(uint64_t* out, uint64_t* cycles, int iters) {
  uint32_t a0 = threadIdx.x * 2654435761u + 1, a1 = a0 ^ 0x9e3779b9u, a2 = a0 + 77, a3 = a1 + 99;
  double f0 = 1.0 + threadIdx.x * 1e-9, f1 = 1.0000001, f2 = 0.9999999, f3 = 1.0000002;
  const uint64_t t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      if (KIND == 0) {  // v_mad_u64_u32, 4 independent chains
        uint64_t p0 = (uint64_t)a0 * 0xD2511F53u + a1, p1 = (uint64_t)a1 * 0xCD9E8D57u + a2;
        uint64_t p2 = (uint64_t)a2 * 0xD2511F53u + a3, p3 = (uint64_t)a3 * 0xCD9E8D57u + a0;
        a0 = (uint32_t)(p0 >> 32) ^ (uint32_t)p1; a1 = (uint32_t)(p1 >> 32) ^ (uint32_t)p2;
        a2 = (uint32_t)(p2 >> 32) ^ (uint32_t)p3; a3 = (uint32_t)(p3 >> 32) ^ (uint32_t)p0;
      } else if (KIND == 1) {  // v_fma_f64
        f0 = __builtin_fma(f0, f1, f2); f1 = __builtin_fma(f1, f2, f3); f2 = __builtin_fma(f2, f3, f0); f3 = __builtin_fma(f3, f0, f1);
      } else if (KIND == 2) {  // v_xor_b32 + v_add_u32
        a0 = (a0 ^ a1) + a2; a1 = (a1 ^ a2) + a3; a2 = (a2 ^ a3) + a0; a3 = (a3 ^ a0) + a1;
      } else if (KIND == 3) {  // v_mul_lo_u32
        a0 = a0 * a1; a1 = a1 * a2; a2 = a2 * a3; a3 = a3 * a0;
      } else if (KIND == 4) {  // v_mul_f64
        f0 = f0 * f1; f1 = f1 * f2; f2 = f2 * f3; f3 = f3 * f0;
      } else if (KIND == 5) {  // v_rcp_f64
        f0 = __builtin_amdgcn_rcp(f0); f1 = __builtin_amdgcn_rcp(f1); f2 = __builtin_amdgcn_rcp(f2); f3 = __builtin_amdgcn_rcp(f3);
      }
    }
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  out[blockIdx.x * 1024 + threadIdx.x] = a0 + a1 + a2 + a3 + (uint64_t)(f0 + f1 + f2 + f3);
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int KIND>
void run(const char* name, int per_iter) {
  uint64_t *out, *cyc;
  hipMalloc(&out, 256 * 1024 * 8); hipMalloc(&cyc, 256 * 8);
  const int iters = 4096;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(rate_kernel<KIND>, dim3(256), dim3(1024), 0, 0, out, cyc, 16);
  hipEventRecord(e0);
  hipLaunchKernelGGL(rate_kernel<KIND>, dim3(256), dim3(1024), 0, 0, out, cyc, iters);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  // per SIMD: 4 waves x iters x 16 x per_iter instructions
  const double inst_per_simd = 4.0 * iters * 16.0 * per_iter;
  printf("%-28s %8.3f ms  -> %.2f ns per wave-instruction per SIMD (%.1f cycles at 2.4 GHz)\n", name, ms, ms * 1e6 / inst_per_simd,
         ms * 1e6 / inst_per_simd * 2.4);
  hipFree(out); hipFree(cyc);
}

int main() {
  run<0>("v_mad_u64_u32 (+xor)", 8);   // 4 mad + 4 xor per step
  run<2>("v_xor_b32 + v_add_u32", 8);
  run<1>("v_fma_f64", 4);
  run<4>("v_mul_f64", 4);
  run<3>("v_mul_lo_u32", 4);
  run<5>("v_rcp_f64", 4);
  return 0;
}
