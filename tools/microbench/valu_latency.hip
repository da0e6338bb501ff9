// Dependent-chain latency of the VALU instruction classes of the ray kernel on gfx950: ONE serial chain per lane, 1 / 2 / 4
// waves per SIMD (256 / 512 / 1024-thread workgroups, one per CU).  Cycles per instruction per wave from s_memtime.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int KIND>
__global__ void chain_kernel(double* out, uint64_t* cycles, int iters, double c, double d) {
  double f = 1.0 + threadIdx.x * 1e-9;
  uint32_t a = threadIdx.x * 2654435761u + 1;
  const uint64_t t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 32; ++k) {
      if (KIND == 0) f = __builtin_fma(f, c, d);                 // v_fma_f64, serial
      else if (KIND == 1) f = f * c;                             // v_mul_f64
      else if (KIND == 2) f = f + d;                             // v_add_f64
      else if (KIND == 3) a = (a ^ 0x9e3779b9u) + 0x7f4a7c15u;   // two 32-bit ops
      else if (KIND == 4) { uint64_t p = (uint64_t)a * 0xD2511F53u; a = (uint32_t)(p >> 32) ^ (uint32_t)p; }   // mad_u64 + xor
      else if (KIND == 5) f = __builtin_amdgcn_rcp(f) + d;       // v_rcp_f64 + add
    }
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = f + a;
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int KIND>
void run(const char* name, int per_iter) {
  double* out; uint64_t* cyc;
  hipMalloc(&out, 256 * 1024 * 8); hipMalloc(&cyc, 256 * 8);
  for (int block : {256, 512, 1024}) {
    const int iters = 2048;
    hipLaunchKernelGGL(chain_kernel<KIND>, dim3(256), dim3(block), 0, 0, out, cyc, 16, 0.999999, 1e-7);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(chain_kernel<KIND>, dim3(256), dim3(block), 0, 0, out, cyc, iters, 0.999999, 1e-7);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    uint64_t h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double mean = 0; for (int i = 0; i < 256; ++i) mean += (double)h[i]; mean /= 256;
    printf("%-26s %d waves/SIMD: %.2f s_memtime ticks, %.2f ns per instruction per wave (kernel %.3f ms)\n", name, block / 256,
           mean / (iters * 32.0 * per_iter), ms * 1e6 / (iters * 32.0 * per_iter), ms);
  }
  hipFree(out); hipFree(cyc);
}

int main() {
  run<0>("v_fma_f64 chain", 1);
  run<1>("v_mul_f64 chain", 1);
  run<2>("v_add_f64 chain", 1);
  run<3>("v_xor + v_add chain", 2);
  run<4>("v_mad_u64_u32 + xor chain", 2);
  run<5>("v_rcp_f64 + add chain", 2);
  return 0;
}
