// Energy per operation on gfx950 under the socket power cap: every instruction class of the ray kernel runs alone for a few
// seconds on all 256 CUs (1024-thread workgroups, 4 waves / SIMD) while rocm-smi reports socket power and shader clock.
//   energy per wave-instruction = (P - P_idle) / (wave-instructions per second)
// P_idle = the same grid spinning in s_sleep.  Used for the energy budget of the ray kernel in DESIGN.md.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <unistd.h>
#include <string>

template <int KIND>
__global__ __launch_bounds__(1024) void burn(double* out, const double* table, uint32_t table_mask, int iters) {
  __shared__ double lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 1024) lds[i] = 1.0 + i * 1e-9;
  __syncthreads();
  uint32_t a0 = threadIdx.x * 2654435761u + blockIdx.x + 1, a1 = a0 ^ 0x9e3779b9u, a2 = a0 + 77, a3 = a1 + 99;
  double f0 = 1.0 + threadIdx.x * 1e-9, f1 = 1.0000001, f2 = 0.9999999, f3 = 1.0000002;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      if (KIND == 0) {          // idle: the waves sleep
        __builtin_amdgcn_s_sleep(64);
      } else if (KIND == 1) {   // v_fma_f64 x 4
        f0 = __builtin_fma(f0, f1, f2); f1 = __builtin_fma(f1, f2, f3); f2 = __builtin_fma(f2, f3, f0); f3 = __builtin_fma(f3, f0, f1);
      } else if (KIND == 2) {   // v_mad_u64_u32 x 4 (+ 4 xor)
        uint64_t p0 = (uint64_t)a0 * 0xD2511F53u + a1, p1 = (uint64_t)a1 * 0xCD9E8D57u + a2;
        uint64_t p2 = (uint64_t)a2 * 0xD2511F53u + a3, p3 = (uint64_t)a3 * 0xCD9E8D57u + a0;
        a0 = (uint32_t)(p0 >> 32) ^ (uint32_t)p1; a1 = (uint32_t)(p1 >> 32) ^ (uint32_t)p2;
        a2 = (uint32_t)(p2 >> 32) ^ (uint32_t)p3; a3 = (uint32_t)(p3 >> 32) ^ (uint32_t)p0;
      } else if (KIND == 3) {   // v_xor_b32 + v_add_u32 x 4
        a0 = (a0 ^ a1) + a2; a1 = (a1 ^ a2) + a3; a2 = (a2 ^ a3) + a0; a3 = (a3 ^ a0) + a1;
      } else if (KIND == 4) {   // ds_read_b64 gather (random rows) x 2 + the address arithmetic
        a0 = a0 * 1664525u + 1013904223u;
        f0 += lds[(a0 >> 9) & 8191u]; f1 += lds[(a0 >> 19) & 8191u];
      } else if (KIND == 5) {   // global 16-byte gather from a table that misses L2
        a0 = a0 * 1664525u + 1013904223u;
        const double2 v = *reinterpret_cast<const double2*>(table + 2u * ((a0 >> 6) & table_mask));
        f0 += v.x; f1 += v.y;
      } else if (KIND == 6) {   // v_mul_f64 x 4
        f0 = f0 * f1; f1 = f1 * f2; f2 = f2 * f3; f3 = f3 * f0;
      } else if (KIND == 7) {   // v_mul_hi_u32 + v_mul_lo_u32 x 4 (+ 4 xor): the two halves of a Philox product as two instructions
        uint32_t h0, l0, h1, l1, h2, l2, h3, l3;
        const uint32_t m0 = 0xD2511F53u, m1 = 0xCD9E8D57u;
        asm volatile("v_mul_hi_u32 %0, %2, %3\n\tv_mul_lo_u32 %1, %2, %3" : "=&v"(h0), "=&v"(l0) : "v"(a0), "s"(m0));
        asm volatile("v_mul_hi_u32 %0, %2, %3\n\tv_mul_lo_u32 %1, %2, %3" : "=&v"(h1), "=&v"(l1) : "v"(a1), "s"(m1));
        asm volatile("v_mul_hi_u32 %0, %2, %3\n\tv_mul_lo_u32 %1, %2, %3" : "=&v"(h2), "=&v"(l2) : "v"(a2), "s"(m0));
        asm volatile("v_mul_hi_u32 %0, %2, %3\n\tv_mul_lo_u32 %1, %2, %3" : "=&v"(h3), "=&v"(l3) : "v"(a3), "s"(m1));
        a0 = h0 ^ l1; a1 = h1 ^ l2; a2 = h2 ^ l3; a3 = h3 ^ l0;
      }
    }
  }
  out[blockIdx.x * 1024 + threadIdx.x] = f0 + f1 + f2 + f3 + (double)(a0 + a1 + a2 + a3);
}

static std::string smi() {
  FILE* p = popen("rocm-smi --showpower --showclocks 2>/dev/null | grep -E 'Package Power|sclk' | sed 's/.*: //' | tr '\\n' ' '", "r");
  char buf[512] = {0};
  if (p) { if (!fgets(buf, sizeof buf, p)) buf[0] = 0; pclose(p); }
  return std::string(buf);
}

template <int KIND>
void run(const char* name, double insts_per_inner, double* out, const double* table, uint32_t mask, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(burn<KIND>, dim3(256), dim3(1024), 0, 0, out, table, mask, 64);   // warm
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(burn<KIND>, dim3(256), dim3(1024), 0, 0, out, table, mask, iters);
  hipEventRecord(e1);
  usleep(2500000);
  const std::string s1 = smi();
  usleep(800000);
  const std::string s2 = smi();
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double wave_insts = 256.0 * 16.0 * (double)iters * 16.0 * insts_per_inner;   // wave-level instructions of the class
  printf("%-34s %8.1f ms  %.4g wave-inst/s  | smi: %s| %s\n", name, ms, wave_insts / (ms * 1e-3), s1.c_str(), s2.c_str());
  fflush(stdout);
}

int main() {
  double* out; hipMalloc(&out, 256 * 1024 * 8);
  const uint32_t n_table = 1u << 23;   // 2^23 x 16 B = 128 MB
  double* table; hipMalloc(&table, (size_t)n_table * 16); hipMemset(table, 0, (size_t)n_table * 16);
  run<0>("idle (s_sleep)", 1, out, table, n_table - 1, 160000);
  run<1>("v_fma_f64", 4, out, table, n_table - 1, 10000000);
  run<6>("v_mul_f64", 4, out, table, n_table - 1, 10000000);
  run<2>("v_mad_u64_u32 (+ as many v_xor)", 4, out, table, n_table - 1, 5000000);
  run<7>("v_mul_hi_u32 + v_mul_lo_u32 pairs (+ xor)", 4, out, table, n_table - 1, 5000000);
  run<3>("v_xor_b32 + v_add_u32", 8, out, table, n_table - 1, 10000000);
  run<4>("ds_read_b64 gather (x2 per step)", 2, out, table, n_table - 1, 3500000);
  run<5>("global 16 B gather, 128 MB table", 1, out, table, n_table - 1, 60000);
  run<0>("idle (s_sleep)", 1, out, table, n_table - 1, 160000);
  return 0;
}
