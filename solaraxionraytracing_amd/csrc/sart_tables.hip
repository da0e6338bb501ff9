// sart_tables.hip — the solar sampling tables built ON THE DEVICE from an emission-rate table that is already there
// (sart_set_solar_tables_device): fluxRadiusCDF, diffFluxCDFs (initFullSetup, reference src/raytracer.nim:2670-2705) and
// the two guide tables in front of them (sart_device.h: kRadiusGuide, kEnergyGuide*), without the device -> host cumsum ->
// guide build -> device round trip of sart_set_solar_tables.
//
// Results are BIT-IDENTICAL to the host path (sart_host_build_cdfs + the guide construction of sart_set_solar_tables):
// the reference's order of the floating-point operations is kept - one lane walks one radius row from the first energy to
// the last - and every product, sum and quotient is a separately rounded IEEE operation: floating-point contraction is
// switched off for this translation unit (HIP's __dmul_rn / __dadd_rn are plain operators, which clang's default
// -ffp-contract=fast fuses into FMAs - one rounding where the reference, and the host path, have two).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sart_device.h"

#pragma clang fp contract(off)

namespace sart {

// ---- diffFluxCDFs: one lane per radius row (:2684-2703) ------------------------------------------------------------------
//   diffFlux[iE]      = emRate[iE] * (E * E) * radius * radius           (left to right, :2690)
//   radiusCumSum[iE]  = diffSum += diffFlux[iE]                          (:2691-2692)
//   row_sum[iRad]     = diffSum                                          (feeds diffRadiusSum, :2701)
//   cdf[iRad][iE]     = radiusCumSum[iE] / radiusCumSum[^1]              (toCdf, :2675-2677)
// Rows are written with the library's stride (n_energies + kEnergyCdfPad, pad = 1.0).  A wave's 64 lanes read 64 rows at
// the same column: every 128-byte line a lane touches serves its next 16 iterations out of the vector L1 / L2.
__global__ __launch_bounds__(64) void cdf_rows_kernel(const double* __restrict__ em, const double* __restrict__ radii,
                                                      const double* __restrict__ energies, int n_radii, int n_energies,
                                                      double* __restrict__ cdf, double* __restrict__ row_sum,
                                                      uint32_t* __restrict__ status) {
  const int r = blockIdx.x * 64 + threadIdx.x;
  if (r >= n_radii) return;
  const size_t stride = (size_t)n_energies + kEnergyCdfPad;
  const double* src = em + (size_t)r * n_energies;
  double* dst = cdf + (size_t)r * stride;
  const double radius = radii[r];
  double sum = 0.0;
  bool monotone = true;
  for (int i = 0; i < n_energies; ++i) {
    const double e = energies[i];
    const double flux = __dmul_rn(__dmul_rn(__dmul_rn(src[i], __dmul_rn(e, e)), radius), radius);
    const double next = __dadd_rn(sum, flux);
    monotone = monotone && (next >= sum);   // false for a negative or NaN emission rate
    sum = next;
    dst[i] = sum;
  }
  row_sum[r] = sum;
  for (int i = 0; i < n_energies; ++i) dst[i] = __ddiv_rn(dst[i], sum);
  for (int i = 0; i < kEnergyCdfPad; ++i) dst[n_energies + i] = 1.0;
  // what sart_set_solar_tables checks on the host: every row non-decreasing and ending at exactly 1.0 (x / x; fails for a row
  // that sums to 0, infinity or NaN)
  if (!monotone || !(dst[n_energies - 1] == 1.0)) atomicOr(status, 1u);
}

// ---- fluxRadiusCDF: the running sum of the row sums, in row order (:2701-2702, :2705) ------------------------------------
// 2048 dependent additions at most: one lane; the divisions are spread over the workgroup.
__global__ __launch_bounds__(256) void radius_cdf_kernel(const double* __restrict__ row_sum, int n_radii, double* __restrict__ rcdf,
                                                         uint32_t* __restrict__ status) {
  __shared__ double total;
  if (threadIdx.x == 0) {
    double s = 0.0;
    bool monotone = true;
    for (int r = 0; r < n_radii; ++r) {
      const double next = __dadd_rn(s, row_sum[r]);
      monotone = monotone && (next >= s);
      s = next;
      rcdf[r] = s;
    }
    total = s;
    if (!monotone || !(__ddiv_rn(s, s) == 1.0)) atomicOr(status, 2u);
  }
  __syncthreads();
  const double t = total;
  for (int r = threadIdx.x; r < n_radii; r += 256) rcdf[r] = __ddiv_rn(rcdf[r], t);
}

__device__ __forceinline__ int lower_bound_dev(const double* a, int n, double key) {   // first i with a[i] >= key, or n
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (a[mid] < key) lo = mid + 1; else hi = mid;
  }
  return lo;
}
__device__ __forceinline__ int upper_bound_dev(const double* a, int n, double key) {   // first i with a[i] > key, or n
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (!(key < a[mid])) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// ---- radius guide (sart_device.h): g[k] = min(lowerBound(rcdf, k / 2048), nR - 1), k = 0 .. 2048, then
//      g[2049 + j] = min(lowerBound(rcdf, 31/32 + j / 32768), nR - 1), j = 0 .. 1024; status[1] = the widest bracket a draw can meet
__global__ __launch_bounds__(256) void radius_guide_kernel(const double* __restrict__ rcdf, int n_radii, uint16_t* __restrict__ guide,
                                                           uint32_t* __restrict__ status) {
  __shared__ uint16_t g[kRadiusGuideEntries];
  for (int k = threadIdx.x; k < kRadiusGuideEntries; k += 256) {
    const double edge = k <= kRadiusGuide ? (double)k / (double)kRadiusGuide
                                          : 0.96875 + (double)(k - (kRadiusGuide + 1)) / 32768.0;   // exact: multiples of 2^-15
    const int i = min(lower_bound_dev(rcdf, n_radii, edge), n_radii - 1);
    g[k] = (uint16_t)i;
    guide[k] = (uint16_t)i;
  }
  __syncthreads();
  uint32_t span = 0;
  for (int k = threadIdx.x; k < kRadiusGuideEntries - 1; k += 256) {
    const bool used = k < kRadiusGuide * 31 / 32 || k > kRadiusGuide;   // buckets of u < 31/32, and the fine ones above
    if (used) span = max(span, (uint32_t)((int)g[k + 1] - (int)g[k]));
  }
  atomicMax(status + 1, span);
}

// ---- energy guide: one thread per (row, entry); entry layout of sart_device.h / sart_set_solar_tables ---------------------
//   k <= Uniform:      lowerBound(row, k / Div)                          (Div = kEnergyGuideDiv, Uniform = Div * 31/32)
//   k = Uniform + j:   upperBound(row, 1 - decode(code0 - j + 1)),  j = 1 .. 1600   (decode(c) = the double whose high word is c << 14)
//   last entry:     n_energies - 1
__global__ __launch_bounds__(256) void energy_guide_kernel(const double* __restrict__ cdf, int n_radii, int n_energies,
                                                           uint16_t* __restrict__ guide) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  const int r = blockIdx.y;
  if (k >= kEnergyGuideEntries) return;
  const double* row = cdf + (size_t)r * ((size_t)n_energies + kEnergyCdfPad);
  const int last = n_energies - 1;
  int v;
  if (k <= kEnergyGuideUniform) {
    v = min(lower_bound_dev(row, n_energies, (double)k / (double)kEnergyGuideDiv), last);
  } else if (k < kEnergyGuideBuckets) {
    const uint32_t j = (uint32_t)(k - kEnergyGuideUniform);
    const uint64_t bits = (uint64_t)(kEnergyGuideCode0 - j + 1u) << (14 + 32);
    const double edge = __dsub_rn(1.0, __longlong_as_double((long long)bits));   // exact: the subtrahend is <= 1/32 with six mantissa bits
    v = min(upper_bound_dev(row, n_energies, edge), last);
  } else {
    v = last;
  }
  guide[(size_t)r * kEnergyGuideEntries + k] = (uint16_t)v;
}

// cdf_hi32[i] = min(floor(cdf[i] 2^52) >> 20, 2^32 - 1) for every entry of the padded table (HotB::cdf_hi32)
__global__ __launch_bounds__(256) void cdf_hi32_kernel(const double* __restrict__ cdf, uint32_t* __restrict__ out, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const unsigned long long t = __double2ull_rd(cdf[i] * 4503599627370496.0) >> 20;   // the product is exact: cdf <= 1
  out[i] = t > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)t;
}
void launch_cdf_hi32(const double* cdf_dev, uint32_t* out_dev, size_t n, hipStream_t stream) {
  hipLaunchKernelGGL(cdf_hi32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, cdf_dev, out_dev, n);
}

// status[0]: bit 0 = a diffFluxCDFs row is not a CDF, bit 1 = fluxRadiusCDF is not; status[1] = radius_span
void launch_build_solar_tables(const double* em_dev, const double* radii_dev, const double* energies_dev, int n_radii, int n_energies,
                               double* cdf_dev, double* row_sum_dev, double* rcdf_dev, uint16_t* rguide_dev, uint16_t* eguide_dev,
                               uint32_t* status_dev, hipStream_t stream) {
  (void)hipMemsetAsync(status_dev, 0, 2 * sizeof(uint32_t), stream);
  hipLaunchKernelGGL(cdf_rows_kernel, dim3((n_radii + 63) / 64), dim3(64), 0, stream, em_dev, radii_dev, energies_dev, n_radii,
                     n_energies, cdf_dev, row_sum_dev, status_dev);
  hipLaunchKernelGGL(radius_cdf_kernel, dim3(1), dim3(256), 0, stream, row_sum_dev, n_radii, rcdf_dev, status_dev);
  hipLaunchKernelGGL(radius_guide_kernel, dim3(1), dim3(256), 0, stream, rcdf_dev, n_radii, rguide_dev, status_dev);
  hipLaunchKernelGGL(energy_guide_kernel, dim3((kEnergyGuideEntries + 255) / 256, n_radii), dim3(256), 0, stream, cdf_dev, n_radii,
                     n_energies, eguide_dev);
}

}  // namespace sart
