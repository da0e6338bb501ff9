// sart_math.h — f64 reciprocal / reciprocal square root / square root for the gfx950 kernels of libsart
// (sart_kernels.hip, sart_emission.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace sart {

// 1/x, 1/sqrt(x) and sqrt(x) from the hardware seeds (v_rcp_f64 / v_rsq_f64, relative error e0 <= ~2^-23) plus ONE
// third-order step (error ~e0^3 < 2^-68, i.e. the result is the seed-independent f64 rounding, <= 1 ulp) instead of
// the scaling / fix-up sequence of the IEEE-exact expansions: the path never meets denormals or infinities here, and
// every consumer is tolerance-compared, never bit-compared.  tests/test_gpu_math.py measures the errors on the device.
__device__ __forceinline__ double frcp(double x) {
  const double r = __builtin_amdgcn_rcp(x);
  const double e = fma(-x, r, 1.0);           // 1/x = r / (1 - e) = r (1 + e + e^2 + ...)
  return fma(r, fma(e, e, e), r);
}
__device__ __forceinline__ double frsq(double x) {   // 1/sqrt(x), x > 0
  const double y = __builtin_amdgcn_rsq(x);
  const double e = fma(-(x * y), y, 1.0);     // 1/sqrt(x) = y / sqrt(1 - e) = y (1 + e/2 + 3 e^2 / 8 + ...)
  return fma(y, e * fma(0.375, e, 0.5), y);
}
// sqrt(x) for x > 0: Goldschmidt step from the seed + one Newton correction of the residual.  x < 0 gives NaN (every
// comparison downstream is then false: a miss); x == 0 also gives NaN (0 * inf) - use fsqrt() where an exact zero can occur.
__device__ __forceinline__ double fsqrt_pos(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y;
  const double h = 0.5 * y;
  g = fma(g, fma(-h, g, 0.5), g);             // error ~ 3/8 e0^2
  return fma(fma(-g, g, x), h, g);            // g + (x - g^2) / (2 sqrt x); h's own error only enters at e0^3
}
__device__ __forceinline__ double fsqrt(double x) {
  const double g = fsqrt_pos(x);
  return (x > 0.0) ? g : ((x == 0.0) ? 0.0 : __builtin_nan(""));   // rsq(0) = inf would give NaN; negative -> NaN
}

}  // namespace sart
