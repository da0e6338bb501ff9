// sart_kernels.hip — gfx950 (MI355X, CDNA4) kernels of the per-ray hot path.
//
// One ray per lane, f64 throughout (the Sun sits 1.5e14 mm away and the mirror quadratics cancel
// ~1e5-magnitude terms).  The algorithm is the reference's `traceAxion`
// (src/raytracer.nim:1736-2221), restructured for the device:
//   * every per-setup / per-shell / per-energy-index quantity is a table built on the host
//     (sart_api.hip: hoist_*), so the kernel has no setup-only transcendental;
//   * the straight line through bore and pipes is carried as (point on the bore exit, slopes
//     dx/dz, dy/dz) instead of differences of 1e11-magnitude points: same geometry, without the
//     reference's ~1e-5 mm cancellation noise;
//   * mirror hits use the cancellation-free form of the same quadratic roots; the reflection
//     v cos2a - (v x axis) sin2a (:778-779) is evaluated algebraically (no asin/sin/cos), and on a
//     hit the surface normals reduce to closed forms without a square root;
//   * a Philox4x32-7 counter block per ray (key = seed, counter = global ray id) replaces the
//     reference's shared xoroshiro stream, draw order as in the reference;
//   * the path is split where most rays die (bore, pipes, spider, glass fronts: ~2/3 of all rays for
//     BabyIAXO) into three stages that each run on full waves: A0 (one Philox block: rays that the
//     radius of their bore-exit point alone proves dead), A1 = phase A (sample + cuts + shell
//     selection), B = phase B (mirrors + weights + accumulation); survivors are compacted with a
//     wavefront ballot + prefix count into per-wave LDS rings between the stages;
//   * phase A's scalars travel in the kernel arguments (re-read with scalar loads at the start of
//     each pass); the radius CDF, its guide table, the shell table and the shell look-up table
//     are staged into LDS once per workgroup;
//   * results are accumulated on the device: f64 atomics into the focal-plane image, wave
//     reductions for the scalars.
// Lines cited as ":NNNN" refer to src/raytracer.nim of the reference.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <type_traits>

#include "../../include/sart.h"
#include "sart_device.h"
#include "sart_math.h"

namespace sart {

// ------------------------------------------------------------------------------------------------
// Philox4x32-7 (Salmon et al., SC'11: seven rounds are the fewest that pass BigCrush; ten are the paper's default with margin).
// Round 6 went from ten rounds and two blocks per ray to seven rounds and one block per ray (uniforms_of): the counter RNG was a
// quarter of the ray kernel's vector instructions, and v_mad_u64_u32 is not a full-rate one.  The stream is a free parameter of
// the path (the reference draws from xoroshiro128+; parity with it is statistical, tests/test_gpu_parity.py: independent streams).
// ------------------------------------------------------------------------------------------------
struct U4 { uint32_t x, y, z, w; };
constexpr int kPhiloxRounds = 7;

__device__ __forceinline__ U4 philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                         uint32_t k1) {
#pragma unroll
  for (int r = 0; r < kPhiloxRounds; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;   // one v_mad_u64_u32 yields both halves
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    // a ^ b ^ c in ONE instruction: gfx950's v_bitop3_b32 with the truth table of the three-input xor (0x96); the compiler
    // does not form it from two `^` by itself (it emits two v_xor_b32)
    const uint32_t n0 = __builtin_amdgcn_bitop3_b32(hi1, c1, k0, 0x96);
    const uint32_t n2 = __builtin_amdgcn_bitop3_b32(hi0, c3, k1, 0x96);
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return U4{c0, c1, c2, c3};
}

// Word n of the shared word stream behind the high word of u3: word (n & 3) of the Philox block with counter
// (n >> 2, 3, 0).  Four consecutive ray ids share one block (stage A0 computes it once per lane for four rays).
__device__ __forceinline__ U4 stream_block(uint64_t group, uint32_t k0, uint32_t k1) {
  return philox4x32((uint32_t)group, (uint32_t)(group >> 32), 3u, 0u, k0, k1);
}
__device__ __forceinline__ uint32_t word_of(const U4& b, uint32_t k) {
  return k == 0u ? b.x : (k == 1u ? b.y : (k == 2u ? b.z : b.w));
}

// Assembly comment at the head of a pipeline stage: tools/isa_histogram.py attributes the instructions of the compiler's
// listing to stages by these markers.  No instruction, no operands, no clobbers.
#define SART_STAGE_MARK(name) asm volatile("; SART_STAGE " name)

// Lane mask of a predicate.  HIP's __ballot() materialises the predicate as 0 / 1 and compares again (two VALU
// instructions); the builtin takes the compare's lane mask as it is.
__device__ __forceinline__ uint64_t ballot64(bool pred) { return __builtin_amdgcn_ballot_w64(pred); }

// A compile-time f64 constant in a scalar register pair.  LLVM otherwise materialises a polynomial coefficient with two
// v_mov_b32 into the destination of a two-address v_fmac_f64 (three vector instructions per Horner step instead of
// one); s_mov_b32 issues on the scalar unit beside other waves' vector instructions.  `volatile` keeps the moves next to
// their use (hoisted out of the persistent loop the registers would be spilled through VGPR lanes).
template <uint64_t BITS>
__device__ __forceinline__ double scalar_const() {
  uint32_t lo, hi;
  asm volatile("s_mov_b32 %0, %2\n\ts_mov_b32 %1, %3" : "=s"(lo), "=s"(hi) : "n"((uint32_t)BITS), "n"((uint32_t)(BITS >> 32)));
  return __longlong_as_double((long long)(((uint64_t)hi << 32) | (uint64_t)lo));
}
#define SC(x) (scalar_const<__builtin_bit_cast(uint64_t, (double)(x))>())

// One Horner step p x + C with the coefficient C as a scalar operand: v_fma_f64 v, v, v, s.  Left to itself LLVM selects the
// two-address v_fmac_f64 for a single-use addend and copies the constant into the destination first (two v_mov_b32 per
// step, also when the constant already sits in scalar registers).
__device__ __forceinline__ double fma_vvs(double p, double x, double c_scalar) {
  double r;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(p), "v"(x), "s"(c_scalar));
  return r;
}
#define HORNER(p, x, c) fma_vvs((p), (x), SC(c))

// Uniform in [0, 1) from two words (hi word first): 52 random mantissa bits under the exponent of 1.0,
// minus 1.0 — the construction of Nim's std/random rand(1.0) (and of the oracle).
__device__ __forceinline__ double u52(uint32_t hi, uint32_t lo) {
  // 0x3FF0000000000000 | ((hi:lo) >> 12) as two funnel shifts (v_alignbit_b32): low word = (hi:lo) >> 12, high word =
  // (0x3FF:hi) >> 12 = 0x3FF00000 | (hi >> 12)
  const uint32_t mlo = __builtin_amdgcn_alignbit(hi, lo, 12u);
  const uint32_t mhi = __builtin_amdgcn_alignbit(0x3FFu, hi, 12u);
  return __longlong_as_double((long long)(((uint64_t)mhi << 32) | (uint64_t)mlo)) - 1.0;
}

// ------------------------------------------------------------------------------------------------
// small math helpers
// ------------------------------------------------------------------------------------------------
// sin and cos of pi a, a = TURNS * u for u in [0, 1) (TURNS = 2: a full turn, 1: half a turn), to <= ~1.5 ulp:
// k = rint(64 a), r = a - k / 64 exactly (|r| <= 1/128), (C, S) = (cos, sin)(pi k / 64) from the 129-entry table in LDS
// (correctly rounded on the host, exact zeros and ones at the multiples of pi/2), short Taylor polynomials in r
// (truncation < 4e-17 relative for |pi r| <= pi/128) and the angle-addition formulas.  No quadrant selects, and 7
// polynomial coefficients (scalar registers, shared by the three evaluations of a ray) instead of 16.
struct SinCosCoef { double s3, s2, s1, s0, d3, d2, d1; };   // sin(pi r) = r (s0 + y (s1 + y (s2 + y s3))), cos(pi r) = 1 + y (d1 + ...)
__device__ __forceinline__ SinCosCoef sincos_coef() {
  return SinCosCoef{SC(-0.5992645293207921), SC(2.5501640398773455), SC(-5.16771278004997), SC(3.141592653589793),
                    SC(-1.3352627688545895), SC(4.0587121264167685), SC(-4.934802200544679)};
}
template <int TURNS>
__device__ __forceinline__ void sincos_turns(double u, const double* __restrict__ table, const SinCosCoef& K, double* sn, double* cs) {
  constexpr double T = (double)TURNS;
  const double kf = __builtin_rint(u * (64.0 * T));          // 0 .. 64 TURNS
  const double r = fma(-1.0 / 64.0, kf, u * T);              // T u - k / 64, exact (T is a power of two)
  const double2 t = reinterpret_cast<const double2*>(table)[(int)kf];   // one ds_read_b128: (C, S)
  const double x = r * r;
  const double sr = fma(fma(fma(K.s3, x, K.s2), x, K.s1), x, K.s0) * r;
  const double cr = fma(fma(fma(K.d3, x, K.d2), x, K.d1), x, 1.0);
  *sn = fma(t.x, sr, t.y * cr);                               // sin(a + b) = S cr + C sr
  *cs = fma(-t.y, sr, t.x * cr);                              // cos(a + b) = C cr - S sr
}

// asin for the grazing angles of the path (|x| < ~0.03): odd Taylor series
// asin x = x + x^3/6 + 3x^5/40 + 5x^7/112 + 35x^9/1152 + 63x^11/2816 + 231x^13/13312 + 143x^15/10240,
// truncation error < 1e-19 below 0.06; the library function outside.
// The series is evaluated unconditionally (straight-line code the scheduler can interleave with its surroundings); the
// library call is a divergent alternative that a wave skips when no lane needs it.
__device__ __forceinline__ double asin_small(double x, double x2) {   // x2 = x^2 (the caller has it)
  double p = 0.01396484375;                        // 143/10240
  p = HORNER(p, x2, 0.017352764423076924);         // 231/13312
  p = HORNER(p, x2, 0.022372159090909092);         // 63/2816
  p = HORNER(p, x2, 0.030381944444444444);         // 35/1152
  p = HORNER(p, x2, 0.044642857142857144);         // 5/112
  p = HORNER(p, x2, 0.075);                        // 3/40
  p = HORNER(p, x2, 0.16666666666666666);          // 1/6
  double r = fma(x * x2, p, x);
  if (!(fabs(x) < 0.06)) {
    asm volatile("; rare: asin outside the series' range");
    r = asin(x);
  }
  return r;
}

// atan for ray slopes (|x| < 0.05): alternating series to x^15; library outside.
__device__ __forceinline__ double atan_small(double x) {
  const double x2 = x * x;
  double p = -1.0 / 15.0;
  p = fma(p, x2, 1.0 / 13.0);
  p = fma(p, x2, -1.0 / 11.0);
  p = fma(p, x2, 1.0 / 9.0);
  p = fma(p, x2, -1.0 / 7.0);
  p = fma(p, x2, 1.0 / 5.0);
  p = fma(p, x2, -1.0 / 3.0);
  double r = fma(x * x2, p, x);
  if (!(fabs(x) < 0.05)) {
    asm volatile("; rare: atan outside the series' range");
    r = atan(x);
  }
  return r;
}

// cos on |x| <= 1 by its Taylor series to x^20 (error < 1e-19 there); library outside.
__device__ __forceinline__ double cos_small(double x) {
  const double x2 = x * x;
  double p = 4.110317623312165e-19;            //  1/20!
  p = fma(p, x2, -1.5619206968586225e-16);     // -1/18!
  p = fma(p, x2, 4.779477332387385e-14);       //  1/16!
  p = fma(p, x2, -1.1470745597729725e-11);     // -1/14!
  p = fma(p, x2, 2.08767569878681e-09);        //  1/12!
  p = fma(p, x2, -2.755731922398589e-07);      // -1/10!
  p = fma(p, x2, 2.48015873015873e-05);        //  1/8!
  p = fma(p, x2, -0.001388888888888889);       // -1/6!
  p = fma(p, x2, 0.041666666666666664);        //  1/4!
  p = fma(p, x2, -0.5);
  double r = fma(p, x2, 1.0);
  if (!(fabs(x) <= 1.0)) {
    asm volatile("; rare: cos outside the series' range");
    r = cos(x);
  }
  return r;
}

// cos(ya) for the yaw angle ya = -deg(atan(t)) of a ray with slope t, the degree value taken as radians - sic
// (:1598, :2101-2115): cos((180 / pi) atan t) as ONE even power series in t (composed at 60 digits, mpmath), valid for
// |t| < 0.006 (every ray from the Sun: |slope| < 4.7e-3), truncation error < 7e-18; the two-step evaluation outside.
__device__ __forceinline__ double cos_yaw_of_slope(double t) {
  const double x = t * t;
  double p = 2976939695167.2104;                  // t^12
  p = HORNER(p, x, -112889008417.68803);          // t^10
  p = HORNER(p, x, 2979383355.295581);            // t^8
  p = HORNER(p, x, -49735946.89381117);           // t^6
  p = HORNER(p, x, 450128.3326032301);            // t^4
  p = HORNER(p, x, -1641.403175005872);           // t^2 : -(180 / pi)^2 / 2
  double r = fma(p, x, 1.0);
  if (!(fabs(t) < 0.006)) {
    asm volatile("; rare: slope outside the composed series' range");
    r = cos_small(-atan_small(t) * 57.29577951308232);
  }
  return r;
}

// exp(x) for x <= 0 - the two exponentials of the gas stage, exp(-Gamma L / 2) and exp(-(mu_pipe d + mu_magnet L))
// (axionMassforMagnet.nim:75-113) - in 20 vector instructions instead of the library's ~40: n = rint(x log2 e),
// r = x - n ln 2 in two pieces (|r| <= 0.3466), Taylor series to r^13 (truncation 4e-18), one v_ldexp_f64.  <= 1.5 ulp
// (tests/test_gpu_math.py); x < -746 gives 0, like the library.
__device__ __forceinline__ double exp_neg(double x) {
  x = fmax(x, -746.0);
  const double n = __builtin_rint(x * 1.4426950408889634);
  double r = fma(n, -0.6931471805598903, x);         // ln 2, high part (low 11 bits zero: n * hi is exact)
  r = fma(n, -5.497923018708371e-14, r);             // ln 2, low part
  double p = 1.6059043836821613e-10;                 // 1/13!
  p = HORNER(p, r, 2.08767569878681e-09);            // 1/12!
  p = HORNER(p, r, 2.505210838544172e-08);           // 1/11!
  p = HORNER(p, r, 2.755731922398589e-07);           // 1/10!
  p = HORNER(p, r, 2.7557319223985893e-06);          // 1/9!
  p = HORNER(p, r, 2.48015873015873e-05);            // 1/8!
  p = HORNER(p, r, 0.0001984126984126984);           // 1/7!
  p = HORNER(p, r, 0.001388888888888889);            // 1/6!
  p = HORNER(p, r, 0.008333333333333333);            // 1/5!
  p = HORNER(p, r, 0.041666666666666664);            // 1/4!
  p = HORNER(p, r, 0.16666666666666666);             // 1/3!
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  return ldexp(p, (int)n);
}

// cos(y) for the phase q L of the gas-stage conversion probability (any finite y; thousands of radians far from the
// resonance): y / (2 pi) with 1 / (2 pi) in two pieces (the FMA keeps the product exact before the integer part is taken
// away, the second piece corrects it: the fraction of a turn is good to ~1e-16 for |y| < 1e15), then the table-based
// cosine of the sampling angles (sincos_turns: (cos, sin)(pi k / 64) from LDS + degree-3 polynomials).  Absolute error
// <= 2.5e-16 + 7e-16 |y| 2^-53 ... in practice <= 1.5 ulp(1); the library's Payne-Hanek path costs ~100 instructions.
__device__ __forceinline__ double cos_any(double y, const double* __restrict__ table, const SinCosCoef& K) {
  const double n = __builtin_rint(y * 0.15915494309189535);
  double f = fma(y, 0.15915494309189535, -n);        // 1 / (2 pi), high part
  f = fma(y, -9.839338337591243e-18, f);             // low part
  double sn, cs;
  sincos_turns<2>(fabs(f), table, K, &sn, &cs);      // cos is even; |f| <= 0.5 + 1e-16
  if (!(fabs(y) < 1e15)) {
    asm volatile("; rare: cos of a huge or non-finite phase");
    cs = cos(y);
  }
  return cs;
}

// Gas-stage conversion probability axionConversionProb2 (axionMassforMagnet.nim:75-98) of one ray for one axion mass:
//   P = (g B / 2)^2 / (q^2 + Gamma^2 / 4) (1 + exp(-Gamma L) - 2 exp(-Gamma L / 2) cos(q L)),   q = |m_gamma^2 - m_a^2| / (2 E)
// split into what depends on the ray alone (GasRay: built once per ray in phase B) and the mass (dm2_abs).  The histogram
// kernels apply it as the LAST factor of the weight, and the fused mass scan calls it once per mass on the same GasRay: every
// operation below is an explicit multiplication or FMA (nothing for the compiler to contract differently in different
// surroundings), so a scan's weight for mass k is bit for bit the weight of a single-mass launch with that mass.
struct GasRay {
  double inv_two_e;   // 1 / (2 E[eV])
  double g2q;         // Gamma^2 / 4
  double m2eh;        // -2 exp(-Gamma L / 2)
  double eh2p1;       // 1 + exp(-Gamma L)
  double Lnat;        // L in 1 / eV
};
// A value the compiler must keep in a vector register from here on (a polynomial's leading coefficient in front of a loop: with
// -disable-machine-licm nothing hoists the two v_mov_b32 that materialise an f64 constant out of a loop by itself).
__device__ __forceinline__ double in_vgpr(double x) {
  asm volatile("" : "+v"(x));
  return x;
}
// The leading coefficients of the two residual polynomials of the table-based cosine, in vector registers (GasCos::make() in
// front of the per-mass loop of the scan); the other five are scalar operands of the Horner steps.
struct GasCos {
  double s3, d3;
  static __device__ __forceinline__ GasCos make() { return GasCos{in_vgpr(SC(-0.5992645293207921)), in_vgpr(SC(-1.3352627688545895))}; }
};
// cos_any with every Horner step as v_fma_f64 v, v, v, s (fma_vvs): the same operations in the same order as cos_any /
// sincos_turns<2> - bit for bit the same cosine - but no coefficient is copied into a vector register per evaluation (LLVM
// otherwise selects v_fmac_f64 and moves each addend into its destination first: 12 v_mov_b32 per mass in the scan's loop, a
// quarter of its vector instructions).
__device__ __forceinline__ double cos_any_vvs(double y, const double* __restrict__ table, const GasCos& C) {
  const double n = __builtin_rint(y * 0.15915494309189535);
  double f = fma(y, 0.15915494309189535, -n);        // 1 / (2 pi), high part
  f = fma(y, -9.839338337591243e-18, f);             // low part
  const double u = fabs(f);                          // cos is even; |f| <= 0.5 + 1e-16
  const double kf = __builtin_rint(u * 128.0);
  const double r = fma(-1.0 / 64.0, kf, u * 2.0);
  const double2 t = reinterpret_cast<const double2*>(table)[(int)kf];
  const double x = r * r;
  const double sr = HORNER(HORNER(HORNER(C.s3, x, 2.5501640398773455), x, -5.16771278004997), x, 3.141592653589793) * r;
  const double cr = fma(HORNER(HORNER(C.d3, x, 4.0587121264167685), x, -4.934802200544679), x, 1.0);
  double cs = fma(-t.y, sr, t.x * cr);
  if (!(fabs(y) < 1e15)) {
    asm volatile("; rare: cos of a huge or non-finite phase");
    cs = cos(y);
  }
  return cs;
}
__device__ __forceinline__ double gas_conversion_prob(double dm2_abs, const GasRay& G, double term1, const double* __restrict__ table,
                                                      const GasCos& C) {
  const double q = dm2_abs * G.inv_two_e;
  const double term2 = frcp(fma(q, q, G.g2q));
  const double term3 = fma(G.m2eh, cos_any_vvs(q * G.Lnat, table, C), G.eh2p1);
  return (term1 * term2) * term3;
}

// The table pointers of a launch are read from the LDS copy of the parameter blob, where the compiler cannot see
// their address space and would emit flat_load (which counts on both vmcnt and lgkmcnt and so serialises against the
// LDS traffic of the same wave).  They always point to global memory: say so.
template <typename T>
using global_ptr = const __attribute__((address_space(1))) T*;
template <typename T>
__device__ __forceinline__ global_ptr<T> as_global(const T* p) {
  return (global_ptr<T>)p;
}

// An object in LDS seen through an address the compiler cannot fold: field accesses become ds_read with ONE base register
// and immediate offsets.  With the address known at compile time every access to a field beyond the 64 KB immediate range
// gets its own v_mov_b32 of the absolute address (one vector instruction per parameter read).
template <typename T>
__device__ __forceinline__ const T& lds_opaque(const T& obj) {
  typedef const __attribute__((address_space(3))) T* lds_ptr;
  uint32_t a = (uint32_t)(uintptr_t)(lds_ptr)&obj;
  asm volatile("" : "+v"(a));
  return *(const T*)(lds_ptr)(uintptr_t)a;
}

// lowerBound(a, key) with the answer known to lie in [lo, hi] (guide-table bracket):
// first index i with a[i] >= key  (std/algorithm.lowerBound semantics).
template <typename Ptr>
__device__ __forceinline__ int lower_bound_bracket(Ptr a, int lo, int hi, double key) {
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (a[mid] < key) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// Roots of a z^2 + 2 hb z + c = 0 in the reference's naming:
//   root1 = (-hb - sqrt(hb^2 - a c)) / a,  root2 = (-hb + sqrt(hb^2 - a c)) / a     (:649-650)
// evaluated without cancellation (q-form); root1 is preferred, then root2, else miss (:651-656).
__device__ __forceinline__ bool pick_root(double a, double hb, double c, double zlo, double zhi, double& z, uint64_t& hit_m) {
  const double disc = fma(hb, hb, -a * c);
  const double sq = fsqrt_pos(disc);  // NaN for disc <= 0 -> every comparison below is false -> miss (disc == 0: tangent ray, measure zero)
  // q = -(hb + sign(hb) sq); the two roots are q / a and c / q.  With hb >= 0 (sign bit clear) q / a is the reference's
  // root1 and c / q its root2, with hb < 0 the other way round.
  const double q = -hb - copysign(sq, hb);
  const double raq = frcp(a * q);            // one reciprocal: 1/a = q raq, 1/q = a raq
  const double ra = q * q * raq, rb = c * a * raq;
  const bool a_lo = ra > zlo, a_hi = ra < zhi, b_lo = rb > zlo, b_hi = rb < zhi;
  const bool in_a = a_lo & a_hi, in_b = b_lo & b_hi;
  hit_m = (ballot64(a_lo) & ballot64(a_hi)) | (ballot64(b_lo) & ballot64(b_hi));   // the verdict as a lane mask (see phase_a_core)
  // root1 if it lies in the mirror's z range, else root2, else miss (:651-656): only when both roots are in range does
  // the order matter, and then root1 is q / a exactly when the sign bit of hb is clear
  const bool a_first = __double2hiint(hb) >= 0;
  z = (in_b & !(in_a & a_first)) ? rb : ra;
  return in_a | in_b;
}

// Reflection of the (un-normalised) direction w, |w|^2 = L, at a surface with (un-normalised)
// normal n, |n|^2 = N2:  the reference's v' = v cos 2a - (v x axis) sin 2a with a = asin|n.v|/|n|
// (:774-779) equals  v (1 - 2c^2 + 2c|c|) - 2|c| n^  with c = n^.v ; written for w = |w| v.
// Returns sin^2(a) = c^2 and, in sin_a, sin(a) = |c| (from the reciprocal square root the formula needs anyway: the
// reflectivity lookup wants the angle itself, and a separate square root of c^2 would cost seven more instructions).
__device__ __forceinline__ double reflect(double& wx, double& wy, double& wz, double L, double nx, double ny,
                                          double nz, double N2, double& sin_a) {
  const double dnw = fma(nx, wx, fma(ny, wy, nz * wz));
  const double y = frsq(N2 * L);             // 1 / (|n| |w|)
  const double rnl = y * y;                  // 1 / (N2 L): serves 1/N2 and 1/(N2 L)
  sin_a = fabs(dnw) * y;
  const double c2 = dnw * dnw * rnl;
  const double f = dnw * (L * rnl);
  const double ox = wx, oy = wy, oz = wz;
  wx = fma(-2.0 * f, nx, ox);                // mirror reflection (normal on the far side of the ray: n.w >= 0)
  wy = fma(-2.0 * f, ny, oy);
  wz = fma(-2.0 * f, nz, oz);
  if (dnw < 0.0) {                           // practically never taken (skipped when no lane needs it)
    // normal facing the ray: the reference's formula is then not a mirror reflection; keep it
    asm volatile("; rare: normal facing the ray");   // keeps LLVM from turning the block into always-executed selects
    const double k = 1.0 - 4.0 * c2;
    wx = fma(2.0 * f, nx, k * ox);
    wy = fma(2.0 * f, ny, k * oy);
    wz = fma(2.0 * f, nz, k * oz);
  }
  return c2;
}

// calcNormalVec (:731-759) z-component for an arbitrary point (general form with the square root; needed when the ray missed
// the mirror and the reference evaluates the normal at its input point: the rays that go on to the nickel test, an eighth of
// BabyIAXO's phase-B rays - some lane of nearly every pass).  rho / sqrt(w) = rho^2 / sqrt(rho^2 w): ONE reciprocal square root
// from the hardware seed (sart_math.h) instead of two IEEE square roots and a division (17 instead of ~90 vector instructions).
__device__ __forceinline__ double normal_z_general(const DevParams& P, const ShellDev& sh, int mirror, double x,
                                                   double y, double z) {
  const double rho2 = fma(x, x, y * y);
  if (!P.telescope_wolter) return (mirror == 1 ? sh.n1_tan : sh.n2_tan) * (rho2 * frsq(rho2));
  const double lz = P.l_mirror - z;
  if (mirror == 1) return sh.n1_r3t * (rho2 * frsq(rho2 * fma(sh.n1_e, lz, sh.n1_r3sq)));
  const double w = fma(sh.n2_e * lz, fma(lz, sh.n2_invF, 1.0), sh.n2_r3sq);
  return sh.n2_r3t * fma(2.0 * lz, sh.n2_invF, 1.0) * (rho2 * frsq(rho2 * w));
}

// ------------------------------------------------------------------------------------------------
// LDS-resident tables of one workgroup
// ------------------------------------------------------------------------------------------------
struct LdsTables {
  const double* sincos;      // (cos, sin)(pi k / 64), k = 0 .. 128
  const uint32_t* rcdf_hi;   // fluxRadiusCDF as the upper 32 of the 52 bits of floor(cdf 2^52), + 1 (stage_tables)
  const double* rcdf_f64;    // the f64 table in device memory: decides the ties of the 32-bit compare (one draw in ~1e9)
  const uint16_t* rguide;    // guide table in front of it
  const ShellDev* shells;
  const uint8_t* lut;        // radial look-up table of the shell selection
};

// State of a ray between phase A and phase B.
struct RayState {
  double X0, Y0;     // pointEntranceXRT (telescope frame, z = 0)
  double tsx, tsy;   // slopes dX/dz, dY/dz in the telescope frame
  double path_cb;    // path length inside the magnetic field
  double u5;         // uniform of the energy draw
  double zcb;        // z of pointExitCB in the telescope frame (z0 of :2051)
  int r_idx;         // sampled solar radius index (row of diffFluxCDFs)
  int shell;         // hit layer
};

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// SART_ACCUM_FIXED64: rint(x * scale) as an integer, |x * scale| < 2^51, in one FMA and one 64-bit subtraction: adding
// 1.5 * 2^52 puts the sum into the binade whose unit in the last place is 1, so the FMA's single rounding (to nearest,
// ties to even) IS the rounding to an integer, and the integer sits in the low mantissa bits.
__device__ __forceinline__ long long to_fixed(double x, double scale) {
  constexpr double kMagic = 6755399441055744.0;   // 1.5 * 2^52
  return __double_as_longlong(fma(x, scale, kMagic)) - __double_as_longlong(kMagic);
}
__device__ __forceinline__ long long wave_sum_i64(long long v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
// 8-byte slots of the accumulators hold doubles (SART_ACCUM_F64) or int64 bit patterns (SART_ACCUM_FIXED64)
__device__ __forceinline__ void atomic_add_slot_i64(double* slot, long long v) {
  atomicAdd(reinterpret_cast<unsigned long long*>(slot), (unsigned long long)v);   // global_atomic_add_x2, no return
}

// Gather from an HBM-resident table: wave-uniform base (scalar registers) + 32-bit byte offset per lane.
typedef double d2 __attribute__((ext_vector_type(2)));   // native vector: loads as one 16-byte access
template <typename T>
__device__ __forceinline__ T gload(const void* base, uint32_t byte_off) {
  typedef const __attribute__((address_space(1))) char* gbytes;
  return *reinterpret_cast<const __attribute__((address_space(1))) T*>((gbytes)base + byte_off);
}

// Energy index drawn with u5 from the CDF row of the sampled radius (getRandomEnergyFromSolarModel, :444-471) =
// lowerBound(row, u5), in three steps that phase B spreads over its mirror arithmetic so that the three gathers of the
// draw (guide word -> four CDF candidates -> [energy row, reflectivities]) are in flight while the wave computes:
//   begin       bucket of u5 (sart_device.h: kEnergyGuide*), gather of the guide word (entries k, k + 1 of the row)
//   candidates  bracket [lo, hi] from the guide word, gather of row[lo .. lo + 3] (rows are padded with 1.0)
//   finish      lo + #{candidates < u5}: the row is sorted and row[hi] >= u5, so candidates at or beyond hi never count;
//               only a lane whose four candidates are all < u5 with hi > lo + 4 goes on with a binary search
typedef uint32_t u4v __attribute__((ext_vector_type(4)));   // native vector: loads as one 16-byte access
struct EnergyDraw {
  double u;
  uint32_t khi;        // upper 32 of the 52 bits of u 2^52
  uint32_t row;        // element offset of the CDF row
  uint32_t gword;      // guide entries k (low half) and k + 1 (high half)
  uint32_t lo, hi;
  u4v c;               // cdf_hi32 of row[lo .. lo + 3]
};
__device__ __forceinline__ void energy_draw_begin(const HotB& HB, int r_idx, double u5, EnergyDraw& d, int r_idx_guide = -1) {
  if (r_idx_guide < 0) r_idx_guide = r_idx;   // (experiment builds pass another row for the guide gather: working-set sensitivity)
  d.u = u5;
  {
    // u5 = k 2^-52 exactly (u52: 52 random mantissa bits under the exponent of 1.0, minus 1.0): 1 + u5 carries k in its
    // mantissa, and its upper 32 bits are one funnel shift away.  (A uniform handed in from outside - the test entry - may
    // have bits below 2^-52: the sum then rounds k by at most 1/2, which energy_draw_finish allows for; the clamp keeps a
    // uniform within 2^-53 of 1 out of the next binade.)
    const double w = fmin(u5 + 1.0, 1.9999999999999998);
    d.khi = __builtin_amdgcn_alignbit((uint32_t)__double2hiint(w), (uint32_t)__double2loint(w), 20u);
  }
  const double v = 1.0 - u5;
  const uint32_t ku = (uint32_t)(int)(u5 * (double)kEnergyGuideDiv);
  const uint32_t code = (uint32_t)__double2hiint(v) >> 14;                       // exponent and six mantissa bits of v
  const uint32_t kl = (uint32_t)kEnergyGuideUniform + min(kEnergyGuideCode0 - code, (uint32_t)kEnergyGuideLogMax);
  const uint32_t k = (v > 0.03125) ? ku : kl;                                    // u5 < 31/32: uniform buckets
  d.row = __umul24((uint32_t)r_idx, (uint32_t)HB.cdf_stride);                    // both < 2^24
  // two adjacent u16 as one (possibly unaligned) 32-bit load
  d.gword = gload<uint32_t>(HB.energy_guide, (__umul24((uint32_t)r_idx_guide, (uint32_t)kEnergyGuideEntries) + k) * 2u);
}
__device__ __forceinline__ void energy_draw_candidates(const HotB& HB, EnergyDraw& d) {
  d.lo = d.gword & 0xFFFFu;
  d.hi = d.gword >> 16;
  // four consecutive u32 of a row whose stride is a multiple of four entries only by accident: the 16 bytes may straddle a
  // 16-byte boundary, global_load_dwordx4 needs 4-byte alignment only
  d.c = gload<u4v>(HB.cdf_hi32, (d.row + d.lo) * 4u);
}
__device__ __forceinline__ int energy_draw_finish(const HotB& HB, const EnergyDraw& d) {
  const double u = d.u;
  // With T = floor(cdf 2^52) >> 20 (the table) and K = the upper 32 bits of u 2^52 rounded to an integer k':  T < K - 1 =>
  // floor(cdf 2^52) <= k' - 2^20 - 1 => cdf < u;  T > K => cdf 2^52 >= k' + 1 => cdf >= u (u 2^52 lies within 1/2 of k').
  // T in {K - 1, K}: undecided (2^-31 per candidate) - the f64 row decides, as it does in a wide bucket whose first four
  // entries are all below u.  (The 1.0 pads behind a row read 0xFFFFFFFF.)
  const uint32_t km1 = max(d.khi, 1u) - 1u;
  uint32_t lo = d.lo + (uint32_t)(d.c.x < km1) + (uint32_t)(d.c.y < km1) + (uint32_t)(d.c.z < km1) + (uint32_t)(d.c.w < km1);
  const bool tie = ((d.c.x - km1) < 2u) | ((d.c.y - km1) < 2u) | ((d.c.z - km1) < 2u) | ((d.c.w - km1) < 2u);
  if (((d.c.w < km1) & (d.hi > d.lo + 4u)) | tie) {
    asm volatile("; rare: energy bucket wider than four entries, or a tie in the upper 32 bits");
    if (tie) lo = d.lo;
    uint32_t hi = d.hi;
    while (lo < hi) {
      const uint32_t mid = (lo + hi) >> 1;
      if (gload<double>(HB.diff_flux_cdfs, (d.row + mid) * 8u) < u) lo = mid + 1; else hi = mid;
    }
  }
  return (int)min(lo, (uint32_t)(HB.n_energies - 1));
}
// the three steps back to back (record mode)
__device__ __forceinline__ int sample_energy_index(const HotB& HB, int r_idx, double u5) {
  EnergyDraw d;
  energy_draw_begin(HB, r_idx, u5, d);
  energy_draw_candidates(HB, d);
  return energy_draw_finish(HB, d);
}

static_assert(sizeof(EnergyDev) == 8 * sizeof(double), "EnergyDev is loaded as eight f64");
__device__ __forceinline__ EnergyDev load_energy_row(const HotB& HB, int e_idx) {
  // field by field: the compiler merges neighbours and drops what a variant does not use (a wider load with a dead half
  // would have that half's registers reused while the gather is still in flight, which forces an early wait)
  const uint32_t off = (uint32_t)e_idx * 64u;
  return EnergyDev{gload<double>(HB.energy_tab, off), gload<double>(HB.energy_tab, off + 8u), gload<double>(HB.energy_tab, off + 16u),
                   gload<double>(HB.energy_tab, off + 24u), gload<double>(HB.energy_tab, off + 32u), gload<double>(HB.energy_tab, off + 40u),
                   gload<double>(HB.energy_tab, off + 48u), gload<double>(HB.energy_tab, off + 56u)};
}

// cos(n phi) from c = cos(phi) for the spoke test cos(n phi) >= cos(n w): Chebyshev T6 = T2(T3), or T16 by four doublings in
// scaled form, one FMA each: with u = c^2 - 1/2, v = u^2 - 1/8, w = v^2 - 1/128, z = w^2 - 1/32768 the doublings give
// T2 = 2u, T4 = 8v, T8 = 128w, T16 = 32768 z (powers of two: exact), so the function returns z and the host stores the
// threshold divided by 32768 (sart_api.hip: hot_of).
// The addends are scalar operands (HORNER: v_fma_f64 v, v, v, s): as literals LLVM copies each of them into the destination
// of a two-address v_fmac_f64 first, two v_mov_b32 per step (16 of the vector instructions of a phase-A pass).
__device__ __forceinline__ double spoke_measure(int n, double c) {
  if (n == 16) {
    const double u = HORNER(c, c, -0.5);
    const double v = HORNER(u, u, -0.125);
    const double w = HORNER(v, v, -0.0078125);
    return HORNER(w, w, -1.0 / 32768.0);
  }
  const double t3 = c * HORNER(4.0 * c, c, -3.0);
  return HORNER(2.0 * t3, t3, -1.0);
}

// ------------------------------------------------------------------------------------------------
// phase A: sample -> bore -> pipes -> telescope frame -> opaque structures -> shell selection
// (:1746-1957).  Returns true if the ray goes on to the mirrors; `reached` = survived bore + pipes.
// st.r_idx / st.u5 are valid whenever the collimator cut (X-ray test source) was passed (`sampled`).
//
// Written in predicated form: every cut of the reference is an `ok &= ...` instead of an early return.
// A wave of 64 rays practically always has survivors up to the shell selection, so an early return
// saves no issue slots - it only costs exec-mask bookkeeping (SALU + SGPRs).  Lanes whose ray is dead
// keep computing on finite values; their results are discarded.
// ------------------------------------------------------------------------------------------------
//
// FAST = the configuration known at compile time to be: solar source (no X-ray test source), telescope
// not rotated, no hole loop in the optics.  The generic instantiation reads those switches at run time.
// ROT: telescope rotation known at compile time (0 / 1) or read at run time (-1).
// ZEXT: st.path_cb carries the z extent of the path in the magnetic field instead of its length (phase B of the vacuum,
// unrotated specialisation multiplies pathCB^2 by 1 + slope^2 itself and needs no square root here).
// The six uniforms of a ray in the reference's draw order: solar source u0, u1 -> angles of the solar point
// (:433-434), u2 -> radius CDF (:436), u3 -> disc radius (:418), u4 -> disc angle (:419), u5 -> energy CDF (:464).
struct Uniforms {
  double u0, u1, u2, u3, u4, u5;
  uint32_t u2_hi;   // floor(u2 2^32): the radius draw's guide bucket, and what it compares with the 32-bit copy of the CDF in LDS
};
// u2_hi of a uniform that does not come with its random words (the test entry of the record kernel)
__device__ __forceinline__ uint32_t upper32_of_uniform(double u) {
  return (uint32_t)(u * 4294967296.0);   // floor(u 2^32), exact for every u in [0, 1)
}
struct LaneMasks { uint64_t ok, reached; };   // phase A's verdicts as wave-wide lane masks (see phase_a_bore)

// Phase A is written in two halves that meet where the magnet's pipes end and the telescope begins (:1868 | :1878): everything
// in the first half - sampling, bore, cold-bore exit, pipe cuts - is independent of the telescope's angles, everything in the
// second half depends on them.  The histogram and record kernels run them back to back (phase_a_core); the fused angular scan
// (trace_angular_scan_kernel) runs the first half once per ray and the second half once per ray and angle.
// What the first half hands to the second:
struct BoreRay {
  double x1, y1;     // the ray at the cold-bore exit (z = -Lp in the telescope's frame before its rotation)
  double x3, y3;     // ... and at the end of the second pipe (z = 0 there)
  double sx, sy;     // slopes dx/dz, dy/dz in the magnet frame
  bool ok;           // alive so far (= `reached`)
  uint64_t okm;      // the same as a lane mask
};
// The rotation of the telescope about (0, 0, lT/2) (:1888-1894) as hoist_setup() evaluates it (DevParams holds the same five
// values in the same order: phase_a_core takes them from there; the angular scan has one set per angle in its kernel arguments).
struct TelRot {
  double rx_c, rx_s, ry_c, ry_s, half_length_telescope;
};
static_assert(offsetof(DevParams, rx_s) == offsetof(DevParams, rx_c) + 8 && offsetof(DevParams, ry_c) == offsetof(DevParams, rx_c) + 16 &&
                  offsetof(DevParams, ry_s) == offsetof(DevParams, rx_c) + 24 && offsetof(DevParams, half_length_telescope) == offsetof(DevParams, rx_c) + 32,
              "TelRot mirrors these five fields of DevParams");
__device__ __forceinline__ const TelRot& tel_rot_of(const DevParams& P) { return *reinterpret_cast<const TelRot*>(&P.rx_c); }

// ---- first half: sample -> bore -> cold-bore exit -> pipes (:1746-1868) ----
template <bool FAST, bool ZEXT, bool NOWALL = false>
__device__ __forceinline__ void phase_a_bore(const HotA& H, const DevParams& P, const LdsTables& L, const Uniforms& U, RayState& st,
                                             bool& sampled, bool& reached, BoreRay& br, LaneMasks& M) {
  const bool cfg_test = FAST ? false : (H.test_active != 0);
  const double u0 = U.u0, u1 = U.u1, u2 = U.u2, u3 = U.u3, u4 = U.u4;
  st.u5 = U.u5;
  st.r_idx = 0;

  bool ok = true;
  double ex, ey;            // point on the magnetic-field exit plane z = lengthB
  double sx, sy;            // ray slopes dx/dz, dy/dz in the magnet frame
  if (!cfg_test) {
    // getRandomPointFromSolarModel (:425-442): theta1 = 360 u0 deg, theta2 = 180 u1 deg (uniform in theta)
    double s1, c1, s2, c2;
    const SinCosCoef K = sincos_coef();   // one set of scalar constants for the three evaluations
    sincos_turns<2>(u0, L.sincos, K, &s1, &c1);
    sincos_turns<1>(u1, L.sincos, K, &s2, &c2);
    {
      // lowerBound(fluxRadiusCDF, u2) (:437).  K = floor(u2 2^32) picks the guide bucket (sart_device.h: 2048 buckets, and 1024
      // finer ones for u2 >= 31/32) whose two entries bracket the index.  The table sits in LDS as T' = floor(cdf 2^32) + 1 (half the
      // bytes of the f64 table: the other half is image tile):  T' < K => cdf < u2;  T' > K + 1 => cdf > u2;  T' in {K, K + 1}:
      // undecided, 2^-31 per entry looked at - the f64 table in device memory decides.  (The 1.0 pads read 0xFFFFFFFF.)
      const uint32_t khi = U.u2_hi;
      const uint32_t kb = khi >> 21, kt = (khi >> 17) - ((kRadiusGuideTopStart >> 17) - (uint32_t)(kRadiusGuide + 1));
      const uint32_t k = khi >= kRadiusGuideTopStart ? kt : kb;
      const int lo0 = (int)L.rguide[k];
      const int hi0 = (int)L.rguide[k + 1];
      // four consecutive candidates in ONE round trip to LDS: the table is sorted and rcdf[hi] >= u2, so entries at or beyond hi
      // never count (the stage pads the table with four entries of 1.0)
      const uint32_t c0 = L.rcdf_hi[lo0], c1 = L.rcdf_hi[lo0 + 1], c2 = L.rcdf_hi[lo0 + 2], c3 = L.rcdf_hi[lo0 + 3];
      int lo = lo0 + (int)(c0 < khi) + (int)(c1 < khi) + (int)(c2 < khi) + (int)(c3 < khi);
      // undecided: an entry in {K, K + 1}, i.e. the smallest of the four differences (unsigned: entries below K wrap to huge) < 2
      bool tie = min(min(c0 - khi, c1 - khi), min(c2 - khi, c3 - khi)) < 2u;
      if ((c3 < khi) & (hi0 > lo0 + 4)) {
        asm volatile("; rare: radius bucket wider than four entries");
        int hi = hi0;
        while (lo < hi) {   // first entry that is not certainly below u2
          const int mid = (lo + hi) >> 1;
          if (L.rcdf_hi[mid] < khi) lo = mid + 1; else hi = mid;
        }
        tie = (L.rcdf_hi[lo] - khi) < 2u;   // that entry is undecided (everything behind it is >= it; the four in front were certain)
      }
      if (tie) {
        asm volatile("; rare: a tie in the upper 32 bits of the radius draw");
        lo = lower_bound_bracket(as_global(L.rcdf_f64), lo0, hi0, u2);
      }
      st.r_idx = lo;
    }
    const double r = (0.0015 + (double)st.r_idx * 0.0005) * H.sun_radius;
    const double ox = c1 * s2 * r, oy = s1 * s2 * r, oz = c2 * r - H.sun_distance;
    // getRandomPointOnDisk (:412-422)
    double sp, cp;
    sincos_turns<2>(u4, L.sincos, K, &sp, &cp);
    // + 1e-300: exact no-op unless u3 == 0 (one ray in 2^44: its point lands 1e-150 R from the axis instead of on it); saves
    // the zero / negative special-casing of fsqrt() (two compares, four selects) in every pass
    const double rr = H.radius_cb * fsqrt_pos(u3 + 1e-300);
    ex = cp * rr;
    ey = sp * rr;
    const double inv_dz = frcp(H.length_b - oz);
    sx = (ex - ox) * inv_dz;
    sy = (ey - oy) * inv_dz;
  } else {
    // X-ray test source (:1765-1806)
    double sp, cp;
    const SinCosCoef K = sincos_coef();
    sincos_turns<2>(u1, L.sincos, K, &sp, &cp);
    const double rr = P.test_radius * fsqrt(u0);
    const double ox = cp * rr + P.test_x, oy = sp * rr + P.test_y, oz = P.test_z;
    if (P.test_parallel) {
      ex = ox + (u2 * 0.5) - 0.25;
      ey = oy + (u3 * 0.5) - 0.25;
    } else {
      double sq, cq;
      sincos_turns<2>(u3, L.sincos, K, &sq, &cq);
      const double r2 = H.radius_cb * fsqrt(u2);
      ex = cq * r2;
      ey = sq * r2;
    }
    const double inv_dz = frcp(H.length_b - oz);
    sx = (ex - ox) * inv_dz;
    sy = (ey - oy) * inv_dz;
    // collimator (:1800): lineIntersectsCircle(origin, exit point, collimator, source radius)
    const double dzc = P.test_collimator_z - H.length_b;
    const double cx = fma(dzc, sx, ex) - P.test_x, cy = fma(dzc, sy, ey) - P.test_y;
    ok = fma(cx, cx, cy * cy) < P.test_radius_sq;
  }
  sampled = ok;
  // The same verdicts as lane masks, combined on the scalar unit from the ballots of the single compares (the ballot of a
  // direct compare is the compare's own result; a ballot of `a & b & c` costs a select and a second compare): what the
  // histogram kernel counts and compacts with.  The bools stay for the divergent regions and for the record kernel.
  uint64_t okm = cfg_test ? ballot64(ok) : ~0ull;

  // ---- bore (:1813-1848) ----
  const double A2 = fma(sx, sx, sy * sy);     // slope^2
  const double norm = ZEXT ? 1.0 : fsqrt_pos(1.0 + A2);   // ZEXT: phase B applies the factor (1 + A2) to pathCB^2 itself
  // entrance plane z = 0
  double path_cb = H.length_b * norm;          // |exit point - entrance-plane point| (:1836-1843); FAST: its z extent
  // NOWALL (constant-path variant): the host has proved that no ray that passes the three cuts below entered through the bore
  // wall (sart_api.hip: path_is_constant), so a wall entry changes neither `ok` after those cuts nor `reached`, and nobody
  // reads the path: the entrance-plane test and the wall branch are not compiled.
  bool hits_entrance = true;
  if (!NOWALL) {
    const double x0 = fma(-H.length_b, sx, ex), y0 = fma(-H.length_b, sy, ey);
    hits_entrance = fma(x0, x0, y0 * y0) < H.radius_cb_sq;
  }
  if (!NOWALL && (ok & !hits_entrance)) {      // divergent, skipped when no lane needs it: the ray entered through the bore wall
    // lineIntersectsCylinderOnce (:591-604): intersections of the line with the bore wall,
    // t = z - lengthB:  A2 t^2 + 2 Dm t + (Qm - R^2) = 0.  inter1 = larger z, inter2 = smaller z.
    const double Dm = fma(ex, sx, ey * sy);
    const double c = fma(ex, ex, ey * ey) - H.radius_cb_sq;
    const double sq = fsqrt(fma(Dm, Dm, -A2 * c));
    const double q = (Dm >= 0.0) ? (-Dm - sq) : (-Dm + sq);
    const double qa = q * frcp(A2), cq = c * frcp(q);
    const double t_lo = (Dm >= 0.0) ? qa : cq, t_hi = (Dm >= 0.0) ? cq : qa;
    const double z_hi = H.length_b + t_hi, z_lo = H.length_b + t_lo;
    const bool v1 = (z_hi > 0.0) && (z_hi < H.length_coldbore);
    const bool v2 = (z_lo > 0.0) && (z_lo < H.length_coldbore);
    const double t = v1 ? t_hi : t_lo;         // :616
    ok = (v1 != v2);                           // exactly one valid intersection (:598-600, :1825)
    path_cb = fabs(t) * norm;
  }
  if (!NOWALL) okm = ballot64(ok);             // `ok` may come out of the divergent region above
  st.path_cb = path_cb;
  // exit of the cold bore (:1846), pipe CB -> VT3 (:1856), VT3 -> XRT (:1866; same radius — sic)
  const double x1 = fma(H.dz1, sx, ex), y1 = fma(H.dz1, sy, ey);
  const double x2 = fma(H.dz2, sx, ex), y2 = fma(H.dz2, sy, ey);
  const double x3 = fma(H.dz3, sx, ex), y3 = fma(H.dz3, sy, ey);
  // bitwise on purpose: `&&` keeps the short-circuit as three nested divergent regions (exec-mask bookkeeping around eight
  // f64 operations that nearly every lane needs anyway)
  const bool in1 = fma(x1, x1, y1 * y1) < H.radius_cb_sq, in2 = fma(x2, x2, y2 * y2) < H.pipe1_radius_sq,
             in3 = fma(x3, x3, y3 * y3) < H.pipe1_radius_sq;
  ok = ok & in1 & in2 & in3;
  okm &= ballot64(in1) & ballot64(in2) & ballot64(in3);
  reached = ok;
  M.reached = okm;
  br.x1 = x1; br.y1 = y1; br.x3 = x3; br.y3 = y3; br.sx = sx; br.sy = sy;
  br.ok = ok;
  br.okm = okm;
}

// ---- second half: telescope frame -> opaque structures -> shell selection (:1878-1957) ----
// `R`: the telescope's rotation; `shell0`-style shortcuts stay with the caller.  Returns true if the ray goes on to the mirrors.
template <bool FAST, int ROT>
__device__ __forceinline__ bool phase_a_telescope(const HotA& H, const DevParams& P, const TelRot& R, const LdsTables& L, const BoreRay& br,
                                                  RayState& st, double& radial_out, LaneMasks& M) {
  const bool cfg_rotated = (ROT < 0) ? (H.rotated != 0) : (ROT != 0);
  const bool cfg_holes = FAST ? false : (H.telescope_kind == SART_TK_XMM && H.inner_blocks < 0);
  const double x1 = br.x1, y1 = br.y1, x3 = br.x3, y3 = br.y3, sx = br.sx, sy = br.sy;
  bool ok = br.ok;
  uint64_t okm = br.okm;

  // ---- telescope frame (:1878-1899) ----
  // pointExitCB' (z = -Lp before rotation) and pointExitPipeVT3XRT' (z = 0 before rotation)
  const double Lp = H.dz3 - H.dz1;
  double X0, Y0, tsx, tsy, zcb;
  if (!cfg_rotated) {
    X0 = x3 - H.entrance_x;
    Y0 = y3 - H.entrance_y;
    tsx = sx;
    tsy = sy;
    zcb = -Lp;
  } else {
    // rotateInY(rotateInX(p, turnedX, lT/2), turnedY, lT/2) applied to both points (:1888-1894)
    auto rot = [&](double px, double py, double pz, double& qx, double& qy, double& qz) {
      const double zz = pz - R.half_length_telescope;
      const double ax = fma(px, R.rx_c, zz * R.rx_s);
      const double az = fma(zz, R.rx_c, -px * R.rx_s);   // rotateInX adds lT/2 back, rotateInY removes it again
      qx = ax;
      qy = fma(py, R.ry_c, -az * R.ry_s);
      qz = fma(az, R.ry_c, py * R.ry_s) + R.half_length_telescope;
    };
    double ax, ay, az, bx, by, bz;
    rot(x1, y1, -Lp, ax, ay, az);
    rot(x3, y3, 0.0, bx, by, bz);
    ax -= H.entrance_x; ay -= H.entrance_y;
    bx -= H.entrance_x; by -= H.entrance_y;
    const double inv = frcp(bz - az);
    tsx = (bx - ax) * inv;
    tsy = (by - ay) * inv;
    X0 = fma(-az, tsx, ax);     // pointEntranceXRT: z = 0 (:1897-1898)
    Y0 = fma(-az, tsy, ay);
    zcb = az;
  }
  st.X0 = X0; st.Y0 = Y0; st.tsx = tsx; st.tsy = tsy; st.zcb = zcb;
  const double Q0 = fma(X0, X0, Y0 * Y0);
  const double inv_radial = frsq(Q0);
  const double radial = Q0 * inv_radial;       // radialDist (:1905)

  // ---- opaque structures (:1635-1704) ----
  if (H.telescope_kind != SART_TK_LLNL) {      // LLNL: the graphite block never blocks (:1646)
    const bool inner_le = radial <= H.inner_radius, inner_lt = radial < H.inner_radius;
    const bool inner = (H.telescope_kind == SART_TK_XMM) ? inner_le : inner_lt;
    // htNone: the hole test is always false => inner disc blocked (:1683-1688, :527); Abrixas :1653;
    // XMM ring :1691; spider spokes tested on phi = acos(x / r) at the entrance plane and at
    // z = spider_z (:1695-1701): every 360/n degrees, |phi - k 360/n| <= w  <=>  cos(n phi) >= cos(n w)
    const bool ring_a = radial < H.ring_hi, ring_b = radial > H.ring_lo;
    const bool ring = (H.telescope_kind == SART_TK_XMM) & ring_a & ring_b;
    const double c_ent = X0 * inv_radial;
    const double xs = fma(H.spider_z, tsx, X0), ys = fma(H.spider_z, tsy, Y0);
    const double c_sp = xs * frsq(fma(xs, xs, ys * ys));
    // (bitwise: the second test is needed by every lane the first one does not block - no divergent region around it)
    const bool spoke_a = spoke_measure(H.spoke_n, c_ent) >= H.spoke_cos_thr, spoke_b = spoke_measure(H.spoke_n, c_sp) >= H.spoke_cos_thr;
    const bool spoke = spoke_a | spoke_b;
    bool blocked = inner | ring | spoke;
    const bool xmm = H.telescope_kind == SART_TK_XMM;   // wave-uniform
    uint64_t blocked_m = (xmm ? ballot64(inner_le) : ballot64(inner_lt)) | (xmm ? (ballot64(ring_a) & ballot64(ring_b)) : 0ull) |
                         ballot64(spoke_a) | ballot64(spoke_b);
    if (cfg_holes) {
      // hole loop (:1675-1688) with lineIntersectsObject (:494-527) on the entrance plane; replaces the
      // verdict for rays inside the inner disc
      const int nH = P.number_of_holes;
      const int lim = nH - (int)ceil((double)nH / 2.0);
      bool res = false, done = false;
      for (int l = -lim; l <= lim; ++l) {
        double hx = 0.0, hy = 0.0;
        if (l != 0) {
          if ((abs(l) & 1) == 0) hy += 2.0 * (double)l * P.hole_in_optics;
          else hx += 2.0 * ((double)l + ((double)l / (double)abs(l))) * P.hole_in_optics;
        }
        const double ix = X0 - hx, iy = Y0 - hy, rad = P.hole_in_optics;
        const double tx = ix / sqrt(2.0) - iy / sqrt(2.0), ty = ix / sqrt(2.0) + iy / sqrt(2.0);
        const double axx = fabs(ix), ayy = fabs(iy), atx = fabs(tx), aty = fabs(ty);
        bool through = false;
        switch (P.hole_type) {
          case SART_HT_CIRCLE: through = sqrt(fma(ix, ix, iy * iy)) < rad; break;
          case SART_HT_CROSS:
            through = (axx < rad && ayy < rad * 16.0) || (ayy < rad && axx < rad * 16.0); break;
          case SART_HT_STAR:
            through = (axx < rad && ayy < rad * 16.0) || (ayy < rad && axx < rad * 16.0) ||
                      (atx < rad && aty < rad * 16.0) || (aty < rad && atx < rad * 16.0); break;
          case SART_HT_SQUARE: through = axx < rad && ayy < rad; break;
          case SART_HT_DIAMOND: through = atx < rad && aty < rad; break;
          default: through = false;
        }
        if (!done) { res = !through; done = through; }   // `break` at the first hole the ray passes through
      }
      if (inner) blocked = res;
      blocked_m = ballot64(blocked);
    }
    ok = ok & !blocked;
    okm &= ~blocked_m;
  }

  // ---- shell selection (:1932-1957) ----
  const bool beyond = radial > H.r1_last;
  ok = ok & !beyond;
  okm &= ~ballot64(beyond);
  // R1 ascending: the nearest shell above is the first j with R1[j] > radial; the look-up cell (narrower
  // than any shell spacing) gives it up to one step
  const int nS = H.n_shells;
  const int j0 = (int)L.lut[max(min((int)(radial * H.lut_inv_step), H.lut_n - 1), 0)];
  // the radii of the cell's shell and of the one below it, read together (one round trip to LDS instead of two dependent ones)
  const int jc = min(j0, nS - 1), jb = max(jc - 1, 0);
  const double c_r1 = L.shells[jc].r1, c_ro = L.shells[jc].r1_outer;
  const double b_r1 = L.shells[jb].r1, b_ro = L.shells[jb].r1_outer;
  const bool step = (j0 < nS) & !(c_r1 > radial);
  const int j = j0 + (step ? 1 : 0);
  const bool has_shell = j < nS;
  ok = ok & has_shell;   // radial == R1[last] exactly (measure zero; the reference then uses a zero shell)
  okm &= ballot64(has_shell);
  // glass front (:1942-1944): only the shell just below the selected one can contain radial (thickness < spacing, checked on
  // the host): the cell's shell after a step, the one below it otherwise
  const double g_r1 = step ? c_r1 : b_r1, g_ro = step ? c_ro : b_ro;
  const bool gf_a = j > 0, gf_b = radial > g_r1, gf_c = radial < g_ro;
  ok = ok & !(gf_a & gf_b & gf_c);
  // (mask form: both candidates compared, the choice between them made on the masks - four compares instead of four selects
  // and two compares)
  const uint64_t step_m = ballot64(j0 < nS) & ~ballot64(c_r1 > radial);
  const uint64_t in_c = ballot64(radial > c_r1) & ballot64(radial < c_ro), in_b = ballot64(radial > b_r1) & ballot64(radial < b_ro);
  okm &= ~(ballot64(gf_a) & ((step_m & in_c) | (~step_m & in_b)));
  M.ok = okm;
  st.shell = min(j, nS - 1);
  radial_out = radial;
  return ok;
}

// Both halves back to back (histogram and record kernels).
template <bool FAST, int ROT, bool ZEXT, bool NOWALL = false>
__device__ __forceinline__ bool phase_a_core(const HotA& H, const DevParams& P, const LdsTables& L, const Uniforms& U, RayState& st,
                                             bool& sampled, bool& reached, double& radial_out, LaneMasks& M) {
  static_assert(!ZEXT || (FAST && ROT == 0), "the z-extent form needs the magnet-frame slopes in phase B");
  static_assert(!NOWALL || ZEXT, "the constant-path form is a specialisation of the vacuum, unrotated one");
  BoreRay br;
  phase_a_bore<FAST, ZEXT, NOWALL>(H, P, L, U, st, sampled, reached, br, M);
  return phase_a_telescope<FAST, ROT>(H, P, tel_rot_of(P), L, br, st, radial_out, M);
}

// Phase A of the ray with global id `ray_id`: its six uniforms from ONE Philox counter block (x, y, z, w) + its word s of the
// shared stream (sart_oracle_uniforms) - 160 random bits, none used twice:
//   u2 (radius CDF draw)  = x / 2^32            u5 (energy CDF draw) = z / 2^32          u3 (disc radius) = s / 2^32
//   u0 (solar point, theta1) = (y >> 11) / 2^21    u1 (theta2) = (w >> 11) / 2^21
//   u4 (disc angle) = ((y & 0x7FF) << 11 | (w & 0x7FF)) / 2^22
// 2^-21 of a turn is 3e-6 rad of position on a source 4.6e-3 rad wide seen from the magnet; 2^-22 of a turn on the bore-exit disc
// is under a micrometre of arc; the CDF draws resolve 2^-32 where the tables' entries are 5e-4 (radius) and ~7e-4 (energy) apart.
// u3_hi = word ray_id of the shared stream.
__device__ __forceinline__ Uniforms uniforms_of(uint32_t seed_lo, uint32_t seed_hi, uint64_t ray_id, uint32_t u3_hi) {
  const uint32_t id_lo = (uint32_t)ray_id, id_hi = (uint32_t)(ray_id >> 32);
  const U4 b0 = philox4x32(id_lo, id_hi, 0u, 0u, seed_lo, seed_hi);
  Uniforms U;
  U.u2 = u52(b0.x, 0u);
  U.u2_hi = b0.x;   // floor(u2 2^32) is the word itself
  U.u5 = u52(b0.z, 0u);
  U.u0 = u52(b0.y & 0xFFFFF800u, 0u);
  U.u1 = u52(b0.w & 0xFFFFF800u, 0u);
  U.u4 = u52((b0.y << 21) | ((b0.w & 0x7FFu) << 10), 0u);
  U.u3 = u52(u3_hi, 0u);
  return U;
}
template <bool FAST, int ROT, bool ZEXT, bool NOWALL = false>
__device__ __forceinline__ bool phase_a(const HotA& H, const DevParams& P, const LdsTables& L, uint32_t seed_lo,
                                        uint32_t seed_hi, uint64_t ray_id, uint32_t u3_hi, RayState& st, bool& sampled,
                                        bool& reached, double& radial, LaneMasks& M) {
  return phase_a_core<FAST, ROT, ZEXT, NOWALL>(H, P, L, uniforms_of(seed_lo, seed_hi, ray_id, u3_hi), st, sampled, reached, radial, M);
}

// z of pointExitCB in the rotated telescope frame (z0 of :2051) from the ray as phase B knows it: the point of the ray whose z
// *before* the rotation (rotateInY(rotateInX(.)) about (0, 0, lT/2), :1888-1894) is -Lp.  With m = third column of that rotation
// and q(Z) = (qx + tsx Z, qy + tsy Z, Z), qx / qy = X0 / Y0 + entrance offset:  m . (q - c) + lT/2 = -Lp, linear in Z.
__device__ __forceinline__ double zcb_rotated(double mx, double my, double mz, double h, double qx, double qy, double tsx, double tsy, double Lp) {
  const double num = -Lp - h - fma(mx, qx, fma(my, qy, -mz * h));
  return num * frcp(fma(mx, tsx, fma(my, tsy, mz)));
}

// Results of phase B for one ray (record mode needs all of them; histogram mode a few).
struct RayOut {
  // lane masks (ballots taken where the predicates are computed: a bool carried across basic blocks is materialised as a
  // per-lane 0 / 1 and compared again)
  uint64_t m_nickel = 0, m_till = 0, m_passed = 0;
  bool passed = false;      // finished && weight != 0
#ifdef SART_STAGE_TIMING
  uint64_t tb[6] = {0, 0, 0, 0, 0, 0};
#endif
  double px = 0.0, py = 0.0, rdet = 0.0, weight = 0.0, reflect = 0.0;
  int e_idx = 0;
  GasRay gas = {0.0, 0.0, 0.0, 0.0, 0.0};   // SCAN: what the per-mass conversion probability needs of this ray
};

// ------------------------------------------------------------------------------------------------
// phase B: mirrors -> detector plane -> weights -> window (:1971-2221), predicated like phase A:
// `live` carries the reference's early returns.  Dead lanes keep computing (indices are clamped so that
// every table access stays in range); only record fields and the final outputs look at `live`.
// ------------------------------------------------------------------------------------------------
// GAS: stage known at compile time (0 vacuum / 1 gas) or read at run time (-1).  ZEXT: see phase_a.
// SCAN (fused mass scan): the gas-stage conversion probability is left out - out.weight is the mass-independent factor of the
// weight, out.gas what gas_conversion_prob() needs of this ray, out.m_passed the rays on the chip for which that factor is not
// zero; the caller applies the probability per mass.
// NODRAW: the caller always hands the energy index in (fused angular scan: drawn once per ray in front of the angle loop).
template <bool RECORDS, bool FAST, int GAS, bool ZEXT, bool SCAN = false, bool NODRAW = false>
__device__ __forceinline__ void phase_b(const DevParams& P, const LdsTables& L, const HotB& HB, const TraceArgs& A,
                                        const RayState& st, int e_idx_in, bool live, RayOut& out, sart_axion_t* rec) {
#ifdef SART_STAGE_TIMING
#define SART_B_STAMP(k, dep) do { asm volatile("" :: "v"(dep)); out.tb[k] = __builtin_readcyclecounter(); } while (0)
#else
#define SART_B_STAMP(k, dep)
#endif
  SART_B_STAMP(0, st.X0);
  // the energy draw runs beside the mirror arithmetic (its gathers are issued early, consumed late)
  const bool draw_energy = NODRAW ? false : __builtin_amdgcn_readfirstlane(e_idx_in) < 0;   // wave-uniform: false for the X-ray test source
  EnergyDraw ed = {};
#ifdef SART_DEBUG_KNOBS
  // working-set experiments (wrong results by design): 0x04000000 guide rows folded onto 16, 0x02000000 CDF rows folded onto 16,
  // 0x01000000 reflectivity / energy rows folded onto 32
  const uint32_t dbg = (uint32_t)__builtin_amdgcn_readfirstlane((int)A.flags);
  if (draw_energy) energy_draw_begin(HB, (dbg & 0x02000000u) ? (st.r_idx & 15) : st.r_idx, st.u5, ed, (dbg & 0x04000000u) ? (st.r_idx & 15) : st.r_idx);
#else
  if (draw_energy) energy_draw_begin(HB, st.r_idx, st.u5, ed);
#endif
  const ShellDev& sh = L.shells[st.shell];
  // P / A may live in LDS: branch conditions are made wave-uniform (scalar branches) explicitly.
  // FAST: vacuum stage, solar source (known at compile time).
  const int wolter = __builtin_amdgcn_readfirstlane(P.telescope_wolter);
  static_assert(!ZEXT || GAS >= 0, "the z-extent form needs the stage at compile time");
  static_assert(!SCAN || !RECORDS, "the mass scan accumulates, it writes no records");
  const int stage_gas = (GAS >= 0) ? GAS : __builtin_amdgcn_readfirstlane(P.stage_gas);
  const int test_active = FAST ? 0 : __builtin_amdgcn_readfirstlane(P.test_active);
  const int n_half_strips = __builtin_amdgcn_readfirstlane(P.n_half_strips);
  const uint32_t flags = (uint32_t)__builtin_amdgcn_readfirstlane((int)A.flags);
  const double X0 = st.X0, Y0 = st.Y0, tsx = st.tsx, tsy = st.tsy, zcb = st.zcb;
  const double Q0 = fma(X0, X0, Y0 * Y0);

  // ---- mirror 1 (:1985-1992 / :2012-2019) ----
  const double D0 = fma(X0, tsx, Y0 * tsy);
  const double A0 = fma(tsx, tsx, tsy * tsy);
  const double L0 = 1.0 + A0;                 // |w|^2 of the un-normalised direction w = (tsx, tsy, 1)
  double z1;
  uint64_t live_m = ballot64(live), hit1_m, hit2_m;   // `live` and the cuts below as lane masks, combined on the scalar unit
  const bool hit1 = pick_root(A0 - sh.m1_k, D0 + sh.m1_hb, Q0 - sh.m1_cc, 0.0, sh.m1_zhi, z1, hit1_m);
  if (!hit1) z1 = zcb;                        // the reference returns its input point (:656-658)
  const double m1x = fma(z1, tsx, X0), m1y = fma(z1, tsy, Y0);
  // on the surface the normal's z-component is closed-form: cone tan(b) rho(z); paraboloid r3 tan(b)
  double n1z = wolter ? sh.n1_r3t : sh.n1_tan * fma(-sh.n1_tan, z1, sh.r1);
  if (live & !hit1)                           // divergent, skipped when no lane needs it: the normal at the (off-surface) input point
    n1z = normal_z_general(P, sh, 1, m1x, m1y, z1);
  if (draw_energy) energy_draw_candidates(HB, ed);   // the guide word has had the first mirror's arithmetic to arrive
  double wx = tsx, wy = tsy, wz = 1.0;
  const double N1 = fma(m1x, m1x, fma(m1y, m1y, n1z * n1z));
  double sin_a1, sin_a2;
  const double sin2_a1 = reflect(wx, wy, wz, L0, m1x, m1y, n1z, N1, sin_a1);

  // lineHitsNickel (:1706-1734), evaluated before the no-hit test (:2040-2057):
  // tan(a1) > num / (l - z1)  <=>  sin^2(a1) ((l - z1)^2 + num^2) > num^2   (num >= 0, l - z1 > 0)
  {
    const double lz = P.l_mirror - z1, num = sh.nickel_num;
    bool nick = sin2_a1 * fma(lz, lz, num * num) > num * num;
    uint64_t nick_m = ballot64(nick);
    if (ballot64(!(lz > 0.0))) {               // wave-uniform; only for a missed mirror with z1 >= l (never in practice)
      if (!(lz > 0.0)) nick = fsqrt(sin2_a1 / (1.0 - sin2_a1)) > num / lz;
      nick_m = ballot64(nick);
    }
    const bool upper = st.shell > 0;
    const bool hit_nickel = live & upper & nick;
    out.m_nickel = live_m & ballot64(upper) & nick_m;
    if (RECORDS && hit_nickel) rec->hitNickel = 1;
    live = live & !hit_nickel & hit1;         // nickel (:2045), almostEqual(z1, z0) (:2055)
    live_m = live_m & ~out.m_nickel & hit1_m;
  }

  SART_B_STAMP(1, wz);
  // ---- mirror 2 (:1994-2001 / :2021-2028): ray through (m1x, m1y, z1) along w ----
  const double inv_wz = frcp(wz);
  const double s2x = wx * inv_wz, s2y = wy * inv_wz;
  const double X1 = fma(-z1, s2x, m1x), Y1 = fma(-z1, s2y, m1y);   // extrapolated to z = 0
  const double A1 = fma(s2x, s2x, s2y * s2y);
  const double D1 = fma(X1, s2x, Y1 * s2y);
  const double Q1 = fma(X1, X1, Y1 * Y1);
  double z2;
  const bool hit2 = pick_root(A1 - sh.m2_k, D1 + sh.m2_hb, Q1 - sh.m2_cc, sh.m2_zlo, sh.m2_zhi, z2, hit2_m);
  live = live & hit2;                         // almostEqual(z1, z2) (:2055)
  live_m &= hit2_m;
  if (!hit2) z2 = sh.m2_zlo;
  const double m2x = fma(z2, s2x, X1), m2y = fma(z2, s2y, Y1);
  const double n2z = wolter ? sh.n2_r3t * fma(2.0 * (P.l_mirror - z2), sh.n2_invF, 1.0)
                            : sh.n2_tan * fma(-sh.n2_tan, z2 - sh.m2_zlo, sh.m2_rc);
  const double N2 = fma(m2x, m2x, fma(m2y, m2y, n2z * n2z));
  const double sin2_a2 = reflect(wx, wy, wz, L0, m2x, m2y, n2z, N2, sin_a2);

  SART_B_STAMP(2, wz);
  // ---- reflectivity lookups, angle part (computeReflectivity :1533-1580: bilinear in (angle, energy); the energy
  // interpolation is folded into the per-energy-index table, leaving a linear interpolation in the angle).  The cell and
  // the fraction inside it do not depend on the energy index: computed while the CDF candidates are in flight ----
  const bool use_refl = !(flags & SART_CF_IGNORE_REFLECTION);
  int ia1 = 0, ia2 = 0;
  double xu1 = 0.0, xu2 = 0.0;
  if (use_refl) {
    const int na2 = HB.refl_n_angles - 2;
    const double amin = P.refl_angle_min, inv_da = P.refl_inv_dangle, da = P.refl_dangle;
    auto angle_cell = [&](double sina, double sin2a, double& xu_out) {
      const double alpha = asin_small(sina, sin2a) * 57.29577951308232;   // getMirrorAngle (:782-795), degrees
      const double t = (alpha - amin) * inv_da;
      int i = (int)t;                 // = floor(t) for t >= 0; negative or NaN t ends in cell 0 through the clamp
      i = max(min(i, na2), 0);
      xu_out = (alpha - fma((double)i, da, amin)) * inv_da;
      return i;
    };
    ia1 = angle_cell(sin_a1, sin2_a1, xu1);
    ia2 = angle_cell(sin_a2, sin2_a2, xu2);
  }

  // ---- energy index; the energy row and the two reflectivity pairs are requested the moment it is known and consumed
  // behind the detector-plane and window geometry ----
#ifdef SART_DEBUG_KNOBS
  const int e_idx = (dbg & 0x01000000u) ? ((draw_energy ? energy_draw_finish(HB, ed) : e_idx_in) & 31) : (draw_energy ? energy_draw_finish(HB, ed) : e_idx_in);
#else
  const int e_idx = draw_energy ? energy_draw_finish(HB, ed) : e_idx_in;
#endif
  SART_B_STAMP(3, e_idx);
  const EnergyDev en = load_energy_row(HB, e_idx);
  d2 g1 = {1.0, 1.0}, g2 = {1.0, 1.0};
  if (use_refl) {
    // row (coating, e_idx) of refl[][n_angles]: 32-bit element offset (the table is < 4 GB, checked on the host)
    const uint32_t row = __umul24((uint32_t)(sh.refl_row0 + e_idx), (uint32_t)HB.refl_n_angles);
    g1 = gload<d2>(HB.refl, (row + (uint32_t)ia1) * 8u);   // g[i], g[i + 1] (8-byte aligned pair)
    g2 = gload<d2>(HB.refl, (row + (uint32_t)ia2) * 8u);
  }

  // ---- detector plane: getPointDetectorWindow (:797-814, :2070-2083) ----
  const double pmx = fma(m2x, P.pipe_c, z2 * P.pipe_s) - P.d_cb_xray;
  const double pmz = fma(z2, P.pipe_c, -m2x * P.pipe_s);
  const double vx = fma(wx, P.pipe_c, wz * P.pipe_s);
  const double vz = fma(wz, P.pipe_c, -wx * P.pipe_s);
  const double inv_vz = frcp(vz);
  const double nwin = (sh.dist_det - pmz) * inv_vz;
  double pdx = fma(nwin, vx, pmx), pdy = fma(nwin, wy, m2y), pdz = sh.dist_det;

  // yaw angle (:2101-2115): ya = deg(atan2(-v_z, -v_y)) + 90 = -deg(atan(v_y / v_z)); only its cosine is used unless
  // the record is written
  const double ya = RECORDS ? -atan_small(tsy) * 57.29577951308232 : 0.0;
  const double cos_ya = RECORDS ? cos_small(ya) : cos_yaw_of_slope(tsy);

  if (RECORDS && live) {
    const double nend = (sh.dist_det_end - pmz) * inv_vz;
    const double ddx = (nend - nwin) * vx, ddy = (nend - nwin) * wy;
    rec->deviationDet = sqrt(fma(ddx, ddx, ddy * ddy));           // :2085-2088
    rec->pointdataXBefore = X0;                                    // :2091-2094
    rec->pointdataYBefore = Y0;
    rec->pixvalsX = floor(X0 / (48.0 / 1400.0)) + 700.0;           // getPixelValue (:618-623)
    rec->pixvalsY = floor(Y0 / (48.0 / 1400.0)) + 700.0;
  }

  const double distance_pipe_m = (pdz - zcb) * 1e-3;               // :2116 (before the straight-through override of pdz)
  if (test_active) {   // straight through the hole in the optics (:2130-2132)
    const bool through = (sh.r1 - fsqrt(Q0)) > 100.0;
    pdx = through ? fma(sh.dist_det_raw, tsx, X0) : pdx;
    pdy = through ? fma(sh.dist_det_raw, tsy, Y0) : pdy;
    pdz = through ? sh.dist_det_raw : pdz;
  }
  pdx -= P.lateral_shift;
  pdy -= P.transversal_shift;

  // ---- geometry of the detector window / chip (:2138-2147) and of the window strips (:2149-2187: rotateAroundZ by theta,
  // strips along x): none of it needs the gathers, so it runs before they are consumed ----
  const double rdet2 = fma(pdx, pdx, pdy * pdy);
  const bool off_window = rdet2 > P.radius_window_sq, off_x = fabs(pdx) > P.chip_cx, off_y = fabs(pdy) > P.chip_cy;
  const bool on_chip = !((!(flags & SART_CF_IGNORE_DET_WINDOW)) & off_window) & !(off_x | off_y);
  const uint64_t on_chip_m = ~(((flags & SART_CF_IGNORE_DET_WINDOW) ? 0ull : ballot64(off_window)) | ballot64(off_x) | ballot64(off_y));
  const double yt = fabs(fma(pdy, P.theta_c, -pdx * P.theta_s));
  bool in_strip = false;
  for (int i = 0; i < n_half_strips; ++i) in_strip = in_strip | ((yt > P.strip_lo[i]) & (yt < P.strip_hi[i]));

  // ---- weights (:2116-2128) ----
  const double path_cb = st.path_cb;
  double trans_magnet;
  // Gas stage, accumulating kernels: the conversion probability is the one factor that depends on the axion mass; it is
  // applied LAST (below; per mass by the caller when SCAN), trans_magnet goes without it.  Wave-uniform.
  bool prob_deferred = false;
  {
    double prob = 1.0;
    double absorb = 1.0;
    if (!stage_gas) {
      // conversionProb (:363-365) = conv_k pathCB^2; FAST carries the z extent of the path: pathCB^2 = z^2 (1 + slope^2)
      if (!(flags & SART_CF_IGNORE_CONV_PROB)) prob = ZEXT ? (P.conv_k * L0) * (path_cb * path_cb) : P.conv_k * path_cb * path_cb;
    } else {
      // axionConversionProb2 / intensitySuppression2 (axionMassforMagnet.nim:75-113) with pathCB as length
      const double path_len = ZEXT ? path_cb * fsqrt_pos(L0) : path_cb;   // ZEXT: the ring carries the z extent of the path
      const double Lnat = path_len * P.gas_inv_hbarc_m;              // length / 1.97e-7, length in m
      if (!(flags & SART_CF_IGNORE_CONV_PROB)) {
        const double g = en.gamma;
        const double eh = exp_neg(-g * Lnat * 0.5);                    // exp(-Gamma L) = eh^2: one exp for both terms
        out.gas = GasRay{en.inv_two_e_ev, g * g * 0.25, -2.0 * eh, fma(eh, eh, 1.0), Lnat};
        if (RECORDS) prob = gas_conversion_prob(P.gas_dm2_abs, out.gas, P.gas_term1, L.sincos, GasCos::make());
        else prob_deferred = true;
      }
      // intensitySuppression2 (axionMassforMagnet.nim:100-113): exp(-mu_pipe d) exp(-mu_magnet L) as one exponential
      absorb = exp_neg(-fma(en.mu_pipe, distance_pipe_m, en.mu_magnet * (path_len * 1e-3)));
    }
    trans_magnet = cos_ya * prob * absorb;          // cos of a degree value taken as radians — sic (:1598)
  }
  // R(alpha, E_idx) = g[i] + xUnit (g[i + 1] - g[i]) for both mirrors (with ignoreReflection: g = (1, 1))
  const double reflectv = fma(xu1, g1.y - g1.x, g1.x) * fma(xu2, g2.y - g2.x, g2.x);
  double weight = reflectv * trans_magnet;
  SART_B_STAMP(4, weight);
  if (RECORDS && live) {
    rec->transmissionMagnet = trans_magnet;
    rec->yawAngles = ya;
    rec->reflect = reflectv;
  }

  const bool has_weight = weight != 0.0;
  const bool till_window = live & has_weight;
  out.m_till = live_m & ballot64(has_weight);   // (a deferred conversion probability of exactly zero is taken out below)
  if (RECORDS && till_window) rec->passedTillWindow = 1;
  live = live & on_chip;
  live_m &= on_chip_m;

  const double trans_window = (n_half_strips > 0) ? (in_strip ? en.t_strongback : en.t_window) : 0.0;
  const uint8_t kind_w = in_strip ? SART_MK_SI : SART_MK_SI3N4;
  // wave-uniform switches as scalar branches around one multiplication each (the empty asm keeps LLVM from turning them
  // into a multiplication plus a two-instruction select that every launch pays)
  if (!(flags & SART_CF_IGNORE_DET_WINDOW)) { asm volatile(""); weight *= trans_window; }
  if (!(flags & SART_CF_IGNORE_GAS_ABS)) { asm volatile(""); weight *= en.a_gas; }        // :2190-2192
  if (!(flags & SART_CF_XRAY_TEST)) { asm volatile(""); weight *= P.exposure; }           // :2207-2212
  if (!SCAN && prob_deferred) {
    const double prob = gas_conversion_prob(P.gas_dm2_abs, out.gas, P.gas_term1, L.sincos, GasCos::make());
    weight *= prob;
    out.m_till &= ballot64(prob != 0.0);
  }

  SART_B_STAMP(5, weight);
  const bool final_weight = weight != 0.0;
  out.m_passed = live_m & ballot64(final_weight);
  out.passed = RECORDS ? (live & final_weight) : __builtin_amdgcn_inverse_ballot_w64(out.m_passed);
  out.reflect = reflectv;
  out.e_idx = e_idx;
  out.rdet = fsqrt_pos(rdet2 + 1e-300);       // + 1e-300: exact no-op unless the ray hits the chip centre to the last bit
  out.px = -pdx + P.chip_cx;                                          // :2203-2204
  out.py = pdy + P.chip_cy;
  out.weight = weight;
  if (RECORDS && live) {
    if (n_half_strips > 0) {
      rec->transProbWindow = trans_window;
      rec->energiesAxWindow = en.energy;
      rec->kindsWindow = kind_w;
    }
    rec->transProbArgon = en.a_gas;
    rec->transProbDetector = en.a_gas;
    rec->energiesAxAll = en.energy;
    rec->kinds = SART_MK_AR;
    rec->energiesAx = en.energy;
    rec->shellNumber = st.shell;
    rec->pointdataR = out.rdet;
    rec->pointdataX = out.px;
    rec->pointdataY = out.py;
    rec->weights = weight;
    rec->weightsAll = weight;
    rec->passed = (weight != 0.0) ? 1 : 0;
  }
}

// ------------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------------
// Wave priorities by pipeline stage (s_setprio: the SIMD's arbiter picks the ready wave with the highest priority): the four
// persistent waves of a SIMD are in different stages at any moment, and the later the stage the sooner its instructions should
// go - a wave in phase B has three dependent table gathers to get into flight (everything it issues ahead of them is latency
// the others could have hidden), a wave in phase A feeds it, the accumulation behind phase B (atomics nobody waits for) and
// stage A0 are filler that runs whenever nobody else can.  Priced in profiles/r05_exp_setprio.txt (same-box A/B of eleven
// assignments, bitwise the same results): B 3 > A1 2 > accumulation 1 > A0 / loop glue 0 is worth 9 % on BabyIAXO / XMM, 9 % on
// the gas stage and the rotated telescope, 8 % on CAST / LLNL against every wave at priority 0 (oldest first); what matters
// most is stage A0 below everything else (A1 = B = 1: 6 %), then B above A1 (+3 %).
#ifndef SART_PRIO_A1
#define SART_PRIO_A1 2
#endif
#ifndef SART_PRIO_B
#define SART_PRIO_B 3
#endif
#ifndef SART_PRIO_ACC
#define SART_PRIO_ACC 1
#endif
constexpr int kQueue = 128;   // ring capacity per wave: < 64 left over + <= 64 new survivors

struct __align__(16) TablesLds {
  double sincos[2 * kSinCosEntries];
  uint32_t rcdf_hi[kMaxRadii + 4];   // (floor(fluxRadiusCDF 2^52) >> 20) + 1, saturated; + four entries of 1.0 behind the table (four-wide candidate read)
  ShellDev shells[kMaxShells];
  uint16_t rguide[kRadiusGuideEntries + 6];
  uint8_t lut[kShellLutMax];
};

// per-wave survivor rings (structure of arrays: lane i reads slot (head + i) % 128 -> conflict-free)
// One wave's rings side by side (7.7 KB): every field is then reachable from a single per-wave base address with an
// immediate offset, instead of one base register per array.
struct __align__(16) WaveRings {
  double X0[kQueue], Y0[kQueue], tsx[kQueue], tsy[kQueue];
  double path[kQueue], u5[kQueue];
  int idx[kQueue];                        // r_idx | shell << 16
  uint32_t ray[kQueue];                   // ring 0: rays that passed stage A0 (ray id relative to the launch's first chunk)
  uint32_t u3hi[kQueue];                  //         and their word of the shared stream (high word of u3)
};
template <int WAVES>
struct __align__(16) QueueLds {
  WaveRings w[WAVES];
};

template <int BLOCK>
__device__ __forceinline__ void stage_tables(TablesLds& S, const DevParams& P, const DevTables& T) {
  for (int i = threadIdx.x; i < 2 * kSinCosEntries; i += BLOCK) S.sincos[i] = as_global(T.sincos_tab)[i];
  for (int i = threadIdx.x; i < P.n_radii + 4; i += BLOCK) {   // (cdf_hi32_kernel's definition, sart_tables.hip)
    const double c = (i < P.n_radii) ? as_global(T.flux_radius_cdf)[i] : 1.0;
    const unsigned long long t = (__double2ull_rd(c * 4503599627370496.0) >> 20) + 1ull;   // the product is exact: cdf <= 1
    S.rcdf_hi[i] = t > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)t;
  }
  for (int i = threadIdx.x; i < kRadiusGuideEntries; i += BLOCK) S.rguide[i] = as_global(T.radius_guide)[i];
  {
    const auto src = as_global(reinterpret_cast<const uint64_t*>(T.shells));
    uint64_t* dst = reinterpret_cast<uint64_t*>(S.shells);
    const int n = P.n_shells * (int)(sizeof(ShellDev) / 8);
    for (int i = threadIdx.x; i < n; i += BLOCK) dst[i] = src[i];
  }
  for (int i = threadIdx.x; i < P.lut_n; i += BLOCK) S.lut[i] = as_global(T.shell_lut)[i];
  __syncthreads();
}

// Fused trace + accumulate (traceAxionWrapper + prepareHeatmap + flux sum + counters) as a three-stage
// pipeline inside one persistent wave, with wavefront compaction (ballot + prefix count into per-wave LDS
// rings) between the stages:
//   A0  word of the shared stream -> radius of the point on the bore exit -> provably dead rays leave (HotA zones)
//   A1  phase A (full sampling + cuts + shell selection) on full waves of A0 survivors
//   B   phase B (mirrors + weights + accumulation) on full waves of A1 survivors
// Rays are taken in chunks of 256 consecutive global ids, aligned to 256: lane l of the wave that owns a chunk handles
// ids 4l .. 4l+3 of it in four successive passes, so that the one block of the shared word stream it computes (stage A0's
// only Philox block) serves all four.  Ray i of this launch has the global id ray_id_offset + i; n_rays < 2^31 per launch.
// FAST: solar source, vacuum, no hole loop, telescope not rotated (all known at compile time).  ROT: rotated telescope
// (phase B then recomputes z of pointExitCB from the ray instead of carrying it through ring 1).
// Re-reads the HotA block (the first kernel argument) from the kernel-argument segment with scalar loads.  The pointer
// is laundered so that the loads stay where they are written (the start of a phase-A pass) instead of being hoisted
// out of the persistent loop: ~50 SGPRs held across stage B get spilled through VGPR lanes (v_readlane = VALU slots).
__device__ __forceinline__ void reload_hot(HotA& dst) {
  typedef const __attribute__((address_space(4))) uint32_t* kernarg_ptr;
  kernarg_ptr p = (kernarg_ptr)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(p));
  uint32_t* d = reinterpret_cast<uint32_t*>(&dst);
#pragma unroll
  for (int k = 0; k < (int)(sizeof(HotA) / 4); ++k) d[k] = p[k];
}

// Same for the zone table of stage A0 (the tail of HotA): ten scalar registers that would otherwise be spilled and
// reloaded through VGPR lanes in every pass.
struct ZoneTable {
  int32_t n_zones;
  uint32_t zone_reached;
  uint32_t lo[kMaxZones], hi[kMaxZones];
};
static_assert(sizeof(ZoneTable) == sizeof(HotA) - offsetof(HotA, n_zones), "ZoneTable mirrors the tail of HotA");
static_assert(offsetof(WaveRings, u3hi) == offsetof(WaveRings, ray) + kQueue * sizeof(uint32_t) && offsetof(WaveRings, ray) % 8 == 0,
              "the image tile uses ray + u3hi of every wave as 128 contiguous doubles");
__device__ __forceinline__ void reload_zones(ZoneTable& dst) {
  typedef const __attribute__((address_space(4))) uint32_t* kernarg_ptr;
  kernarg_ptr p = (kernarg_ptr)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(HotA, n_zones) / 4;
  asm volatile("" : "+s"(p));
  uint32_t* d = reinterpret_cast<uint32_t*>(&dst);
#pragma unroll
  for (int k = 0; k < (int)(sizeof(ZoneTable) / 4); ++k) d[k] = p[k];
}

// The argument block of trace_histogram_kernel as the code object lays it out (arguments in order, each at its natural
// alignment; tests/test_host_and_abi.py::test_kernel_argument_layout_matches_the_code_object compares these offsets with the
// code object's metadata on every run of the CPU suite).
struct HistKernArgs {
  HotA H;
  const DevBlob* blob;
  TraceArgs A;
  double* acc;
  HotB HB;
  ScanArgs SC;
};
static_assert(offsetof(HistKernArgs, H) == 0 && offsetof(HistKernArgs, blob) == (sizeof(HotA) + 7) / 8 * 8 &&
                  offsetof(HistKernArgs, A) == offsetof(HistKernArgs, blob) + 8 &&
                  offsetof(HistKernArgs, acc) == offsetof(HistKernArgs, A) + sizeof(TraceArgs) &&
                  offsetof(HistKernArgs, HB) == offsetof(HistKernArgs, acc) + 8 &&
                  offsetof(HistKernArgs, SC) == offsetof(HistKernArgs, HB) + sizeof(HotB) && sizeof(HotA) % 4 == 0 &&
                  sizeof(TraceArgs) % 8 == 0 && sizeof(HotB) % 8 == 0,
              "reload_hot / reload_zones / reload_kernarg read the arguments at these offsets");
// Re-reads one argument (or a leading part of it) from the kernel-argument segment with scalar loads at the place of use.
template <typename T>
__device__ __forceinline__ void reload_kernarg(T& dst, size_t byte_offset) {
  static_assert(sizeof(T) % 4 == 0, "dword granularity");
  typedef const __attribute__((address_space(4))) uint32_t* kernarg_ptr;
  kernarg_ptr p = (kernarg_ptr)__builtin_amdgcn_kernarg_segment_ptr() + byte_offset / 4;
  asm volatile("" : "+s"(p));
  uint32_t* d = reinterpret_cast<uint32_t*>(&dst);
#pragma unroll
  for (int k = 0; k < (int)(sizeof(T) / 4); ++k) d[k] = p[k];
}

// Instantiations: <FAST, ROT, GAS> = <true, false, 0> the common configuration; <true, false, 1> gas stage (the m_a scan of
// BASELINE configs[4]); <true, true, 0> rotated telescope (the angular scan of configs[3]); <false, *, -1> everything else
// (X-ray test source, hole loop, rotated + gas) with the switches read at run time.
// PATHC (common configuration only): the host has proved that no ray that survives phase A entered through the bore wall
// (sart_api.hip: build_zones), so the path in the magnetic field is the constant lengthB for every ray of phase B: ring 1
// does not carry it, and its 128 doubles per wave hold the LDS image tile instead (with stage A0 running beside it).
// FIXED: SART_ACCUM_FIXED64 - every sum is accumulated as an integer multiple of its quantum (to_fixed), with integer
// atomics in LDS and global memory; the slots of the replicas, of the partials and of the accumulator then hold int64.
// SCAN: fused axion-mass scan (sart_trace_mass_scan).  Stage B leaves the gas-stage conversion probability out of the weight and
// a loop over the launch's masses (ScanArgs: kernel arguments, read with scalar loads) applies it per mass:
// sum of w and of w^2 go to per-lane accumulators [mass][2][64] in the 16 KB of LDS that hold the image tile otherwise (ds_add_f64 /
// ds_add_u64 across the 16 waves of the workgroup); no image, no position sums.  Without PATHC that space is ring 0, so stage
// A0 stays off for such a launch whatever zones the host built.
template <int BLOCK, bool FAST, bool ROT, int GAS, bool PATHC, bool FIXED, bool SCAN = false>
__global__ __launch_bounds__(BLOCK) void trace_histogram_kernel(HotA H, const DevBlob* __restrict__ blob, TraceArgs A,
                                                                double* __restrict__ acc, HotB HBarg, ScanArgs SCarg) {
  // One LDS object with the tables FIRST: their addresses then fit the 16-bit offset field of the ds_ instructions, and a
  // lookup is `ds_read v, v_index_scaled offset:TABLE` instead of a literal moved into a register and added to the index
  // (the rings are addressed from a per-wave scalar base anyway).
  struct LdsLayout {
    TablesLds S;
    double tile_extra[kTileExtraCells];   // cells kTileRingCells .. of the image tile
    DevBlob B;
    TraceArgs Ab;
    QueueLds<BLOCK / 64> Q;
  };
  __shared__ LdsLayout lds;
  TablesLds& S = lds.S;
  QueueLds<BLOCK / 64>& Q = lds.Q;
  static_assert(!PATHC || (FAST && !ROT && GAS >= 0), "the constant-path form belongs to the unrotated specialisations");
  static_assert(!SCAN || GAS != 0, "a mass scan needs the gas stage");
  static_assert(!SCAN || kScanMaxMasses * 2 * kScanLanes <= (BLOCK / 64) * kQueue, "the scan accumulators live in the image tile's 128 doubles per wave");
  __shared__ uint32_t scan_zero[kScanMaxMasses];   // SCAN: rays whose weight vanishes for one mass only (conversion probability exactly 0)
  static_assert((BLOCK / 64) * kQueue == kTileRingCells && kImageTileMax * kImageTileMax <= kTileRingCells + kTileExtraCells &&
                    kImageTileExtraMax * kImageTileExtraMax <= kTileExtraCells,
                "the LDS image tile (host: kImageTileMax): 128 doubles per wave of this workgroup's rings + the cells behind the tables");
  static_assert((offsetof(LdsLayout, Q) % 512) == 0, "the rings are addressed with ds_*2st64 offsets (units of 512 bytes for 64-bit columns)");
  // cell t < kTileRingCells of the tile space: 128 doubles per wave, in the space of ring 0 (stage A0 off) or of ring 1's path
  // column (PATHC); the scan's accumulators live there
  auto tile_cell_lo = [&](uint32_t t) -> double* {
    return PATHC ? &Q.w[t >> 7].path[t & 127u] : reinterpret_cast<double*>(&Q.w[t >> 7].ray[0]) + (t & 127u);
  };
  // cell t of the LDS image tile: the ring space first, then the cells behind the tables
  auto tile_cell = [&](uint32_t t) -> double* {
    return t < (uint32_t)kTileRingCells ? tile_cell_lo(t) : &lds.tile_extra[t - (uint32_t)kTileRingCells];
  };
  // Only the ~20 scalars phase A needs for every ray travel in the kernel arguments (SGPRs); everything
  // else is read from an LDS copy of the parameter blob (broadcast ds_read).  All of them together do not
  // fit the 102 SGPRs of a wave and would be spilled through VGPR lanes (v_readlane = VALU slots).
  DevBlob& B = lds.B;
  TraceArgs& Ab = lds.Ab;
  {
    const uint64_t* src = reinterpret_cast<const uint64_t*>(blob);
    uint64_t* dst = reinterpret_cast<uint64_t*>(&B);
    for (int i = threadIdx.x; i < (int)(sizeof(DevBlob) / 8); i += BLOCK) dst[i] = src[i];
    if (threadIdx.x == 0) Ab = A;
    __syncthreads();
  }
  const DevParams& Pb = B.P;
  const DevTables& Tb = B.T;
  {
    // zeroed rings: a slot that was never written reads as ray 0 / shell 0 / radius 0 / u = 0 instead of arbitrary bits
    uint64_t* q = reinterpret_cast<uint64_t*>(&Q);
    for (int i = threadIdx.x; i < (int)(sizeof(Q) / 8); i += BLOCK) q[i] = 0ull;
    for (int i = threadIdx.x; i < kTileExtraCells; i += BLOCK) lds.tile_extra[i] = 0.0;
    if (SCAN && threadIdx.x < kScanMaxMasses) scan_zero[threadIdx.x] = 0u;   // (the barrier of stage_tables orders both before their first use)
  }
  stage_tables<BLOCK>(S, Pb, Tb);
  const LdsTables L{S.sincos, S.rcdf_hi, Tb.flux_radius_cdf, S.rguide, S.shells, S.lut};

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave-uniform: scalar addressing of the rings
  const uint64_t waves_total = (uint64_t)gridDim.x * (BLOCK / 64);
  const uint64_t wave_global = (uint64_t)blockIdx.x * (BLOCK / 64) + wave;
  // wave-uniform; the host builds no zones for the X-ray test source.  (SCAN without PATHC: ring 0 holds the scan accumulators.)
  const bool early_reject = (SCAN && !PATHC) ? false : H.n_zones > 0;
  // chunks of 256 ids aligned in global-id space; `rel` = id - id_base (fits 32 bits: n_rays < 2^31)
  const uint64_t first_chunk = A.ray_id_offset >> 8;
  uint64_t id_base = first_chunk << 8;
  // a value of its own: computed in place it stays a part of the eight-register tuple the launch arguments were loaded into, and a
  // phase-A pass that wants it back from the spill lanes reloads all eight
  asm volatile("" : "+s"(id_base));
  const uint32_t rel_begin = (uint32_t)(A.ray_id_offset & 255u);
  const uint32_t rel_end = rel_begin + (uint32_t)A.n_rays;          // one past the last ray (n_rays < 2^31)
  const uint32_t n_chunks = (rel_end + 255u) >> 8;

  // wave-uniform counters (ballot + popcount) and per-lane sums
  uint32_t n_reached = 0, n_shell = 0, n_nickel = 0, n_till = 0, n_passed = 0;
  uint32_t n_outside = 0;    // per lane: passed rays outside the image
  using Sum = std::conditional_t<FIXED, long long, double>;   // per-lane sums: quanta (FIXED) or f64
  Sum sum_w = 0, sum_w2 = 0, sum_x = 0, sum_y = 0, sum_r = 0;
  long long sum_wo = 0;      // FIXED: weights of the passed rays outside the image (the finalize kernel's conservation check)
#ifdef SART_STAGE_TIMING   // diagnostic build (make STAGE_TIMING=1): shader-clock cycles per stage, summed over waves, in scalars 12..15
  uint64_t cyc_a0 = 0, cyc_a1 = 0, cyc_b = 0, cyc_bs[6] = {0, 0, 0, 0, 0, 0};
#endif
  uint32_t h0 = 0, t0 = 0;   // ring 0 (A0 -> A1) positions, monotone; slot = pos % kQueue
  uint32_t h1 = 0, t1 = 0;   // ring 1 (A1 -> B)

  // LDS accesses of one wave execute in order; the fences only keep the compiler from reordering them
  auto ring_sync = [] {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  auto prefix_of = [](uint64_t mask) {   // number of set mask bits below this lane
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
  };

  // stage A1 for the ray with id id_base + rel (valid lanes only count); u3_hi = its word of the shared stream
  auto run_phase_a = [&](uint32_t rel, bool valid, uint32_t u3_hi) {
    SART_STAGE_MARK("A1");
    __builtin_amdgcn_s_setprio(SART_PRIO_A1);
    RayState st;
    bool sampled = false, reached = false;
    double radial;
    HotA Hl;
    reload_hot(Hl);
    constexpr bool ZEXT = FAST && !ROT && GAS >= 0;
    LaneMasks M;
    (void)phase_a<FAST, ROT ? 1 : 0, ZEXT, PATHC>(Hl, Pb, L, A.seed_lo, A.seed_hi, id_base + (uint64_t)rel, u3_hi, st, sampled, reached, radial, M);
    const uint64_t valid_m = ballot64(valid);
    n_reached += (uint32_t)__popcll(valid_m & M.reached);
    const uint64_t selected = valid_m & M.ok;
    n_shell += (uint32_t)__popcll(selected);
    // Innermost shell, far enough below its first mirror (DevParams::shell0_miss_radius; -1 when the host could not prove it): phase
    // B would find no root inside the mirror and stop at the no-hit test without touching a counter.  Such a ray is counted
    // above and ends here.  Below the bound the selected shell is shell 0 (the bound lies below R1[0]): ONE compare against a
    // broadcast read of the parameter block in LDS, whose lane mask is combined with `selected` in scalar registers (the
    // ballot of a direct compare is the compare itself; a ballot of `a && b` costs a select and a second compare).
#ifdef SART_DEBUG_KNOBS
    const uint64_t mask = (A.flags & 0x10000000u) ? 0ull : (selected & ~ballot64(radial < Pb.shell0_miss_radius));   // experiment: nothing reaches phase B
#else
    const uint64_t mask = selected & ~ballot64(radial < Pb.shell0_miss_radius);
#endif
    const bool alive = __builtin_amdgcn_inverse_ballot_w64(mask);
    const uint32_t cnt = (uint32_t)__popcll(mask);
    if (alive) {
      asm volatile("; hot: ring 1 write");
      const uint32_t slot = (t1 + prefix_of(mask)) % kQueue;
      Q.w[wave].X0[slot] = st.X0; Q.w[wave].Y0[slot] = st.Y0;
      Q.w[wave].tsx[slot] = st.tsx; Q.w[wave].tsy[slot] = st.tsy;
      if (!PATHC) Q.w[wave].path[slot] = st.path_cb;
      Q.w[wave].u5[slot] = st.u5;
      Q.w[wave].idx[slot] = st.r_idx | (st.shell << 16);
    }
    t1 += cnt;
    __builtin_amdgcn_s_setprio(0);
  };

  auto run_phase_b = [&](uint32_t n_valid) {
    SART_STAGE_MARK("B");
    __builtin_amdgcn_s_setprio(SART_PRIO_B);
    RayState st;
    const bool valid = (uint32_t)lane < n_valid;
    const uint32_t slot = (h1 + (uint32_t)lane) % kQueue;
    RayOut out;
    {
      // slots beyond n_valid hold the state of earlier rays (or the zeros the rings start with): lanes compute on them
      // predicated off; every index they lead to is one a real ray produced
      st.X0 = Q.w[wave].X0[slot]; st.Y0 = Q.w[wave].Y0[slot];
      st.tsx = Q.w[wave].tsx[slot]; st.tsy = Q.w[wave].tsy[slot];
      st.path_cb = PATHC ? H.length_b : Q.w[wave].path[slot];   // PATHC: z extent of the path = lengthB for every ray
      st.u5 = Q.w[wave].u5[slot];
      if (!ROT) {
        st.zcb = -(H.dz3 - H.dz1);
      } else {
        // z of pointExitCB in the rotated frame, recomputed from the ray instead of carried through ring 1 (zcb_rotated)
        st.zcb = zcb_rotated(Pb.rx_s, -Pb.rx_c * Pb.ry_s, Pb.rx_c * Pb.ry_c, Pb.half_length_telescope, st.X0 + Pb.entrance_x,
                             st.Y0 + Pb.entrance_y, st.tsx, st.tsy, H.dz3 - H.dz1);
      }
      const int packed = Q.w[wave].idx[slot];
      st.r_idx = packed & 0xFFFF;
      st.shell = packed >> 16;
      const DevBlob& Bo = lds_opaque(B);
      HotB HB;
      reload_kernarg(HB, offsetof(HistKernArgs, HB));
      // (one round trip beside the ring reads above, not a second one in the middle of the pass)
      asm volatile("" :: "s"(HB.cdf_hi32), "s"(HB.energy_guide), "s"(HB.energy_tab), "s"(HB.refl), "s"(HB.refl_n_angles), "s"(HB.cdf_stride));
      phase_b<false, FAST, GAS, FAST && !ROT && GAS >= 0, SCAN>(Bo.P, L, HB, lds_opaque(Ab), st, (!FAST && H.test_active) ? Pb.n_energies : -1, valid, out, nullptr);
      if constexpr (SCAN) {
        SART_STAGE_MARK("SCAN");
        // (the per-mass loop stays at phase B's priority: 23.8 against 24.2 ms per 32-mass scan with it below phase A)
        // out.m_passed: rays on the chip whose mass-independent weight factor out.weight is not zero.  Per mass: weight =
        // out.weight x conversion probability (the single-mass kernels multiply in the same order), accumulated per lane.
        n_passed += (uint32_t)__popcll(out.m_passed);
        struct { int32_t n_masses, pad; } hdr;
        reload_kernarg(hdr, offsetof(HistKernArgs, SC));
        const uint32_t fl = (uint32_t)__builtin_amdgcn_readfirstlane((int)Ab.flags);
        const bool use_prob = !(fl & SART_CF_IGNORE_CONV_PROB) & ((GAS >= 0) ? (GAS == 1) : (__builtin_amdgcn_readfirstlane(Bo.P.stage_gas) != 0));
        const double term1 = Bo.P.gas_term1;
        const GasCos cos_lead = GasCos::make();      // (two vector register pairs for the whole loop)
        for (int k = 0; k < hdr.n_masses; ++k) {   // wave-uniform
          asm volatile("; hot: per-mass loop of the fused scan");
          ScanMass M;   // (staging the table in LDS and reading it one entry ahead instead of this scalar load: -1.5 %, not kept)
          reload_kernarg(M, offsetof(HistKernArgs, SC) + offsetof(ScanArgs, m) + (size_t)k * sizeof(ScanMass));
          double w = out.weight;
          if (use_prob) w *= gas_conversion_prob(M.dm2_abs, out.gas, term1, L.sincos, cos_lead);
          const uint64_t nz = out.m_passed & ballot64(w != 0.0);
          if (nz != out.m_passed) {
            asm volatile("; rare: a conversion probability of exactly zero");
            if (lane == 0) atomicAdd(&scan_zero[k], (uint32_t)__popcll(out.m_passed & ~nz));
          }
          if (__builtin_amdgcn_inverse_ballot_w64(nz)) {
            // cell [k][0][lane % 32]; [k][1][.] is kScanLanes further on (lanes l and l + 32 add to the same cell: the LDS unit
            // takes a wave's 64-bit atomics in groups of lanes, the two never meet in one group)
            const uint32_t t = (uint32_t)k * (2u * kScanLanes) + ((uint32_t)lane & (kScanLanes - 1u));
            if constexpr (FIXED) {
              __hip_atomic_fetch_add(reinterpret_cast<unsigned long long*>(tile_cell_lo(t)), (unsigned long long)to_fixed(w, M.fx_scale_w),
                                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              __hip_atomic_fetch_add(reinterpret_cast<unsigned long long*>(tile_cell_lo(t + kScanLanes)), (unsigned long long)to_fixed(w * w, M.fx_scale_w2),
                                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else {
              __hip_atomic_fetch_add(tile_cell_lo(t), w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              __hip_atomic_fetch_add(tile_cell_lo(t + kScanLanes), w * w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
          }
        }
      }
    }
    h1 += n_valid;
    if constexpr (SCAN) {
      n_nickel += (uint32_t)__popcll(out.m_nickel);
      __builtin_amdgcn_s_setprio(0);
      return;
    }
    __builtin_amdgcn_s_setprio(SART_PRIO_ACC);
#ifdef SART_STAGE_TIMING
    for (int k = 0; k < 5; ++k) cyc_bs[k] += out.tb[k + 1] - out.tb[k];
    cyc_bs[5] += out.tb[0];     // entry stamps, to place the sub-stages inside the B span
#endif
    n_nickel += (uint32_t)__popcll(out.m_nickel);
    n_till += (uint32_t)__popcll(out.m_till);
    n_passed += (uint32_t)__popcll(out.m_passed);
    if (out.passed) {
      SART_STAGE_MARK("ACC");
      // the launch's image parameters, re-read from the kernel arguments (scalar registers, short-lived)
      TraceArgs Al;
      reload_kernarg(Al, offsetof(HistKernArgs, A));
      // ONE scalar-memory round trip for everything the accumulation reads: left alone the compiler sinks the loads of the tile and
      // replica parameters into the nested regions that use them - three dependent waits of a scalar load each per pass
      asm volatile("" :: "s"(Al.replicas), "s"(Al.replica_mask), "s"(Al.replica_stride), "s"(Al.image_nx), "s"(Al.image_ny),
                   "s"(Al.image_x_min), "s"(Al.image_y_min), "s"(Al.image_inv_step_x), "s"(Al.image_inv_step_y), "s"(Al.spectra),
                   "s"(Al.tile_x0), "s"(Al.tile_y0), "s"(Al.tile_n), "s"(Al.tile_base));
      long long w_fx = 0;   // FIXED: this ray's weight in quanta (what the image, the sums and the spectra add)
      if constexpr (FIXED) {
        w_fx = to_fixed(out.weight, Al.fx_scale_w);
        sum_w += w_fx;
        sum_w2 += to_fixed(out.weight * out.weight, Al.fx_scale_w2);
        sum_x += to_fixed(out.px, kFixedPositionScale);
        sum_y += to_fixed(out.py, kFixedPositionScale);
        sum_r += to_fixed(out.rdet, kFixedPositionScale);
      } else {
        sum_w += out.weight;
        sum_w2 = fma(out.weight, out.weight, sum_w2);
        sum_x += out.px;
        sum_y += out.py;
        sum_r += out.rdet;
      }
      // prepareHeatmap (:838-842): img[floor(y / step_y), floor(x / step_x)] += w
      // floor(t) in [0, n) <=> 0 <= t < n, and the conversion to int truncates = floor for t >= 0
      const double fx = (out.px - Al.image_x_min) * Al.image_inv_step_x;
      const double fy = (out.py - Al.image_y_min) * Al.image_inv_step_y;
      const int nx = Al.image_nx, ny = Al.image_ny;
      // (lane masks of the four compares combined on the scalar unit; ballots in here see the passed lanes only)
      const uint64_t inside_m = ballot64(fx >= 0.0) & ballot64(fx < (double)nx) & ballot64(fy >= 0.0) & ballot64(fy < (double)ny);
      const bool inside = __builtin_amdgcn_inverse_ballot_w64(inside_m);
      // per lane: this region runs under the lane mask of the passed rays, and a wave-uniform count added in here lands in a
      // vector register whose lane 0 - the one the epilogue reads - only sees the passes in which its own ray passed
      n_outside += inside ? 0u : 1u;
      if constexpr (FIXED) {
        if (out.m_passed & ~inside_m) {   // wave-uniform: no ray of a wave is outside for images that cover the chip
          asm volatile("; rare: passed rays outside the image");
          sum_wo += inside ? 0ll : w_fx;
        }
      }
      // this wave's replica of the image; the pixel's byte offset is 32-bit (image < 2^29 pixels, checked on the host)
#ifdef SART_DEBUG_KNOBS
      const uint32_t rep_key = (Al.flags & 0x08000000u) ? blockIdx.x : (uint32_t)wave_global;   // experiment: replica per XCD
      double* const img = Al.replicas + (size_t)(rep_key & Al.replica_mask) * (size_t)Al.replica_stride;
#else
      double* const img = Al.replicas + (size_t)((uint32_t)wave_global & Al.replica_mask) * (size_t)Al.replica_stride;
#endif
#ifdef SART_DEBUG_KNOBS
      if (inside && !(Al.flags & 0x40000000u))   // SART_DEBUG_NO_IMAGE_ATOMICS (experiment builds only)
#else
      if (inside)
#endif
      {
        const uint32_t ix = (uint32_t)(int)fx, iy = (uint32_t)(int)fy;
        // the workgroup's LDS tile around the focal spot first ("LDS then global atomics"): scattered f64 atomics execute at
        // the memory side, one 64-byte request per lane, ~2e10 / s for the whole chip
        const uint32_t tn = (uint32_t)Al.tile_n;
        const uint32_t tx = ix - (uint32_t)Al.tile_x0, ty = iy - (uint32_t)Al.tile_y0;   // unsigned: below the origin wraps to huge
        if ((tx < tn) & (ty < tn)) {
          const uint32_t t = ty * tn + tx + (uint32_t)Al.tile_base;          // < kTileRingCells + kTileExtraCells (host: tile size per base)
          if constexpr (FIXED)
            __hip_atomic_fetch_add(reinterpret_cast<unsigned long long*>(tile_cell(t)), (unsigned long long)w_fx, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_WORKGROUP);                                             // ds_add_u64
          else
            __hip_atomic_fetch_add(tile_cell(t), out.weight, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // ds_add_f64
        } else {
          const uint32_t pix = iy * (uint32_t)nx + ix;
          typedef __attribute__((address_space(1))) char* gbytes;
          if constexpr (FIXED) atomic_add_slot_i64((double*)((gbytes)img + pix * 8u), w_fx);
          else unsafeAtomicAdd((double*)((gbytes)img + pix * 8u), out.weight);
        }
      }
      if (Al.spectra) {   // wave-uniform: radial and per-energy histograms behind the scalars
        double* rad = acc + (size_t)nx * (size_t)ny + SART_ACC_COUNT;
        double* en = rad + 2 * (size_t)Al.n_radial_bins;
        const size_t ne1 = (size_t)Pb.n_energies + 1;
        const int rb = min((int)(out.rdet * Al.radial_inv_bin), Al.n_radial_bins - 1);
        if constexpr (FIXED) {
          atomic_add_slot_i64(&rad[rb], 1);
          atomic_add_slot_i64(&rad[(size_t)Al.n_radial_bins + rb], w_fx);
          atomic_add_slot_i64(&en[out.e_idx], 1);
          atomic_add_slot_i64(&en[ne1 + out.e_idx], w_fx);
          atomic_add_slot_i64(&en[2 * ne1 + out.e_idx], to_fixed(out.reflect, kFixedReflectScale));
        } else {
          unsafeAtomicAdd(&rad[rb], 1.0);
          unsafeAtomicAdd(&rad[(size_t)Al.n_radial_bins + rb], out.weight);
          unsafeAtomicAdd(&en[out.e_idx], 1.0);
          unsafeAtomicAdd(&en[ne1 + out.e_idx], out.weight);
          unsafeAtomicAdd(&en[2 * ne1 + out.e_idx], out.reflect);
        }
      }
    }
    __builtin_amdgcn_s_setprio(0);
  };

  // Zone bounds of stage A0.  The specialised variants have vector registers to spare (<= 120 of 128) and keep the eight bounds
  // there, one copy per lane, for the whole kernel: the compares read them as they are.  (Re-read from the kernel arguments
  // per pass - what the generic variants do - every stage-A0 pass waits for a scalar-memory round trip with nothing to put
  // in front of it; held in scalar registers they would be spilled through VGPR lanes.)
  constexpr bool kZonesInVgprs = FAST;
  uint32_t zone_lo_v[kMaxZones], zone_hi_v[kMaxZones];
#pragma unroll
  for (int z = 0; z < kMaxZones; ++z) {
    zone_lo_v[z] = H.zone_lo[z];
    zone_hi_v[z] = H.zone_hi[z];
    if (kZonesInVgprs) asm volatile("" : "+v"(zone_lo_v[z]), "+v"(zone_hi_v[z]));   // opaque: stays in a vector register
  }
  uint32_t zone_reached_v = H.zone_reached;   // (one v_readfirstlane per pass: a scalar register held across the loop would push another value out)
  if (kZonesInVgprs) asm volatile("" : "+v"(zone_reached_v));
  uint32_t chunk = (uint32_t)wave_global;                            // relative to first_chunk (wave-uniform)
  const uint32_t lane4 = 4u * (uint32_t)lane;
  uint32_t pass = 0;                                                 // 0..3: which of its four ids a lane handles now
  U4 stream = U4{0u, 0u, 0u, 0u};                                    // this lane's block of the shared word stream
#ifdef SART_STAGE_TIMING
  const uint64_t cyc_start = __builtin_readcyclecounter();
  const uint64_t real_start = __builtin_amdgcn_s_memrealtime();   // constant 100 MHz: shader clock = d(memtime) / d(memrealtime) x 100 MHz
#define SART_STAMP(var) const uint64_t var = __builtin_readcyclecounter()
#define SART_SPAN(acc, t_begin) acc += __builtin_readcyclecounter() - (t_begin)
#else
#define SART_STAMP(var)
#define SART_SPAN(acc, t_begin)
#endif
  for (;;) {
    const bool have_new = chunk < n_chunks;   // wave-uniform
    SART_STAGE_MARK("LOOP");
    if (have_new) {
      SART_STAGE_MARK("A0");
      SART_STAMP(ts_a0);
      if (pass == 0u) {
        asm volatile("; hot x0.25: one block of the shared word stream per four passes");
        stream = stream_block(((first_chunk + (uint64_t)chunk) << 6) + (uint64_t)lane, A.seed_lo, A.seed_hi);
      }
      const uint32_t w = word_of(stream, pass);                      // high word of u3 (:418) of this pass' ray
      const uint32_t rel = ((chunk << 8) + pass) + lane4;
      if (early_reject) {
        // ---- stage A0: the word against the zones (lane masks and scalar arithmetic only) ----
        ZoneTable Z;
        if constexpr (!kZonesInVgprs) reload_zones(Z);
        const uint32_t zone_reached = kZonesInVgprs ? (uint32_t)__builtin_amdgcn_readfirstlane((int)zone_reached_v) : Z.zone_reached;
        // lane masks of direct compares (v_cmp writes them) and scalar arithmetic on the masks; the final mask becomes the
        // EXEC mask of the ring write as it is (inverse ballot), with no per-lane 0 / 1 in between
        uint64_t dead_m = 0, reached_m = 0;
#pragma unroll
        for (int z = 0; z < kMaxZones; ++z) {   // unused zones are empty: lo > hi
          const uint64_t in = kZonesInVgprs ? (ballot64(w >= zone_lo_v[z]) & ballot64(w <= zone_hi_v[z]))
                                            : (ballot64(w >= Z.lo[z]) & ballot64(w <= Z.hi[z]));
          dead_m |= in;
          reached_m |= ((zone_reached >> z) & 1u) ? in : 0ull;      // wave-uniform select
        }
        // all four ids of every lane lie inside the launch for every chunk but the first and the last (wave-uniform test)
        const uint32_t chunk_lo = chunk << 8;
        uint64_t valid_m = ~0ull;
        if ((chunk_lo < rel_begin) | (chunk_lo + 256u > rel_end)) valid_m = ballot64(rel >= rel_begin) & ballot64(rel < rel_end);
        n_reached += (uint32_t)__popcll(valid_m & reached_m);
#ifdef SART_DEBUG_KNOBS
        const uint64_t mask = (A.flags & 0x20000000u) ? 0ull : (valid_m & ~dead_m);   // experiment: nothing reaches stage A1
#else
        const uint64_t mask = valid_m & ~dead_m;
#endif
        if (__builtin_amdgcn_inverse_ballot_w64(mask)) {
          asm volatile("; hot: ring 0 write");
          const uint32_t slot = (t0 + prefix_of(mask)) % kQueue;
          Q.w[wave].ray[slot] = rel;
          Q.w[wave].u3hi[slot] = w;
        }
        t0 += (uint32_t)__popcll(mask);
      } else {
        run_phase_a(rel, (rel >= rel_begin) & (rel < rel_end), w);   // no early-rejection stage for this configuration
      }
      pass = (pass + 1u) & 3u;
      if (pass == 0u) chunk += (uint32_t)waves_total;
      ring_sync();
      SART_SPAN(cyc_a0, ts_a0);
    }
    SART_STAGE_MARK("LOOP");
    if (early_reject) {
      // ---- stage A1 on a full wave of A0 survivors (or on the remainder once the input is exhausted) ----
      const uint32_t n0 = t0 - h0;
      if ((n0 >= 64u) | (!have_new & (n0 > 0u))) {   // bitwise on purpose: `||` would become a per-lane select of 0 / 1
        const uint32_t m = min(n0, 64u);
        const bool v = (uint32_t)lane < m;
        const uint32_t slot = (h0 + (uint32_t)lane) % kQueue;
        // slots beyond m hold ids of earlier rays (the rings are zeroed at the start): lanes there trace a ray nobody counts
        const uint32_t rel = Q.w[wave].ray[slot];
        const uint32_t w = Q.w[wave].u3hi[slot];
        h0 += m;
        SART_STAMP(ts_a1);
        run_phase_a(rel, v, w);
        ring_sync();
        SART_SPAN(cyc_a1, ts_a1);
      }
    }
    // ---- stage B on a full wave of A1 survivors (or on the remainder at the very end) ----
    SART_STAGE_MARK("LOOP");
    const uint32_t n1 = t1 - h1;
    const bool draining = !have_new & (t0 == h0);
    if ((n1 >= 64u) | (draining & (n1 > 0u))) {
      SART_STAMP(ts_b);
      run_phase_b(min(n1, 64u));
      ring_sync();
      SART_SPAN(cyc_b, ts_b);
    }
    if (draining & (t1 == h1)) break;
  }

  SART_STAGE_MARK("EPILOGUE");
  if constexpr (SCAN) {
    // per-mass sums of this workgroup: a wave adds up the accumulator cells of a mass (a wave-wide sum over a cell per lane: the
    // order is fixed) -> one plain store per mass and quantity, folded by fold_scan_kernel.  One wave-wide read covers [k][0][.]
    // and [k][1][.]: lanes 0 .. 31 hold the sum-of-w cells, lanes 32 .. 63 the sum-of-w^2 cells.
    __syncthreads();   // every wave of the workgroup has left the loop
    static_assert(2 * kScanLanes == 64, "one cell per lane");
    for (int k = wave; k < SCarg.n_masses; k += BLOCK / 64) {
      using Sum = std::conditional_t<FIXED, long long, double>;
      Sum* const dst = reinterpret_cast<Sum*>(SCarg.partials) + ((size_t)blockIdx.x * kScanMaxMasses + (size_t)k) * kScanPartialSlots;
      const double cell = *tile_cell_lo((uint32_t)k * 64u + (uint32_t)lane);
      Sum v;
      if constexpr (FIXED) v = __double_as_longlong(cell); else v = cell;
#pragma unroll
      for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);   // within each half of the wave
      if (lane == 0) { dst[0] = v; dst[2] = (Sum)scan_zero[k]; dst[3] = 0; }
      if (lane == 32) dst[1] = v;
    }
  }
  // flush of the LDS image tile: one global atomic per non-empty tile pixel and workgroup
  if (!SCAN && A.tile_n > 0) {
    __syncthreads();   // every wave of the workgroup has left the loop (uniform condition: kernel argument)
    const uint32_t tn = (uint32_t)A.tile_n, n_tile = tn * tn;
    double* const img = A.replicas + (size_t)((uint32_t)wave_global & A.replica_mask) * (size_t)A.replica_stride;
    for (uint32_t t = threadIdx.x; t < n_tile; t += BLOCK) {
      const double v = *tile_cell(t + (uint32_t)A.tile_base);
      if (__double_as_longlong(v) != 0ll) {   // FIXED: the cell holds an integer; f64: +0.0 is all zero bits, too
        const uint32_t ty = t / tn, tx = t - ty * tn;
        double* const px = &img[(size_t)((uint32_t)A.tile_y0 + ty) * (size_t)A.image_nx + ((uint32_t)A.tile_x0 + tx)];
        if constexpr (FIXED) atomic_add_slot_i64(px, __double_as_longlong(v));
        else unsafeAtomicAdd(px, v);
      }
    }
  }

  // scalars: wave reduction -> LDS -> one plain store per workgroup and quantity (folded by fold_scalars_kernel)
  // (FIXED: the slots hold int64 - sums in quanta, counters as integers - and are added as integers all the way)
  // (the staging area is the head of wave 0's rings - X0, Y0, tsx: dead once every wave has left the loop, and apart from the
  // image tile / the scan accumulators, which live in the path or the ring-0 columns)
  static_assert(sizeof(Sum) * (BLOCK / 64) * SART_ACC_COUNT <= 3 * kQueue * sizeof(double) && offsetof(WaveRings, Y0) == kQueue * sizeof(double) &&
                    offsetof(WaveRings, tsx) == 2 * kQueue * sizeof(double), "the scalar staging area fits the first three ring columns of wave 0");
  __syncthreads();
  Sum (*const red)[SART_ACC_COUNT] = reinterpret_cast<Sum (*)[SART_ACC_COUNT]>(&Q.w[0].X0[0]);
  Sum sw, sw2, sxx, syy, srr;
  if constexpr (FIXED) {
    sw = wave_sum_i64(sum_w); sw2 = wave_sum_i64(sum_w2); sxx = wave_sum_i64(sum_x); syy = wave_sum_i64(sum_y); srr = wave_sum_i64(sum_r);
  } else {
    sw = wave_sum(sum_w); sw2 = wave_sum(sum_w2); sxx = wave_sum(sum_x); syy = wave_sum(sum_y); srr = wave_sum(sum_r);
  }
  const long long swo = FIXED ? wave_sum_i64(sum_wo) : 0ll;
  const long long n_out = wave_sum_i64((long long)n_outside);   // (a per-lane count, see the accumulation)
  if (lane == 0) {
    Sum* r = red[wave];
    for (int k = 0; k < SART_ACC_COUNT; ++k) r[k] = 0;
    r[SART_ACC_SUM_WEIGHTS] = sw;
    r[SART_ACC_SUM_WEIGHTS_SQ] = sw2;
    r[SART_ACC_SUM_X] = sxx;
    r[SART_ACC_SUM_Y] = syy;
    r[SART_ACC_SUM_R] = srr;
    r[SART_ACC_N_PASSED] = (Sum)n_passed;
    r[SART_ACC_N_PASSED_TILL_WINDOW] = (Sum)n_till;
    r[SART_ACC_N_HIT_NICKEL] = (Sum)n_nickel;
    r[SART_ACC_N_REACHED_TELESCOPE] = (Sum)n_reached;
    r[SART_ACC_N_SHELL_SELECTED] = (Sum)n_shell;
    r[SART_ACC_N_OUTSIDE_IMAGE] = (Sum)n_out;
    if constexpr (FIXED) r[SART_ACC_SUM_WEIGHTS_OUTSIDE] = swo;
#ifdef SART_STAGE_TIMING
    r[12] = (Sum)cyc_a0; r[13] = (Sum)cyc_a1; r[14] = (Sum)cyc_b;
    r[15] = (Sum)(__builtin_readcyclecounter() - cyc_start);
    r[SART_ACC_N_HIT_NICKEL] = (Sum)(__builtin_amdgcn_s_memrealtime() - real_start);
    // sub-stages of B, packed two per slot (each < 2^40, slot = hi * 2^40 + lo would lose bits in f64): use the sums of x/y/r slots
    r[SART_ACC_SUM_X] = (Sum)cyc_bs[0]; r[SART_ACC_SUM_Y] = (Sum)cyc_bs[1]; r[SART_ACC_SUM_R] = (Sum)cyc_bs[2];
    r[SART_ACC_SUM_WEIGHTS_SQ] = (Sum)cyc_bs[3]; r[SART_ACC_N_OUTSIDE_IMAGE] = (Sum)cyc_bs[4];
#endif
  }
  __syncthreads();
  if (threadIdx.x < SART_ACC_COUNT) {
    Sum t = 0;
    for (int w = 0; w < BLOCK / 64; ++w) t += red[w][threadIdx.x];
    reinterpret_cast<Sum*>(A.partials)[(size_t)blockIdx.x * SART_ACC_COUNT + threadIdx.x] = t;
  }
}

// ------------------------------------------------------------------------------------------------
// fused angular scan (include/sart.h: sart_trace_angular_scan; BASELINE configs[3])
// ------------------------------------------------------------------------------------------------
// performAngularScan (:2778-2802) re-runs the whole trace per telescope angle.  The angle enters a ray at the transformation
// into the telescope's frame (:1878-1899) and nowhere before it: sampling, the three sincos, the radius draw, bore, cold-bore
// exit and pipe cuts (phase_a_bore: ~215 of phase A's ~365 vector instructions in the rotated variant, and all of stage A0) do
// not depend on it.  This kernel therefore takes every ray through stage A0 and the first half of phase A ONCE and keeps the
// full wave of rays that come out of it in registers (the "stash": 11 values per lane - x1, y1, x3, y3, path, energy index: the
// energy draw (:444-471: radius index and u5 -> lowerBound in the CDF row) does not depend on the angle either, so its two
// table gathers with the worst locality - guide word, CDF candidates - are paid once per ray, not once per ray and angle);
// a wave-uniform walk over the launch's angles runs the second half of phase A (phase_a_telescope with that angle's rotation,
// read from the kernel arguments with scalar loads) on the stash, compacts the survivors into ring 1 with their angle index, and
// phase B runs on full waves of ring 1 as in the histogram kernel - a pass usually holds rays of two or three angles, so what
// phase B needs of the angle (the third column of the rotation, for z of pointExitCB) comes from a small LDS table per lane and
// the sums go to per-angle LDS cells (ds_add_f64 / ds_add_u64; ring 1's u5 column, free here) and counters (ds_add_u32).  No image,
// no position sums.
// One loop iteration runs at most one pass of each stage (one call site each: the instruction footprint of the histogram kernel):
//   no stash (ka == n_angles):  A0 pass -> ring 0;  ring 0 holds a full wave (or the input is exhausted) -> A1a -> stash, ka = 0
//   stash (ka < n_angles):      A1b for angle ka -> ring 1, ++ka
//   ring 1 holds a full wave (or everything before it has drained) -> B
// Same rays for every angle (common random numbers), the ray ids and uniforms of the histogram kernels: per angle the sums
// equal a sart_trace_histogram launch with that angle set on the same ray ids - bit for bit in SART_ACCUM_FIXED64.
// FAST / GAS as in trace_histogram_kernel (the telescope is always "rotated" here: angle 0 runs through the same arithmetic
// with cos = 1, sin = 0).
struct AScanKernArgs {
  HotA H;
  const DevBlob* blob;
  TraceArgs A;
  double* unused;      // (keeps the argument offsets of trace_histogram_kernel: the reload_* helpers are shared)
  HotB HB;
  AScanArgs AN;
};
static_assert(offsetof(AScanKernArgs, A) == offsetof(HistKernArgs, A) && offsetof(AScanKernArgs, HB) == offsetof(HistKernArgs, HB) &&
                  offsetof(AScanKernArgs, AN) == offsetof(HistKernArgs, SC) && sizeof(AScanAngle) % 8 == 0,
              "the angular-scan kernel re-reads its arguments at the offsets of the histogram kernel's");
constexpr int kAScanCells = kAScanMaxAngles * 2 * kScanLanes;          // [angle][sum w, sum w^2][kScanLanes]: ring 1's u5 column, 128 doubles per wave
constexpr int kAScanMTable = 0;                                        // [angle][mx, my, mz] in the cells behind the tables (the histogram kernels' tile_extra)
constexpr int kAScanCounters = 3 * kAScanMaxAngles;                    // behind it: (4 kAScanMaxAngles + 4) u32
static_assert(kAScanMTable + 3 * kAScanMaxAngles + (4 * kAScanMaxAngles + 4 + 1) / 2 <= kTileExtraCells, "rotation columns + counters live where the histogram kernels keep the image tile behind the tables");
static_assert(2 * kScanLanes == 64, "one cell per lane in the epilogue");

template <int BLOCK, bool FAST, int GAS, bool FIXED>
__global__ __launch_bounds__(BLOCK) void trace_angular_scan_kernel(HotA H, const DevBlob* __restrict__ blob, TraceArgs A,
                                                                   double* __restrict__ unused, HotB HBarg, AScanArgs ANarg) {
  struct LdsLayout {   // the layout of trace_histogram_kernel (tables first: ds_ offsets)
    TablesLds S;
    double cells[kTileExtraCells];
    DevBlob B;
    TraceArgs Ab;
    QueueLds<BLOCK / 64> Q;
  };
  __shared__ LdsLayout lds;
  static_assert((offsetof(LdsLayout, Q) % 512) == 0, "the rings are addressed with ds_*2st64 offsets");
  static_assert(kAScanCells == (BLOCK / 64) * kQueue, "the per-angle cells are the u5 column of the workgroup's rings");
  TablesLds& S = lds.S;
  QueueLds<BLOCK / 64>& Q = lds.Q;
  // per angle: N_PASSED, N_HIT_NICKEL, N_PASSED_TILL_WINDOW, N_SHELL_SELECTED of this workgroup; [4 kAScanMaxAngles]: N_REACHED_TELESCOPE
  uint32_t* const cnt = reinterpret_cast<uint32_t*>(&lds.cells[kAScanCounters]);
  // cell t of the per-angle sums: slot t % 128 of wave t / 128's u5 column
  auto acc_cell = [&](uint32_t t) -> double* { return &Q.w[t >> 7].u5[t & 127u]; };
  DevBlob& B = lds.B;
  TraceArgs& Ab = lds.Ab;
  (void)unused; (void)HBarg;
  {
    const uint64_t* src = reinterpret_cast<const uint64_t*>(blob);
    uint64_t* dst = reinterpret_cast<uint64_t*>(&B);
    for (int i = threadIdx.x; i < (int)(sizeof(DevBlob) / 8); i += BLOCK) dst[i] = src[i];
    if (threadIdx.x == 0) Ab = A;
    uint64_t* q = reinterpret_cast<uint64_t*>(&Q);
    for (int i = threadIdx.x; i < (int)(sizeof(Q) / 8); i += BLOCK) q[i] = 0ull;   // (the rings, and with them the cells)
    if (threadIdx.x < 4 * kAScanMaxAngles + 4) cnt[threadIdx.x] = 0u;
    if (threadIdx.x < 3 * kAScanMaxAngles) {
      // third column of every angle's rotation, from the kernel arguments (a per-thread read of the argument segment)
      typedef const __attribute__((address_space(4))) double* kernarg_f64;
      const kernarg_f64 a0 = (kernarg_f64)((const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr() +
                                           offsetof(AScanKernArgs, AN) + offsetof(AScanArgs, a) + offsetof(AScanAngle, mx));
      const int k = threadIdx.x / 3, j = threadIdx.x - 3 * k;
      lds.cells[kAScanMTable + threadIdx.x] = a0[k * (int)(sizeof(AScanAngle) / 8) + j];
    }
    __syncthreads();
  }
  const DevParams& Pb = B.P;
  const DevTables& Tb = B.T;
  stage_tables<BLOCK>(S, Pb, Tb);
  const LdsTables L{S.sincos, S.rcdf_hi, Tb.flux_radius_cdf, S.rguide, S.shells, S.lut};

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const uint64_t waves_total = (uint64_t)gridDim.x * (BLOCK / 64);
  const uint64_t wave_global = (uint64_t)blockIdx.x * (BLOCK / 64) + wave;
  const bool early_reject = H.n_zones > 0;   // wave-uniform; no zones for the X-ray test source
  const uint64_t first_chunk = A.ray_id_offset >> 8;
  uint64_t id_base = first_chunk << 8;
  asm volatile("" : "+s"(id_base));
  const uint32_t rel_begin = (uint32_t)(A.ray_id_offset & 255u);
  const uint32_t rel_end = rel_begin + (uint32_t)A.n_rays;
  const uint32_t n_chunks = (rel_end + 255u) >> 8;
  const int n_angles = ANarg.n_angles;

  uint32_t n_reached = 0;
  uint32_t h0 = 0, t0 = 0, h1 = 0, t1 = 0;
  auto ring_sync = [] {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  auto prefix_of = [](uint64_t mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
  };

  // the stash: the wave of rays whose angles are being worked through (registers)
  double st_x1 = 0.0, st_y1 = 0.0, st_x3 = 0.0, st_y3 = 0.0, st_path = 0.0;
  int st_eidx = 0;        // energy index of the ray (row n_energies: the X-ray test source's fixed energy)
  uint64_t stash_m = 0;   // its lanes that hold a ray of this launch that reached the telescope
  int ka = n_angles;      // next angle of the stash; n_angles: no stash

  // stage A1a: sample -> bore -> pipes for the ray with id id_base + rel
  auto run_bore = [&](uint32_t rel, bool valid, uint32_t u3_hi) {
    SART_STAGE_MARK("A1a");
    __builtin_amdgcn_s_setprio(SART_PRIO_A1);
    RayState st;
    bool sampled = false, reached = false;
    HotA Hl;
    reload_hot(Hl);
    BoreRay br;
    LaneMasks M;
    phase_a_bore<FAST, false>(Hl, Pb, L, uniforms_of(A.seed_lo, A.seed_hi, id_base + (uint64_t)rel, u3_hi), st, sampled, reached, br, M);
    stash_m = ballot64(valid) & M.reached;
    n_reached += (uint32_t)__popcll(stash_m);
    st_x1 = br.x1; st_y1 = br.y1; st_x3 = br.x3; st_y3 = br.y3;
    st_path = st.path_cb;
    if (!FAST && Hl.test_active) {   // wave-uniform
      st_eidx = Pb.n_energies;
    } else {
      // getRandomEnergyFromSolarModel (:444-471) once per ray: the three dependent gathers of the draw, here and not in phase B
      // (lanes whose ray is dead draw from a valid row with a valid uniform: the result is not used)
      HotB HB;
      reload_kernarg(HB, offsetof(AScanKernArgs, HB));
      st_eidx = sample_energy_index(HB, st.r_idx, st.u5);
    }
    ka = stash_m ? 0 : n_angles;   // (a wave none of whose rays reached the telescope has no angles to walk)
    __builtin_amdgcn_s_setprio(0);
  };

  // stage A1b: the stash through telescope frame, opaque structures and shell selection for angle k (wave-uniform)
  auto run_telescope = [&](int k) {
    SART_STAGE_MARK("A1b");
    __builtin_amdgcn_s_setprio(SART_PRIO_A1);
    HotA Hl;
    reload_hot(Hl);
    struct { TelRot R; double shell0_miss_radius; } ang;   // the head of AScanAngle
    static_assert(offsetof(AScanAngle, rx_c) == 0 && offsetof(AScanAngle, shell0_miss_radius) == sizeof(TelRot), "the head of AScanAngle");
    reload_kernarg(ang, offsetof(AScanKernArgs, AN) + offsetof(AScanArgs, a) + (size_t)k * sizeof(AScanAngle));
    BoreRay br;
    br.x1 = st_x1; br.y1 = st_y1; br.x3 = st_x3; br.y3 = st_y3;
    br.sx = 0.0; br.sy = 0.0;   // (read by the unrotated form only)
    br.ok = __builtin_amdgcn_inverse_ballot_w64(stash_m);
    br.okm = stash_m;
    RayState st;
    st.path_cb = st_path;
    double radial;
    LaneMasks M;
    (void)phase_a_telescope<FAST, 1>(Hl, Pb, ang.R, L, br, st, radial, M);
    const uint64_t selected = M.ok;   // (starts from stash_m: valid, reached)
    if (lane == 0) atomicAdd(&cnt[4 * k + 3], (uint32_t)__popcll(selected));
    // rays that provably miss the first mirror of the innermost shell end here, as in the histogram kernel (this angle's bound)
    const uint64_t mask = selected & ~ballot64(radial < ang.shell0_miss_radius);
    if (__builtin_amdgcn_inverse_ballot_w64(mask)) {
      asm volatile("; hot: ring 1 write");
      const uint32_t slot = (t1 + prefix_of(mask)) % kQueue;
      Q.w[wave].X0[slot] = st.X0; Q.w[wave].Y0[slot] = st.Y0;
      Q.w[wave].tsx[slot] = st.tsx; Q.w[wave].tsy[slot] = st.tsy;
      Q.w[wave].path[slot] = st.path_cb;
      Q.w[wave].idx[slot] = st_eidx | (st.shell << 16) | (k << 24);   // energy index < 2^16 (the guide tables are u16), shell < 64, angle < 32
    }
    t1 += (uint32_t)__popcll(mask);
    __builtin_amdgcn_s_setprio(0);
  };

  // stage B: mirrors -> weight on a wave of ring 1 (rays of several angles), accumulated per angle
  auto run_mirrors = [&](uint32_t n_valid) {
    SART_STAGE_MARK("B");
    __builtin_amdgcn_s_setprio(SART_PRIO_B);
    RayState st;
    const bool valid = (uint32_t)lane < n_valid;
    const uint32_t slot = (h1 + (uint32_t)lane) % kQueue;
    st.X0 = Q.w[wave].X0[slot]; st.Y0 = Q.w[wave].Y0[slot];
    st.tsx = Q.w[wave].tsx[slot]; st.tsy = Q.w[wave].tsy[slot];
    st.path_cb = Q.w[wave].path[slot];
    st.u5 = 0.0;
    st.r_idx = 0;
    const int packed = Q.w[wave].idx[slot];   // (slots beyond n_valid hold earlier rays or zeros: a real ray's indices, angle < n_angles)
    const int e_idx = packed & 0xFFFF;
    st.shell = (packed >> 16) & 0xFF;
    const uint32_t kl = (uint32_t)packed >> 24;
    {
      const double* const m = &lds.cells[kAScanMTable + 3 * kl];
      st.zcb = zcb_rotated(m[0], m[1], m[2], Pb.half_length_telescope, st.X0 + Pb.entrance_x, st.Y0 + Pb.entrance_y, st.tsx, st.tsy,
                           H.dz3 - H.dz1);
    }
    h1 += n_valid;
    const DevBlob& Bo = lds_opaque(B);
    HotB HB;
    reload_kernarg(HB, offsetof(AScanKernArgs, HB));
    asm volatile("" :: "s"(HB.cdf_hi32), "s"(HB.energy_guide), "s"(HB.energy_tab), "s"(HB.refl), "s"(HB.refl_n_angles), "s"(HB.cdf_stride));
    RayOut out;
    phase_b<false, FAST, GAS, false, false, true>(Bo.P, L, HB, lds_opaque(Ab), st, e_idx, valid, out, nullptr);   // (no draw in there)
    SART_STAGE_MARK("ACC");
    __builtin_amdgcn_s_setprio(SART_PRIO_ACC);
    if (__builtin_amdgcn_inverse_ballot_w64(out.m_nickel)) atomicAdd(&cnt[4 * kl + 1], 1u);
    if (__builtin_amdgcn_inverse_ballot_w64(out.m_till)) atomicAdd(&cnt[4 * kl + 2], 1u);
    if (out.passed) {
      atomicAdd(&cnt[4 * kl], 1u);
      // cell [kl][0][lane % 32]; [kl][1][.] is kScanLanes further on (lanes l and l + 32 share a cell, as in the mass scan)
      double* const cell = acc_cell(kl * (2u * kScanLanes) + ((uint32_t)lane & (kScanLanes - 1u)));   // (+ kScanLanes stays inside the wave's column)
      if constexpr (FIXED) {
        double fx_w, fx_w2;
        { const TraceArgs& Al = lds_opaque(Ab); fx_w = Al.fx_scale_w; fx_w2 = Al.fx_scale_w2; }
        __hip_atomic_fetch_add(reinterpret_cast<unsigned long long*>(cell), (unsigned long long)to_fixed(out.weight, fx_w), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(reinterpret_cast<unsigned long long*>(cell + kScanLanes), (unsigned long long)to_fixed(out.weight * out.weight, fx_w2),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      } else {
        __hip_atomic_fetch_add(cell, out.weight, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(cell + kScanLanes, out.weight * out.weight, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
    __builtin_amdgcn_s_setprio(0);
  };

  // zone bounds of stage A0 (see trace_histogram_kernel)
  constexpr bool kZonesInVgprs = FAST;
  uint32_t zone_lo_v[kMaxZones], zone_hi_v[kMaxZones];
#pragma unroll
  for (int z = 0; z < kMaxZones; ++z) {
    zone_lo_v[z] = H.zone_lo[z];
    zone_hi_v[z] = H.zone_hi[z];
    if (kZonesInVgprs) asm volatile("" : "+v"(zone_lo_v[z]), "+v"(zone_hi_v[z]));
  }
  uint32_t zone_reached_v = H.zone_reached;
  if (kZonesInVgprs) asm volatile("" : "+v"(zone_reached_v));
  uint32_t chunk = (uint32_t)wave_global;
  const uint32_t lane4 = 4u * (uint32_t)lane;
  uint32_t pass = 0;
  U4 stream = U4{0u, 0u, 0u, 0u};
  for (;;) {
    const bool have_new = chunk < n_chunks;   // wave-uniform
    SART_STAGE_MARK("LOOP");
    if (ka == n_angles) {   // no stash: take rays in until a full wave of them has passed stage A0
      if (have_new) {
        SART_STAGE_MARK("A0");
        if (pass == 0u) stream = stream_block(((first_chunk + (uint64_t)chunk) << 6) + (uint64_t)lane, A.seed_lo, A.seed_hi);
        const uint32_t w = word_of(stream, pass);
        const uint32_t rel = ((chunk << 8) + pass) + lane4;
        pass = (pass + 1u) & 3u;
        if (pass == 0u) chunk += (uint32_t)waves_total;
        if (early_reject) {
          ZoneTable Z;
          if constexpr (!kZonesInVgprs) reload_zones(Z);
          const uint32_t zone_reached = kZonesInVgprs ? (uint32_t)__builtin_amdgcn_readfirstlane((int)zone_reached_v) : Z.zone_reached;
          uint64_t dead_m = 0, reached_m = 0;
#pragma unroll
          for (int z = 0; z < kMaxZones; ++z) {
            const uint64_t in = kZonesInVgprs ? (ballot64(w >= zone_lo_v[z]) & ballot64(w <= zone_hi_v[z]))
                                              : (ballot64(w >= Z.lo[z]) & ballot64(w <= Z.hi[z]));
            dead_m |= in;
            reached_m |= ((zone_reached >> z) & 1u) ? in : 0ull;
          }
          const uint64_t valid_m = ballot64(rel >= rel_begin) & ballot64(rel < rel_end);
          n_reached += (uint32_t)__popcll(valid_m & reached_m);
          const uint64_t mask = valid_m & ~dead_m;
          if (__builtin_amdgcn_inverse_ballot_w64(mask)) {
            asm volatile("; hot: ring 0 write");
            const uint32_t slot = (t0 + prefix_of(mask)) % kQueue;
            Q.w[wave].ray[slot] = rel;
            Q.w[wave].u3hi[slot] = w;
          }
          t0 += (uint32_t)__popcll(mask);
          ring_sync();
        } else {
          run_bore(rel, (rel >= rel_begin) & (rel < rel_end), w);   // no early-rejection stage for this configuration
        }
      }
      SART_STAGE_MARK("LOOP");
      if (early_reject) {
        const uint32_t n0 = t0 - h0;
        if ((n0 >= 64u) | (!have_new & (n0 > 0u))) {
          const uint32_t m = min(n0, 64u);
          const uint32_t slot = (h0 + (uint32_t)lane) % kQueue;
          const uint32_t rel = Q.w[wave].ray[slot];
          const uint32_t w = Q.w[wave].u3hi[slot];
          h0 += m;
          run_bore(rel, (uint32_t)lane < m, w);
        }
      }
    }
    SART_STAGE_MARK("LOOP");
    if (ka < n_angles) {
      run_telescope(ka);
      ++ka;
      ring_sync();
    }
    SART_STAGE_MARK("LOOP");
    const uint32_t n1 = t1 - h1;
    const bool draining = !have_new & (t0 == h0) & (ka == n_angles);
    if ((n1 >= 64u) | (draining & (n1 > 0u))) {
      run_mirrors(min(n1, 64u));
      ring_sync();
    }
    if (draining & (t1 == h1)) break;
  }

  SART_STAGE_MARK("EPILOGUE");
  // per angle: a wave adds up the angle's cells (lanes 0 .. 31: sum of w, lanes 32 .. 63: sum of w^2; fixed order) -> plain stores
  // into this workgroup's partial row, folded by fold_ascan_kernel
  if (lane == 0) atomicAdd(&cnt[4 * kAScanMaxAngles], n_reached);
  __syncthreads();   // every wave has left the loop
  using Sum = std::conditional_t<FIXED, long long, double>;
  for (int k = wave; k < n_angles; k += BLOCK / 64) {
    Sum* const dst = reinterpret_cast<Sum*>(ANarg.partials) + ((size_t)blockIdx.x * kAScanMaxAngles + (size_t)k) * kAScanPartialSlots;
    const double cell = *acc_cell((uint32_t)(k * 64 + lane));
    Sum v;
    if constexpr (FIXED) v = __double_as_longlong(cell); else v = cell;
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);   // within each half of the wave
    if (lane == 0) dst[0] = v;
    if (lane == 32) dst[1] = v;
    if (lane < 4) dst[2 + lane] = (Sum)cnt[4 * k + lane];
    if (lane == 4) dst[6] = (k == 0) ? (Sum)cnt[4 * kAScanMaxAngles] : (Sum)0;
    if (lane == 5) dst[7] = (Sum)0;
  }
}

// Fused angular scan: rows of the scan accumulator (include/sart.h: SART_ASCAN_*) += the per-workgroup sums of one launch.
// 1024 threads = 128 (angle, partial slot) pairs x 8 groups of workgroups, the summation tree of fold_scan_kernel.  `shared_row`
// (first group of angles of a scan only, else nullptr): the angle-independent counters.
template <bool FIXED>
__global__ __launch_bounds__(1024) void fold_ascan_kernel(double* __restrict__ rows_, double* __restrict__ shared_row_, const double* __restrict__ partials_,
                                                         int n_blocks, int n_angles, double n_rays) {
  using Sum = std::conditional_t<FIXED, long long, double>;
  constexpr long long kMask = (1ll << kFixedLimbBits) - 1;
  constexpr int kPairs = kAScanMaxAngles * kAScanPartialSlots, kGroups = 1024 / kPairs;
  static_assert(kPairs * kGroups == 1024 && kGroups >= 1, "one thread per (angle, partial slot, group of workgroups)");
  Sum* const rows = reinterpret_cast<Sum*>(rows_);
  Sum* const shared_row = reinterpret_cast<Sum*>(shared_row_);
  const Sum* const part = reinterpret_cast<const Sum*>(partials_);
  __shared__ Sum red_lo[kGroups][kPairs], red_hi[kGroups][kPairs];
  const int pair = threadIdx.x % kPairs, g = threadIdx.x / kPairs;
  const int k = pair / kAScanPartialSlots, j = pair % kAScanPartialSlots;
  Sum lo = 0, hi = 0;
  if (k < n_angles && j < 7) {
    for (int b = g; b < n_blocks; b += kGroups) {
      const Sum p = part[((size_t)b * kAScanMaxAngles + k) * kAScanPartialSlots + j];
      if constexpr (FIXED) { lo += p & kMask; hi += p >> kFixedLimbBits; } else lo += p;
    }
  }
  red_lo[g][pair] = lo;
  red_hi[g][pair] = hi;
  __syncthreads();
  if (g != 0) return;
  lo = 0; hi = 0;
  for (int i = 0; i < kGroups; ++i) { lo += red_lo[i][pair]; hi += red_hi[i][pair]; }
  if (k < n_angles && j < 6) {
    Sum* const row = rows + (size_t)k * SART_ASCAN_ROW;
    if (j < 2) {
      const int s = j == 0 ? SART_ASCAN_SUM_WEIGHTS : SART_ASCAN_SUM_WEIGHTS_SQ, sh = j == 0 ? SART_ASCAN_SUM_WEIGHTS_HI : SART_ASCAN_SUM_WEIGHTS_SQ_HI;
      if constexpr (FIXED) {
        lo += row[s];
        row[s] = lo & kMask;
        row[sh] += hi + (lo >> kFixedLimbBits);
      } else {
        row[s] += lo;
      }
    } else {
      const int dst = j == 2 ? SART_ASCAN_N_PASSED : j == 3 ? SART_ASCAN_N_HIT_NICKEL : j == 4 ? SART_ASCAN_N_PASSED_TILL_WINDOW : SART_ASCAN_N_SHELL_SELECTED;
      if constexpr (FIXED) row[dst] += (hi << kFixedLimbBits) + lo; else row[dst] += lo;
    }
  }
  if (shared_row && k == 0 && j == 6) {
    if constexpr (FIXED) shared_row[SART_ASCAN_N_REACHED_TELESCOPE] += (hi << kFixedLimbBits) + lo; else shared_row[SART_ASCAN_N_REACHED_TELESCOPE] += lo;
  }
  if (shared_row && pair == 7) shared_row[SART_ASCAN_N_RAYS] += (Sum)n_rays;
}

// acc scalars += sum over workgroups of the partials; N_RAYS += n_rays.  256 threads: 8 groups x 32 quantity slots (24 used).
__global__ __launch_bounds__(256) void fold_scalars_kernel(double* __restrict__ scalars, const double* __restrict__ partials,
                                                           int n_blocks, double n_rays) {
  static_assert(SART_ACC_COUNT <= 32, "layout of the reduction below");
  __shared__ double red[8][32];
  const int k = threadIdx.x & 31, g = threadIdx.x >> 5;
  double t = 0.0;
  if (k < SART_ACC_COUNT)
    for (int b = g; b < n_blocks; b += 8) t += partials[(size_t)b * SART_ACC_COUNT + k];
  red[g][k] = t;
  __syncthreads();
  if (threadIdx.x < SART_ACC_COUNT) {
    double s = (k == SART_ACC_N_RAYS) ? n_rays : 0.0;
    for (int i = 0; i < 8; ++i) s += red[i][k];
    scalars[k] += s;
  }
}

// SART_ACCUM_FIXED64 form of the same fold, in integers.  The five sums every passed ray of every launch adds to are kept in
// two limbs, value = hi * 2^40 + lo (include/sart.h): the workgroup partials (each < 2^62) are split before they are added,
// so nothing can wrap, and lo is left in [0, 2^40).  Slot k's thread owns slot k and, for a two-limb sum, its *_HI slot; the
// threads of the *_HI slots (whose partials are zero) write nothing.
__device__ __forceinline__ int fixed_hi_slot(int k) {
  return k == SART_ACC_SUM_WEIGHTS ? SART_ACC_SUM_WEIGHTS_HI : k == SART_ACC_SUM_X ? SART_ACC_SUM_X_HI
         : k == SART_ACC_SUM_Y ? SART_ACC_SUM_Y_HI : k == SART_ACC_SUM_R ? SART_ACC_SUM_R_HI
         : k == SART_ACC_SUM_WEIGHTS_SQ ? SART_ACC_SUM_WEIGHTS_SQ_HI : k == SART_ACC_SUM_WEIGHTS_OUTSIDE ? SART_ACC_SUM_WEIGHTS_OUTSIDE_HI : -1;
}
__device__ __forceinline__ bool fixed_is_hi_slot(int k) {
  return (k >= SART_ACC_SUM_WEIGHTS_HI && k <= SART_ACC_SUM_R_HI) || k == SART_ACC_SUM_WEIGHTS_SQ_HI || k == SART_ACC_SUM_WEIGHTS_OUTSIDE_HI;
}
__global__ __launch_bounds__(256) void fold_scalars_fixed_kernel(long long* __restrict__ scalars, const long long* __restrict__ partials,
                                                                 int n_blocks, long long n_rays) {
  static_assert(SART_ACC_COUNT <= 32, "layout of the reduction below");
  constexpr long long kMask = (1ll << kFixedLimbBits) - 1;
  __shared__ long long red_lo[8][32], red_hi[8][32];
  const int k = threadIdx.x & 31, g = threadIdx.x >> 5;
  long long lo = 0, hi = 0;
  if (k < SART_ACC_COUNT)
    for (int b = g; b < n_blocks; b += 8) {
      const long long p = partials[(size_t)b * SART_ACC_COUNT + k];
      lo += p & kMask;                 // arithmetic shift + mask: p = (p >> 40) * 2^40 + (p & mask) for negative p as well
      hi += p >> kFixedLimbBits;
    }
  red_lo[g][k] = lo;
  red_hi[g][k] = hi;
  __syncthreads();
  if (threadIdx.x < SART_ACC_COUNT && !fixed_is_hi_slot(k)) {
    lo = 0; hi = 0;
    for (int i = 0; i < 8; ++i) { lo += red_lo[i][k]; hi += red_hi[i][k]; }
    const int kh = fixed_hi_slot(k);
    if (kh >= 0) {
      lo += scalars[k];              // < 2^40 + 2^14 * 2^40
      scalars[k] = lo & kMask;
      scalars[kh] += hi + (lo >> kFixedLimbBits);
    } else {
      scalars[k] += (hi << kFixedLimbBits) + lo + ((k == SART_ACC_N_RAYS) ? n_rays : 0ll);
    }
  }
}

// acc[i] += sum over replicas; replicas are left zeroed for the next launch.  T = double, or long long (SART_ACCUM_FIXED64).
template <typename T>
__global__ __launch_bounds__(256) void fold_replicas_kernel(T* __restrict__ acc, T* __restrict__ replicas, int n_img,
                                                            int n_replicas, uint32_t stride) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n_img) return;
  T s = 0;
  for (int r = 0; r < n_replicas; ++r) {
    s += replicas[(size_t)r * (size_t)stride + i];
    replicas[(size_t)r * (size_t)stride + i] = 0;
  }
  acc[i] += s;
}

// Fused mass scan: rows of the scan accumulator (include/sart.h: SART_SCAN_*) += the per-workgroup sums of one launch.
// 1024 threads = 128 (mass, quantity) pairs x 8 groups of workgroups; group g sums the partial rows g, g + 8, ... of its pair in
// that order and the pair's first thread adds the eight group sums in order: the summation tree is fixed for a given grid.
// Pair (k, 2) turns the counts into N_PASSED of mass k = rays with a non-zero mass-independent weight factor minus those whose
// conversion probability for this mass is exactly zero.  `shared_row` (first group of masses of a scan only, else nullptr): the
// mass-independent counters.  FIXED: integers, the two sums in two limbs like fold_scalars_fixed_kernel.
template <bool FIXED>
__global__ __launch_bounds__(1024) void fold_scan_kernel(double* __restrict__ rows_, double* __restrict__ shared_row_, const double* __restrict__ scal_partials_,
                                                        const double* __restrict__ scan_partials_, int n_blocks, int n_masses, double n_rays) {
  using Sum = std::conditional_t<FIXED, long long, double>;
  constexpr long long kMask = (1ll << kFixedLimbBits) - 1;
  constexpr int kPairs = kScanMaxMasses * kScanPartialSlots, kGroups = 8;
  static_assert(kPairs * kGroups == 1024, "one thread per (mass, partial slot, group of workgroups)");
  Sum* const rows = reinterpret_cast<Sum*>(rows_);
  Sum* const shared_row = reinterpret_cast<Sum*>(shared_row_);
  const Sum* const scal = reinterpret_cast<const Sum*>(scal_partials_);
  const Sum* const scan = reinterpret_cast<const Sum*>(scan_partials_);
  __shared__ Sum red_lo[kGroups][kPairs], red_hi[kGroups][kPairs], red_cnt[kGroups][4];
  const int pair = threadIdx.x % kPairs, g = threadIdx.x / kPairs;
  const int k = pair >> 2, j = pair & 3;
  Sum lo = 0, hi = 0;
  if (k < n_masses && j < 3) {
    for (int b = g; b < n_blocks; b += kGroups) {
      Sum p = scan[((size_t)b * kScanMaxMasses + k) * kScanPartialSlots + j];
      if (j == 2) p = scal[(size_t)b * SART_ACC_COUNT + SART_ACC_N_PASSED] - p;
      if constexpr (FIXED) { lo += p & kMask; hi += p >> kFixedLimbBits; } else lo += p;
    }
  }
  red_lo[g][pair] = lo;
  red_hi[g][pair] = hi;
  if (pair < 4) {   // the four counters that come from the workgroup partials
    const int src = pair == 0 ? SART_ACC_N_REACHED_TELESCOPE : pair == 1 ? SART_ACC_N_SHELL_SELECTED : pair == 2 ? SART_ACC_N_HIT_NICKEL : SART_ACC_N_PASSED;
    Sum t = 0;
    if (shared_row)
      for (int b = g; b < n_blocks; b += kGroups) t += scal[(size_t)b * SART_ACC_COUNT + src];
    red_cnt[g][pair] = t;
  }
  __syncthreads();
  if (g != 0) return;
  if (k < n_masses && j < 3) {
    lo = 0; hi = 0;
    for (int i = 0; i < kGroups; ++i) { lo += red_lo[i][pair]; hi += red_hi[i][pair]; }
    Sum* const row = rows + (size_t)k * SART_SCAN_ROW;
    if constexpr (FIXED) {
      if (j == 2) {
        row[SART_SCAN_N_PASSED] += (hi << kFixedLimbBits) + lo;
      } else {
        const int s = j == 0 ? SART_SCAN_SUM_WEIGHTS : SART_SCAN_SUM_WEIGHTS_SQ, sh = j == 0 ? SART_SCAN_SUM_WEIGHTS_HI : SART_SCAN_SUM_WEIGHTS_SQ_HI;
        lo += row[s];
        row[s] = lo & kMask;
        row[sh] += hi + (lo >> kFixedLimbBits);
      }
    } else {
      row[j == 0 ? SART_SCAN_SUM_WEIGHTS : j == 1 ? SART_SCAN_SUM_WEIGHTS_SQ : SART_SCAN_N_PASSED] += lo;
    }
  }
  if (shared_row && pair < 4) {
    const int dst = pair == 0 ? SART_SCAN_N_REACHED_TELESCOPE : pair == 1 ? SART_SCAN_N_SHELL_SELECTED : pair == 2 ? SART_SCAN_N_HIT_NICKEL : SART_SCAN_N_ON_DETECTOR;
    Sum t = 0;
    for (int i = 0; i < kGroups; ++i) t += red_cnt[i][pair];
    shared_row[dst] += t;
    if (pair == 0) shared_row[SART_SCAN_N_RAYS] += (Sum)n_rays;
  }
}

// What the finalize kernels report about a raw SART_ACCUM_FIXED64 accumulator (OR-ed into a word of the context that the next
// sart_synchronize reads): integers that no longer mean what they should.
constexpr uint32_t kFixedStatusWrapped = 1u;       // a slot is negative or >= 2^62: wrapped, or about to (weights and counts are >= 0)
constexpr uint32_t kFixedStatusUnresolved = 2u;    // the accumulated weights average below 2^12 quanta per passed ray
constexpr uint32_t kFixedStatusNotConserved = 4u;  // pixels (+ outside) / radial / energy weight bins do not add up to SUM_WEIGHTS: a slot wrapped
// Scratch of one finalize (device memory of the context; word 0 = the status, kept; the rest zeroed before every finalize): exact
// two-limb sums of the raw slots that must add up to SUM_WEIGHTS.  Every passed ray adds the SAME integer to SUM_WEIGHTS and to one
// pixel (or to SUM_WEIGHTS_OUTSIDE), one radial bin and one energy bin, so the equalities hold exactly - unless a slot wrapped,
// however often: each wrap takes 2^64 out of its sum.
struct FixedCheck {
  uint32_t status, _pad;
  long long pix_lo, pix_hi, rad_lo, rad_hi, en_lo, en_hi;   // sums over the slots
  long long want_in_lo, want_in_hi;                         // SUM_WEIGHTS - SUM_WEIGHTS_OUTSIDE
  long long want_lo, want_hi;                               // SUM_WEIGHTS
  long long spectra;
  // (not zeroed per finalize) the sums of the FIRST finalize whose conservation check failed, for the error message: which sum is
  // off, in which direction and by how much says whether a slot wrapped (a multiple of 2^64 quanta) or an update went astray
  long long fail[10];                                       // pix, want_in, rad, en, want: (lo, hi) each
};
__device__ __forceinline__ void fixed_check_add(long long v, long long v_hi, bool mine, long long* lo_dst, long long* hi_dst) {
  constexpr long long kMask = (1ll << kFixedLimbBits) - 1;
  if (!__builtin_amdgcn_ballot_w64(mine)) return;   // wave-uniform
  // (v_hi: the slot's limb in the roll-over array, in units of 2^40 - sart_rollover_accumulator_device; 0 without one)
  long long lo = wave_sum_i64(mine ? (v & kMask) : 0ll), hi = wave_sum_i64(mine ? ((v >> kFixedLimbBits) + v_hi) : 0ll);
  if ((threadIdx.x & 63) == 0) {
    hi += lo >> kFixedLimbBits;
    lo &= kMask;
    atomicAdd(reinterpret_cast<unsigned long long*>(lo_dst), (unsigned long long)lo);
    atomicAdd(reinterpret_cast<unsigned long long*>(hi_dst), (unsigned long long)hi);
  }
}
__global__ void fixed_check_kernel(FixedCheck* C) {
  constexpr long long kMask = (1ll << kFixedLimbBits) - 1;
  auto same = [&](long long alo, long long ahi, long long blo, long long bhi) {
    ahi += alo >> kFixedLimbBits; alo &= kMask;
    bhi += blo >> kFixedLimbBits; blo &= kMask;
    return alo == blo && ahi == bhi;
  };
  bool ok = same(C->pix_lo, C->pix_hi, C->want_in_lo, C->want_in_hi);
  if (C->spectra) ok = ok && same(C->rad_lo, C->rad_hi, C->want_lo, C->want_hi) && same(C->en_lo, C->en_hi, C->want_lo, C->want_hi);
  if (!ok) {
    if (!(C->status & kFixedStatusNotConserved)) {
      const long long v[10] = {C->pix_lo, C->pix_hi, C->want_in_lo, C->want_in_hi, C->rad_lo, C->rad_hi, C->en_lo, C->en_hi, C->want_lo, C->want_hi};
      for (int i = 0; i < 10; ++i) C->fail[i] = v[i];
    }
    atomicOr(&C->status, kFixedStatusNotConserved);
  }
}
__device__ __forceinline__ void fixed_status_check_slot(long long v, uint32_t* status) {
  if (v < 0 || v >= (1ll << 62)) atomicOr(status, kFixedStatusWrapped);
}
// Returns false if the squared weights average below 2^6 quanta per passed ray: the integers do not resolve them (their quantum
// 2^-39 of the squared weight bound is as fine as an int64 per workgroup allows); SUM_WEIGHTS_SQ - an error estimate, nothing
// else depends on it - then reads NaN in the f64 output instead of a number that looks like one.
// Both means are judged only once kFixedMinRaysForMeans rays have passed: the check is there to catch a weight BOUND that is off by
// orders of magnitude (an outlier in a table), and the average of a handful of rays says nothing about that - one faint ray alone
// in an accumulator (a launch of a single low-energy CAST ray, weight 1e-8 of the bound) is exact to half a quantum like every
// other and no reason to fail the call (found by the round-6 stream: ray 0 of seed 5 is such a ray).
constexpr double kFixedMinRaysForMeans = 256.0;
__device__ __forceinline__ bool fixed_status_check_means(double quanta_w, double quanta_w2, double n_passed, uint32_t* status) {
  if (!(n_passed >= kFixedMinRaysForMeans)) return true;
  if (quanta_w < 4096.0 * n_passed) atomicOr(status, kFixedStatusUnresolved);
  return quanta_w2 >= 64.0 * n_passed;
}

// Raw SART_ACCUM_FIXED64 accumulator -> the f64 layout of include/sart.h (sart_finalize_accumulator_device).  Element-wise, so
// `out` may alias `in`; the scalars are converted by one thread, which reads all of them before it writes.
struct FinalizeArgs {
  long long n_img;
  int32_t spectra, n_radial_bins, n_energies1, _pad;
  double q_w, q_w2, q_pos, q_refl;
};
// `hi` (or nullptr): the roll-over limbs of sart_rollover_accumulator_device - slot i then stands for hi[i] 2^40 + in[i].
__global__ __launch_bounds__(256) void finalize_fixed_kernel(const long long* in, const long long* hi, double* out, FinalizeArgs F, FixedCheck* C) {
  uint32_t* const status = &C->status;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long n_scalar0 = F.n_img, n_spec0 = F.n_img + SART_ACC_COUNT;
  const long long total = n_spec0 + (F.spectra ? 2ll * F.n_radial_bins + 3ll * F.n_energies1 : 0ll);
  const double two40 = (double)(1ll << kFixedLimbBits);
  {
    // conservation sums (whole waves take part: the block size divides into waves, the last block is padded with idle threads)
    const long long v = i < total ? in[i] : 0ll;
    const long long h = (hi && i < total) ? hi[i] : 0ll;
    const long long j = i - n_spec0, nr = F.n_radial_bins, ne = F.n_energies1;
    fixed_check_add(v, h, i < n_scalar0, &C->pix_lo, &C->pix_hi);
    if (F.spectra) {
      fixed_check_add(v, h, j >= nr && j < 2 * nr, &C->rad_lo, &C->rad_hi);
      fixed_check_add(v, h, j >= 2 * nr + ne && j < 2 * nr + 2 * ne, &C->en_lo, &C->en_hi);
    }
  }
  if (i < n_scalar0) {
    const long long v = in[i];
    fixed_status_check_slot(v, status);
    // power-of-two quantum: the product is exact; with a roll-over limb the sum of the two exact doubles rounds once
    out[i] = (hi ? fma((double)hi[i], two40, (double)v) : (double)v) * F.q_w;
  } else if (i == n_scalar0) {
    long long v[SART_ACC_COUNT];
    double vd[SART_ACC_COUNT];     // the slot's value with its roll-over limb
    for (int k = 0; k < SART_ACC_COUNT; ++k) {
      v[k] = in[n_scalar0 + k];
      fixed_status_check_slot(v[k], status);
      const long long h = hi ? hi[n_scalar0 + k] : 0ll;
      if (h < 0) atomicOr(status, kFixedStatusWrapped);
      vd[k] = hi ? fma((double)h, two40, (double)v[k]) : (double)v[k];
    }
    auto limbs = [&](int k, int kh) {                      // hi * 2^40 and lo are exact doubles: one rounding in the sum
      return vd[kh] * two40 + vd[k];
    };
    const bool sq_ok = fixed_status_check_means(limbs(SART_ACC_SUM_WEIGHTS, SART_ACC_SUM_WEIGHTS_HI),
                                                limbs(SART_ACC_SUM_WEIGHTS_SQ, SART_ACC_SUM_WEIGHTS_SQ_HI), vd[SART_ACC_N_PASSED], status);
    // SUM_WEIGHTS in units of the quantum as (lo, hi 2^40): its two slots, each with its roll-over limb (the limb of the *_HI slot
    // counts 2^80 quanta: only a sum beyond 2^102 quanta would not fit the int64 below)
    auto hi_of = [&](int k, int kh) {
      return v[kh] + (hi ? hi[n_scalar0 + k] + (hi[n_scalar0 + kh] << kFixedLimbBits) : 0ll);
    };
    C->want_lo = v[SART_ACC_SUM_WEIGHTS];
    C->want_hi = hi_of(SART_ACC_SUM_WEIGHTS, SART_ACC_SUM_WEIGHTS_HI);
    C->want_in_lo = v[SART_ACC_SUM_WEIGHTS] - v[SART_ACC_SUM_WEIGHTS_OUTSIDE];
    C->want_in_hi = C->want_hi - hi_of(SART_ACC_SUM_WEIGHTS_OUTSIDE, SART_ACC_SUM_WEIGHTS_OUTSIDE_HI);
    C->spectra = F.spectra;
    for (int k = 0; k < SART_ACC_COUNT; ++k) {
      double r;
      switch (k) {
        case SART_ACC_SUM_WEIGHTS: r = limbs(k, SART_ACC_SUM_WEIGHTS_HI) * F.q_w; break;
        case SART_ACC_SUM_X: r = limbs(k, SART_ACC_SUM_X_HI) * F.q_pos; break;
        case SART_ACC_SUM_Y: r = limbs(k, SART_ACC_SUM_Y_HI) * F.q_pos; break;
        case SART_ACC_SUM_R: r = limbs(k, SART_ACC_SUM_R_HI) * F.q_pos; break;
        case SART_ACC_SUM_WEIGHTS_SQ: r = sq_ok ? limbs(k, SART_ACC_SUM_WEIGHTS_SQ_HI) * F.q_w2 : __builtin_nan(""); break;
        case SART_ACC_SUM_WEIGHTS_OUTSIDE: r = 0.0; break;   // (a raw-accumulator slot: the f64 layout has none)
        default: r = fixed_is_hi_slot(k) ? 0.0 : vd[k]; break;   // counters (the reserved slots hold 0)
      }
      out[n_scalar0 + k] = r;
    }
  } else if (F.spectra && i >= n_spec0) {
    // radial_counts | radial_weights | energy_counts | energy_weights | energy_reflect
    const long long j = i - n_spec0, nr = F.n_radial_bins, ne = F.n_energies1;
    if (j < 2 * nr + 3 * ne) {
      const double q = j < nr ? 1.0 : j < 2 * nr ? F.q_w : j < 2 * nr + ne ? 1.0 : j < 2 * nr + 2 * ne ? F.q_w : F.q_refl;
      const long long v = in[i];
      fixed_status_check_slot(v, status);
      out[i] = (hi ? fma((double)hi[i], two40, (double)v) : (double)v) * q;
    }
  }
}

// sart_rollover_accumulator_device: the bits of every slot above 2^40 move into the slot's limb in `hi` (units of 2^40); the slot
// keeps [0, 2^40).  A slot that is negative or >= 2^62 has wrapped (or is about to): reported like the finalize kernel does.
__global__ __launch_bounds__(256) void rollover_fixed_kernel(long long* acc, long long* hi, long long n, FixedCheck* C) {
  constexpr long long kMask = (1ll << kFixedLimbBits) - 1;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const long long v = acc[i];
  fixed_status_check_slot(v, &C->status);
  const long long h = hi[i] + (v >> kFixedLimbBits);
  if (h < 0) atomicOr(&C->status, kFixedStatusWrapped);
  hi[i] = h;
  acc[i] = v & kMask;
}

// Raw FIXED64 scan accumulator -> doubles: rows [0, n) of `in` / `out` (a group of up to kScanMaxMasses masses, whose quanta
// travel in the kernel arguments) and, if `shared_row`, the counter row behind the whole scan.  One thread per row.
struct FinalizeScanArgs {
  int32_t n_masses, shared_row;    // shared_row: row index of the counters relative to `in`, or -1
  uint32_t counter_slots, _pad;    // bit j < 8: slot j of a row is a plain counter (mass scan: N_PASSED; angular scan: its four counters)
                                   // bit 31 (kFinalizeScanRowsUnresolvedNaN): see finalize_scan_kernel
  double q_w[kScanMaxMasses], q_w2[kScanMaxMasses];
};
// Unresolved weights (the row's SUM_WEIGHTS averages below 2^12 quanta per passed ray).  The mass scan has a quantum per mass, derived
// from that mass's own weight bound: an unresolved row there means the bound is off, and the whole scan reports
// SART_ERR_ACCUMULATOR.  The angular scan has ONE quantum for all its angles, from the on-axis weight bound - an angle far off axis
// can pass a few rays of tiny R1 R2 that it does not resolve while every other row is fine: with bit 31 of counter_slots that row's
// SUM_WEIGHTS reads NaN (as an unresolved SUM_WEIGHTS_SQ does everywhere) and the status stays clean.
__global__ __launch_bounds__(64) void finalize_scan_kernel(const long long* in, double* out, FinalizeScanArgs F, FixedCheck* C) {
  uint32_t* const status = &C->status;
  const int k = threadIdx.x;
  long long v[SART_SCAN_ROW];
  if (k < F.n_masses) {
    for (int j = 0; j < SART_SCAN_ROW; ++j) { v[j] = in[(size_t)k * SART_SCAN_ROW + j]; fixed_status_check_slot(v[j], status); }
    const double two40 = (double)(1ll << kFixedLimbBits);
    const double sw = (double)v[SART_SCAN_SUM_WEIGHTS_HI] * two40 + (double)v[SART_SCAN_SUM_WEIGHTS];
    const double sw2 = (double)v[SART_SCAN_SUM_WEIGHTS_SQ_HI] * two40 + (double)v[SART_SCAN_SUM_WEIGHTS_SQ];
    const double n_passed = (double)v[SART_SCAN_N_PASSED];
    bool w_ok = true, sq_ok;
    if (F.counter_slots >> 31) {
      w_ok = !(n_passed >= kFixedMinRaysForMeans) || sw >= 4096.0 * n_passed;
      sq_ok = !(n_passed >= kFixedMinRaysForMeans) || sw2 >= 64.0 * n_passed;
    } else {
      sq_ok = fixed_status_check_means(sw, sw2, n_passed, status);
    }
    double* const o = out + (size_t)k * SART_SCAN_ROW;
    for (int j = 0; j < SART_SCAN_ROW; ++j) o[j] = 0.0;
    o[SART_SCAN_SUM_WEIGHTS] = w_ok ? sw * F.q_w[k] : __builtin_nan("");
    o[SART_SCAN_SUM_WEIGHTS_SQ] = sq_ok ? sw2 * F.q_w2[k] : __builtin_nan("");
    for (int j = 0; j < SART_SCAN_ROW; ++j)
      if ((F.counter_slots >> j) & 1u) o[j] = (double)v[j];   // (SART_SCAN_ROW = 8: bit 31 is not a slot)
  } else if (k == 63 && F.shared_row >= 0) {
    for (int j = 0; j < SART_SCAN_ROW; ++j) { v[j] = in[(size_t)F.shared_row * SART_SCAN_ROW + j]; fixed_status_check_slot(v[j], status); }
    for (int j = 0; j < SART_SCAN_ROW; ++j) out[(size_t)F.shared_row * SART_SCAN_ROW + j] = (double)v[j];
  }
}

// Literal drop-in for traceAxionWrapper: one Axion record per ray, in ray order (no compaction).
constexpr int kRecBlock = 256;
// `uniforms` != nullptr (sart_internal_trace_records_uniforms, a test entry): ray i takes its six uniforms from
// uniforms[6 i .. 6 i + 5] (draw order of SURVEY App. B) instead of from its Philox blocks - physics fixtures keyed by
// explicit uniforms survive a re-mapping of the random stream.
// One record to memory as 26 eight-byte words.  The two groups of byte-sized fields are packed by hand: a plain struct copy moves
// their padding arrays, too, and the compiler then keeps those bytes of the local record in scratch memory.
__device__ __forceinline__ void store_record(sart_axion_t* dst, const sart_axion_t& r) {
  static_assert(sizeof(sart_axion_t) == 208 && offsetof(sart_axion_t, pointdataX) == 8 && offsetof(sart_axion_t, kinds) == 128 &&
                    offsetof(sart_axion_t, transProbWindow) == 136 && offsetof(sart_axion_t, shellNumber) == 176,
                "layout of include/sart.h");
  uint64_t* o = reinterpret_cast<uint64_t*>(dst);
  auto bits = [](double v) { return (uint64_t)__double_as_longlong(v); };
  o[0] = (uint64_t)r.passed | ((uint64_t)r.passedTillWindow << 8) | ((uint64_t)r.hitNickel << 16);
  o[1] = bits(r.pointdataX); o[2] = bits(r.pointdataY); o[3] = bits(r.pointdataXBefore); o[4] = bits(r.pointdataYBefore);
  o[5] = bits(r.pointdataR); o[6] = bits(r.weights); o[7] = bits(r.weightsAll); o[8] = bits(r.transmissionMagnet);
  o[9] = bits(r.yawAngles); o[10] = bits(r.pixvalsX); o[11] = bits(r.pixvalsY); o[12] = bits(r.radii);
  o[13] = bits(r.energiesAx); o[14] = bits(r.energiesAxAll); o[15] = bits(r.energiesAxWindow);
  o[16] = (uint64_t)r.kinds | ((uint64_t)r.kindsWindow << 8);
  o[17] = bits(r.transProbWindow); o[18] = bits(r.transProbArgon); o[19] = bits(r.transProbDetector); o[20] = bits(r.transProbMagnet);
  o[21] = bits(r.deviationDet); o[22] = (uint64_t)r.shellNumber; o[23] = bits(r.energiesPre); o[24] = bits(r.emratesPre);
  o[25] = bits(r.reflect);
}

__global__ __launch_bounds__(kRecBlock) void trace_records_kernel(HotA H, const DevBlob* __restrict__ blob, TraceArgs A,
                                                                   sart_axion_t* __restrict__ out, HotB HB,
                                                                   const double* __restrict__ uniforms) {
  __shared__ TablesLds S;
  __shared__ DevBlob B;
  {
    const uint64_t* src = reinterpret_cast<const uint64_t*>(blob);
    uint64_t* dst = reinterpret_cast<uint64_t*>(&B);
    for (int i = threadIdx.x; i < (int)(sizeof(DevBlob) / 8); i += kRecBlock) dst[i] = src[i];
    __syncthreads();
  }
  const DevParams& P = B.P;
  stage_tables<kRecBlock>(S, P, B.T);
  const LdsTables L{S.sincos, S.rcdf_hi, B.T.flux_radius_cdf, S.rguide, S.shells, S.lut};

  const uint64_t stride = (uint64_t)gridDim.x * kRecBlock;
  for (uint64_t i = (uint64_t)blockIdx.x * kRecBlock + threadIdx.x; i < A.n_rays; i += stride) {
    sart_axion_t rec = {};   // newSeq[Axion] zero-initialises (:2760)
    RayState st;
    bool sampled, reached;
    double radial;    // not used here: the record path lets phase B find every miss itself (and so checks the shortcut of the histogram path)
    LaneMasks masks;  // not used here either: this kernel branches per lane on the bools
    const uint64_t ray_id = A.ray_id_offset + i;
    bool alive;
    if (uniforms) {   // wave-uniform
      const double* u = uniforms + 6 * i;
      const Uniforms U{u[0], u[1], u[2], u[3], u[4], u[5], upper32_of_uniform(u[2])};
      alive = phase_a_core<false, -1, false>(H, P, L, U, st, sampled, reached, radial, masks);
    } else {
      const uint32_t u3_hi = word_of(stream_block(ray_id >> 2, A.seed_lo, A.seed_hi), (uint32_t)ray_id & 3u);
      alive = phase_a<false, -1, false>(H, P, L, A.seed_lo, A.seed_hi, ray_id, u3_hi, st, sampled, reached, radial, masks);
    }
    int e_idx = -1;
    if (sampled) {
      e_idx = H.test_active ? P.n_energies : sample_energy_index(HB, st.r_idx, st.u5);
      rec.emratesPre = 1.0;                          // :1818
      rec.energiesPre = load_energy_row(HB, e_idx).energy;  // :1819
    }
    if (ballot64(alive)) {
      RayOut ro;
      if (!sampled) { st.u5 = 0.0; st.r_idx = 0; }
      phase_b<true, false, -1, false>(P, L, HB, A, st, e_idx >= 0 ? e_idx : 0, alive, ro, &rec);
    }
    store_record(&out[i], rec);
  }
}

// ---- passed rays only (include/sart.h: sart_trace_records_passed): the records of one chunk, compacted in ray order ----------
// What generateResultPlots (raytracer.nim:2252-2283) and the scan sum (:2800) read of the record buffer: the records with
// `passed` set, and how many records have passedTillWindow / hitNickel set.  Three small kernels behind trace_records_kernel on
// the same stream: flags -> per-block counts, one-block exclusive scan, cooperative copy (each wave moves the records of its
// passed lanes as one contiguous run of 8-byte words: coalesced stores, 208-byte runs of loads).  Integer arithmetic only: the
// order and the counts do not depend on the launch geometry.
constexpr int kCompactBlock = 1024;                  // records per block of the three kernels
constexpr int kCompactMaxBlocks = 1024;              // blocks one scan covers: chunks of at most 2^20 records
__device__ __forceinline__ uint32_t record_flags(const sart_axion_t* rec, uint32_t i, uint32_t n) {
  return i < n ? *reinterpret_cast<const uint32_t*>(rec + i) : 0u;   // passed @0, passedTillWindow @1, hitNickel @2
}
__global__ __launch_bounds__(kCompactBlock) void records_count_kernel(const sart_axion_t* __restrict__ rec, uint32_t n,
                                                                       uint32_t* __restrict__ block_counts,
                                                                       unsigned long long* __restrict__ counts) {
  __shared__ uint32_t part[kCompactBlock / 64][3];
  const uint32_t f = record_flags(rec, blockIdx.x * kCompactBlock + threadIdx.x, n);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const uint32_t np = (uint32_t)__popcll(ballot64((f & 0xFFu) != 0u)), nw = (uint32_t)__popcll(ballot64((f & 0xFF00u) != 0u)),
                 nn = (uint32_t)__popcll(ballot64((f & 0xFF0000u) != 0u));
  if (lane == 0) { part[wave][0] = np; part[wave][1] = nw; part[wave][2] = nn; }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t a = 0, b = 0, c = 0;
    for (int w = 0; w < kCompactBlock / 64; ++w) { a += part[w][0]; b += part[w][1]; c += part[w][2]; }
    block_counts[blockIdx.x] = a;
    if (b) atomicAdd(&counts[2], (unsigned long long)b);
    if (c) atomicAdd(&counts[3], (unsigned long long)c);
  }
}
// block_counts[b] -> the index of block b's first passed record in the output (counts[1] = passed records of earlier chunks);
// counts[0] += n, counts[1] += the passed records of this chunk
__global__ __launch_bounds__(kCompactMaxBlocks) void records_scan_kernel(uint32_t* __restrict__ block_counts, uint32_t n_blocks,
                                                                          unsigned long long* __restrict__ block_first,
                                                                          unsigned long long* __restrict__ counts, uint32_t n) {
  __shared__ uint32_t wave_sum[kCompactMaxBlocks / 64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const uint32_t v = threadIdx.x < n_blocks ? block_counts[threadIdx.x] : 0u;
  uint32_t incl = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t o = (uint32_t)__shfl_up((int)incl, d, 64);
    if (lane >= d) incl += o;
  }
  if (lane == 63) wave_sum[wave] = incl;
  __syncthreads();
  uint32_t before = 0, total = 0;
  for (int w = 0; w < kCompactMaxBlocks / 64; ++w) { before += w < wave ? wave_sum[w] : 0u; total += wave_sum[w]; }
  const unsigned long long base = counts[1];
  if (threadIdx.x < n_blocks) block_first[threadIdx.x] = base + before + (incl - v);
  __syncthreads();   // every thread has read counts[1]
  if (threadIdx.x == 0) { counts[0] += n; counts[1] = base + total; }
}
__global__ __launch_bounds__(kCompactBlock) void records_scatter_kernel(const sart_axion_t* __restrict__ rec, uint32_t n,
                                                                         const unsigned long long* __restrict__ block_first,
                                                                         sart_axion_t* __restrict__ out, unsigned long long capacity) {
  __shared__ uint32_t wave_count[kCompactBlock / 64];
  __shared__ uint8_t src_lane[kCompactBlock / 64][64];
  const uint32_t i = blockIdx.x * kCompactBlock + threadIdx.x;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const bool passed = (record_flags(rec, i, n) & 0xFFu) != 0u;
  const uint64_t m = ballot64(passed);
  const uint32_t cnt = (uint32_t)__popcll(m);
  const uint32_t prefix = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
  if (passed) src_lane[wave][prefix] = (uint8_t)lane;
  if (lane == 0) wave_count[wave] = cnt;
  __syncthreads();
  uint32_t before = 0;
  for (int w = 0; w < wave; ++w) before += wave_count[w];
  const unsigned long long first = block_first[blockIdx.x] + before;   // output index of this wave's first passed record
  constexpr uint32_t kWords = sizeof(sart_axion_t) / 8;                 // 26
  const uint64_t* src = reinterpret_cast<const uint64_t*>(rec + (blockIdx.x * kCompactBlock + wave * 64));
  uint64_t* dst = reinterpret_cast<uint64_t*>(out);
  for (uint32_t e = (uint32_t)lane; e < cnt * kWords; e += 64u) {
    const uint32_t r = e / kWords, w = e - r * kWords;
    if (first + r < capacity) dst[(first + r) * kWords + w] = src[(uint32_t)src_lane[wave][r] * kWords + w];
  }
}

// The offsets the kernels assume for their own arguments when they re-read them from the kernel-argument segment
// (reload_hot / reload_zones / reload_kernarg).  Not part of the C-ABI: tests/test_host_and_abi.py compares them with the
// argument offsets in the code object's metadata, so that a compiler that lays arguments out differently fails a CPU test
// instead of faulting on the GPU.
extern "C" __attribute__((visibility("default"))) void sart_internal_kernarg_layout(int32_t out[8]) {
  out[7] = (int32_t)sizeof(AScanKernArgs);   // (the angular-scan kernel: same offsets, its own last argument)
  out[0] = (int32_t)offsetof(HistKernArgs, H);
  out[1] = (int32_t)offsetof(HistKernArgs, blob);
  out[2] = (int32_t)offsetof(HistKernArgs, A);
  out[3] = (int32_t)offsetof(HistKernArgs, acc);
  out[4] = (int32_t)offsetof(HistKernArgs, HB);
  out[5] = (int32_t)offsetof(HistKernArgs, SC);
  out[6] = (int32_t)sizeof(HistKernArgs);
}

// ---- device math under test (tests/test_gpu_math.py): evaluates one helper on an array, not part of the C-ABI ----
__global__ void math_eval_kernel(int fn, const double* __restrict__ in, double* __restrict__ out, int n,
                                 const double* __restrict__ sincos_tab) {
  __shared__ double tab[2 * kSinCosEntries];
  for (int i = threadIdx.x; i < 2 * kSinCosEntries; i += blockDim.x) tab[i] = sincos_tab[i];
  __syncthreads();
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double x = in[i];
  double r = 0.0, sn, cs;
  const SinCosCoef K = sincos_coef();
  switch (fn) {
    case 0: r = frcp(x); break;
    case 1: r = frsq(x); break;
    case 2: r = fsqrt_pos(x); break;
    case 3: sincos_turns<2>(x, tab, K, &sn, &cs); r = sn; break;
    case 4: sincos_turns<2>(x, tab, K, &sn, &cs); r = cs; break;
    case 5: sincos_turns<1>(x, tab, K, &sn, &cs); r = sn; break;
    case 6: sincos_turns<1>(x, tab, K, &sn, &cs); r = cs; break;
    case 7: r = asin_small(x, x * x); break;
    case 8: r = cos_yaw_of_slope(x); break;
    case 9: r = atan_small(x); break;
    case 10: r = cos_small(x); break;
    case 11: r = fsqrt(x); break;
    case 12: r = exp_neg(x); break;
    case 13: r = cos_any(x, tab, K); break;
    case 14: r = cos_any_vvs(x, tab, GasCos::make()); break;
    default: break;
  }
  out[i] = r;
}
extern "C" __attribute__((visibility("default"))) int sart_internal_math_eval(int fn, const double* in_host, double* out_host, int n,
                                                                               const double* sincos_table_host) {
  double *d_in = nullptr, *d_out = nullptr, *d_tab = nullptr;
  if (n < 1 || hipMalloc(&d_in, (size_t)n * 8) != hipSuccess || hipMalloc(&d_out, (size_t)n * 8) != hipSuccess ||
      hipMalloc(&d_tab, 2 * kSinCosEntries * 8) != hipSuccess)
    return -1;
  (void)hipMemcpy(d_in, in_host, (size_t)n * 8, hipMemcpyHostToDevice);
  (void)hipMemcpy(d_tab, sincos_table_host, 2 * kSinCosEntries * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(math_eval_kernel, dim3((n + 255) / 256), dim3(256), 0, nullptr, fn, d_in, d_out, n, d_tab);
  const hipError_t e = hipMemcpy(out_host, d_out, (size_t)n * 8, hipMemcpyDeviceToHost);
  (void)hipFree(d_in); (void)hipFree(d_out); (void)hipFree(d_tab);
  return e == hipSuccess ? 0 : -2;
}

// ---- launch wrappers (called from sart_api.hip) ----
int records_block() { return kRecBlock; }
// Variants: 0 = specialised (solar source, no hole loop) vacuum, not rotated; 1 = generic, not rotated; 2 = generic, rotated;
// 3 = specialised, gas stage; 4 = specialised, rotated; 5 = variant 0 with the constant path in the magnetic field (PATHC);
// 6 = variant 3 with the constant path.  All with 1024 threads = 4 waves / SIMD (measured fastest of 256 / 512 / 768 / 1024).
// The fused mass scan exists for the variants that can run the gas stage: 1, 2, 3, 6.
int histogram_block_of(int) { return 1024; }

#define SART_HIST_VARIANTS(X) \
  X(0, true, false, 0, false) X(1, false, false, -1, false) X(2, false, true, -1, false) X(3, true, false, 1, false) \
  X(4, true, true, 0, false) X(5, true, false, 0, true) X(6, true, false, 1, true)
#define SART_SCAN_VARIANTS(X) X(1, false, false, -1, false) X(2, false, true, -1, false) X(3, true, false, 1, false) X(6, true, false, 1, true)

int histogram_blocks_per_cu(int variant) {   // the FIXED64 and SCAN instantiations use the same LDS and launch bounds
  int n = 0;
  hipError_t e = hipErrorInvalidValue;
  switch (variant) {
#define X(ID, F, R, G, PC) case ID: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, trace_histogram_kernel<1024, F, R, G, PC, false>, 1024, 0); break;
    SART_HIST_VARIANTS(X)
#undef X
    default: break;
  }
  return (e == hipSuccess && n > 0) ? n : 1;
}

void launch_trace_histogram(const HotA& H, const HotB& HB, const DevBlob* blob, const TraceArgs& A, double* acc, int n_blocks,
                            hipStream_t stream, int variant, bool fixed) {
  ScanArgs SC;
  memset(&SC, 0, sizeof SC);   // n_masses = 0: not a scan
  switch (variant + (fixed ? 100 : 0)) {
#define X(ID, F, R, G, PC) \
    case ID: hipLaunchKernelGGL((trace_histogram_kernel<1024, F, R, G, PC, false>), dim3(n_blocks), dim3(1024), 0, stream, H, blob, A, acc, HB, SC); break; \
    case 100 + ID: hipLaunchKernelGGL((trace_histogram_kernel<1024, F, R, G, PC, true>), dim3(n_blocks), dim3(1024), 0, stream, H, blob, A, acc, HB, SC); break;
    SART_HIST_VARIANTS(X)
#undef X
    default: return;
  }
  const int n_img = A.image_nx * A.image_ny;
  if (fixed) {
    long long* const acc_i = reinterpret_cast<long long*>(acc);
    hipLaunchKernelGGL(fold_scalars_fixed_kernel, dim3(1), dim3(256), 0, stream, acc_i + n_img,
                       reinterpret_cast<const long long*>(A.partials), n_blocks, (long long)A.n_rays);
    if (A.replica_mask != 0u)
      hipLaunchKernelGGL(fold_replicas_kernel<long long>, dim3((n_img + 255) / 256), dim3(256), 0, stream, acc_i,
                         reinterpret_cast<long long*>(A.replicas), n_img, (int)A.replica_mask + 1, A.replica_stride);
    return;
  }
  hipLaunchKernelGGL(fold_scalars_kernel, dim3(1), dim3(256), 0, stream, acc + n_img, A.partials, n_blocks, (double)A.n_rays);
  if (A.replica_mask != 0u)
    hipLaunchKernelGGL(fold_replicas_kernel<double>, dim3((n_img + 255) / 256), dim3(256), 0, stream, acc, A.replicas, n_img,
                       (int)A.replica_mask + 1, A.replica_stride);
}

// One group of SC.n_masses <= kScanMaxMasses masses over the rays of A: trace + per-mass accumulation, then the fold into
// `rows` (the group's first row of the scan accumulator) and, if `shared_row`, into the scan's counter row.
bool launch_trace_mass_scan(const HotA& H, const HotB& HB, const DevBlob* blob, const TraceArgs& A, const ScanArgs& SC, double* rows,
                            double* shared_row, int n_blocks, hipStream_t stream, int variant, bool fixed) {
  double* const no_acc = nullptr;   // a scan accumulates no image and no spectra
  switch (variant + (fixed ? 100 : 0)) {
#define X(ID, F, R, G, PC) \
    case ID: hipLaunchKernelGGL((trace_histogram_kernel<1024, F, R, G, PC, false, true>), dim3(n_blocks), dim3(1024), 0, stream, H, blob, A, no_acc, HB, SC); break; \
    case 100 + ID: hipLaunchKernelGGL((trace_histogram_kernel<1024, F, R, G, PC, true, true>), dim3(n_blocks), dim3(1024), 0, stream, H, blob, A, no_acc, HB, SC); break;
    SART_SCAN_VARIANTS(X)
#undef X
    default: return false;
  }
  if (fixed)
    hipLaunchKernelGGL(fold_scan_kernel<true>, dim3(1), dim3(1024), 0, stream, rows, shared_row, A.partials, SC.partials, n_blocks, SC.n_masses, (double)A.n_rays);
  else
    hipLaunchKernelGGL(fold_scan_kernel<false>, dim3(1), dim3(1024), 0, stream, rows, shared_row, A.partials, SC.partials, n_blocks, SC.n_masses, (double)A.n_rays);
  return true;
}

// One group of AN.n_angles <= kAScanMaxAngles telescope angles over the rays of A: trace + per-angle accumulation, then the fold
// into `rows` (the group's first row of the scan accumulator) and, if `shared_row`, into the scan's counter row.  `fast`: the
// specialised instantiation (solar source, no hole loop, vacuum), else the one that reads those switches at run time.
int angular_scan_blocks_per_cu(bool fast) {
  int n = 0;
  const hipError_t e = fast ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, trace_angular_scan_kernel<1024, true, 0, false>, 1024, 0)
                            : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, trace_angular_scan_kernel<1024, false, -1, false>, 1024, 0);
  return (e == hipSuccess && n > 0) ? n : 1;
}
void launch_trace_angular_scan(const HotA& H, const HotB& HB, const DevBlob* blob, const TraceArgs& A, const AScanArgs& AN, double* rows,
                               double* shared_row, int n_blocks, hipStream_t stream, bool fast, bool fixed) {
  double* const no_acc = nullptr;
  switch ((fast ? 2 : 0) + (fixed ? 1 : 0)) {
    case 0: hipLaunchKernelGGL((trace_angular_scan_kernel<1024, false, -1, false>), dim3(n_blocks), dim3(1024), 0, stream, H, blob, A, no_acc, HB, AN); break;
    case 1: hipLaunchKernelGGL((trace_angular_scan_kernel<1024, false, -1, true>), dim3(n_blocks), dim3(1024), 0, stream, H, blob, A, no_acc, HB, AN); break;
    case 2: hipLaunchKernelGGL((trace_angular_scan_kernel<1024, true, 0, false>), dim3(n_blocks), dim3(1024), 0, stream, H, blob, A, no_acc, HB, AN); break;
    default: hipLaunchKernelGGL((trace_angular_scan_kernel<1024, true, 0, true>), dim3(n_blocks), dim3(1024), 0, stream, H, blob, A, no_acc, HB, AN); break;
  }
  if (fixed)
    hipLaunchKernelGGL(fold_ascan_kernel<true>, dim3(1), dim3(1024), 0, stream, rows, shared_row, AN.partials, n_blocks, AN.n_angles, (double)A.n_rays);
  else
    hipLaunchKernelGGL(fold_ascan_kernel<false>, dim3(1), dim3(1024), 0, stream, rows, shared_row, AN.partials, n_blocks, AN.n_angles, (double)A.n_rays);
}

size_t fixed_check_bytes() { return sizeof(FixedCheck); }
// "pixels 123 + 456 x 2^40 quanta, wanted ..." from a host copy of the context's FixedCheck (the error message of a failed
// conservation check, sart_api.hip: status_take)
std::string fixed_check_describe(const void* host_copy) {
  const FixedCheck& C = *static_cast<const FixedCheck*>(host_copy);
  constexpr long long kMask = (1ll << kFixedLimbBits) - 1;
  auto norm = [&](long long lo, long long hi, long long& nlo, long long& nhi) { nhi = hi + (lo >> kFixedLimbBits); nlo = lo & kMask; };
  const char* names[5] = {"pixels", "SUM_WEIGHTS - outside", "radial bins", "energy bins", "SUM_WEIGHTS"};
  std::string s = " [quanta as lo + hi x 2^40:";
  for (int i = 0; i < 5; ++i) {
    if ((i == 2 || i == 3) && !C.spectra) continue;
    long long lo, hi;
    norm(C.fail[2 * i], C.fail[2 * i + 1], lo, hi);
    s += std::string(i ? "," : "") + " " + names[i] + " " + std::to_string(lo) + " + " + std::to_string(hi);
  }
  return s + "]";
}
void launch_rollover_fixed(void* acc, void* hi, size_t n, void* check_dev, hipStream_t stream) {
  hipLaunchKernelGGL(rollover_fixed_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, static_cast<long long*>(acc),
                     static_cast<long long*>(hi), (long long)n, static_cast<FixedCheck*>(check_dev));
}
void launch_finalize_fixed(const void* in, const void* hi, double* out, size_t n_img, int spectra, int n_radial_bins, int n_energies1, double q_w,
                           double q_w2, double q_pos, double q_refl, void* check_dev, hipStream_t stream) {
  FinalizeArgs F{(long long)n_img, spectra, n_radial_bins, n_energies1, 0, q_w, q_w2, q_pos, q_refl};
  const size_t total = n_img + SART_ACC_COUNT + (spectra ? 2 * (size_t)n_radial_bins + 3 * (size_t)n_energies1 : 0);
  FixedCheck* const C = static_cast<FixedCheck*>(check_dev);
  (void)hipMemsetAsync(reinterpret_cast<char*>(C) + 8, 0, offsetof(FixedCheck, fail) - 8, stream);   // everything but the status word and the failure record
  hipLaunchKernelGGL(finalize_fixed_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream,
                     reinterpret_cast<const long long*>(in), reinterpret_cast<const long long*>(hi), out, F, C);
  hipLaunchKernelGGL(fixed_check_kernel, dim3(1), dim3(1), 0, stream, C);
}
// rows [0, n_masses) of in / out with the quanta q_w[k], q_w2[k]; shared_row >= 0: that row (relative to `in`) holds the counters
void launch_finalize_scan(const void* in, double* out, int n_masses, const double* q_w, const double* q_w2, int shared_row,
                          void* check_dev, hipStream_t stream, uint32_t counter_slots) {
  FinalizeScanArgs F;
  memset(&F, 0, sizeof F);
  F.n_masses = n_masses;
  F.shared_row = shared_row;
  F.counter_slots = counter_slots;
  for (int k = 0; k < n_masses; ++k) { F.q_w[k] = q_w[k]; F.q_w2[k] = q_w2[k]; }
  hipLaunchKernelGGL(finalize_scan_kernel, dim3(1), dim3(64), 0, stream, reinterpret_cast<const long long*>(in), out, F, static_cast<FixedCheck*>(check_dev));
}
int compact_chunk_max() { return kCompactBlock * kCompactMaxBlocks; }
// scratch: block_counts[kCompactMaxBlocks] u32, block_first[kCompactMaxBlocks] u64; counts[4] = {n_rays, n_passed, n_passed_till_window, n_hit_nickel}
void launch_compact_records(const sart_axion_t* rec, uint32_t n, sart_axion_t* out, unsigned long long capacity, uint32_t* block_counts,
                            unsigned long long* block_first, unsigned long long* counts, hipStream_t stream) {
  const uint32_t n_blocks = (n + kCompactBlock - 1) / kCompactBlock;
  hipLaunchKernelGGL(records_count_kernel, dim3(n_blocks), dim3(kCompactBlock), 0, stream, rec, n, block_counts, counts);
  hipLaunchKernelGGL(records_scan_kernel, dim3(1), dim3(kCompactMaxBlocks), 0, stream, block_counts, n_blocks, block_first, counts, n);
  hipLaunchKernelGGL(records_scatter_kernel, dim3(n_blocks), dim3(kCompactBlock), 0, stream, rec, n, block_first, out, capacity);
}
void launch_trace_records(const HotA& H, const HotB& HB, const DevBlob* blob, const TraceArgs& A, sart_axion_t* out, int n_blocks,
                          hipStream_t stream, const double* uniforms_dev) {
  hipLaunchKernelGGL(trace_records_kernel, dim3(n_blocks), dim3(kRecBlock), 0, stream, H, blob, A, out, HB, uniforms_dev);
}

}  // namespace sart
