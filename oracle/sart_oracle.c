/*
 * sart_oracle.c — CPU oracle for the per-ray hot path. TEST INFRASTRUCTURE.
 *
 * Plain-C, f64, line-by-line restatement of the reference's algorithm, written from
 * the source text of /root/reference (jovoy/SolarAxionRayTracing); every function
 * cites the reference lines it follows ("raytracer.nim" = src/raytracer.nim,
 * "axionMassforMagnet.nim" = axionMass/axionMassforMagnet.nim).
 *
 * PARITY STATUS: the reference is Nim and cannot be built here (no Nim compiler, nimble
 * dependencies not vendored, three input files missing - SURVEY.md 8c), and it ships no
 * working tests or golden vectors for this path.  The oracle is therefore pinned only by
 * the known-answer values the reference's text holds (effPhotonMass2 table
 * axionMassforMagnet.nim:116-119, conversion probabilities, window strip geometry,
 * coating map, lengthTelescope, TestMirrors.nim scenario) - see tests/test_oracle_*.py.
 * At the third-party boundary (numericalnim bilinear/linear1D, unchained unit constants,
 * nim-glm normalize, Nim std/random) parity is UNPINNED: those algorithms are restated from
 * their published form; differences are confined to the last ulp / a global weight scale.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 * Quirks of the reference that look like bugs are reproduced on purpose; each is marked
 * "QUIRK".
 *
 * The reference draws from one global xoroshiro128+ stream shared (racily) by all weave
 * threads (raytracer.nim:276, :2234).  Here every ray owns a Philox4x32-10 counter block
 * keyed by (seed, global ray id) - the draw ORDER inside a ray is the reference's
 * (SURVEY.md Appendix B).
 */
#include "sart_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* `real` is double for the oracle proper.  -DSART_ORACLE_LONG_DOUBLE builds the same source in x87
 * 80-bit long double: a DIAGNOSTIC variant used by tests to measure how much of a difference between
 * two f64 implementations is rounding noise of the reference's formulation (it subtracts points of
 * magnitude 1e11..1.5e14 mm) rather than a different algorithm. */
#if defined(SART_ORACLE_QUAD)
#include <quadmath.h>
typedef __float128 real;   /* second diagnostic variant: IEEE binary128 (libquadmath), 113-bit mantissa */
#define M_(f) f##q
#elif defined(SART_ORACLE_LONG_DOUBLE)
typedef long double real;
#define M_(f) f##l
#else
typedef double real;
#define M_(f) f
#endif
#define M_SQRT M_(sqrt)
#define M_SIN M_(sin)
#define M_COS M_(cos)
#define M_TAN M_(tan)
#define M_ASIN M_(asin)
#define M_ACOS M_(acos)
#define M_ATAN2 M_(atan2)
#define M_FLOOR M_(floor)
#define M_CEIL M_(ceil)
#define M_FABS M_(fabs)
#define M_POW M_(pow)
#define M_EXP M_(exp)
#define M_FMAX M_(fmax)
#define M_ROUND M_(round)

/* ------------------------------------------------------------------------------------------
 * small vector helpers (nim-glm Vec3[float]: length, normalize, cross, dot)
 * ---------------------------------------------------------------------------------------- */
typedef struct { real x, y, z; } v3;

static inline v3 V(real x, real y, real z) { v3 r = {x, y, z}; return r; }
static inline v3 vadd(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 vsub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 vscale(real s, v3 a) { return V(s * a.x, s * a.y, s * a.z); }
static inline real vdot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline real vlength(v3 a) { return M_SQRT(vdot(a, a)); }
/* nim-glm: normalize(v) = v * inversesqrt(dot(v, v)) */
static inline v3 vnormalize(v3 a) { return vscale(1.0 / M_SQRT(vdot(a, a)), a); }
static inline v3 vcross(v3 a, v3 b) {
  return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}

/* Nim std/math: degToRad(d) = d * (PI/180), radToDeg(d) = d / (PI/180). */
#define SART_PI 3.14159265358979323846
static const real RAD_PER_DEG = SART_PI / 180.0;
static inline real deg_to_rad(real d) { return d * RAD_PER_DEG; }
static inline real rad_to_deg(real r) { return r / RAD_PER_DEG; }

/* ------------------------------------------------------------------------------------------
 * RNG: Philox4x32-7 (Salmon et al., SC'11: the fewest rounds that pass BigCrush), key = (seed lo, seed hi),
 * counter = (ray id lo, ray id hi, 0, 0).  The six uniforms of a ray are cut out of that one block and of one
 * word of a stream shared by consecutive rays (see sart_oracle_uniforms).  Round 6 of the HIP path went from ten
 * rounds and two blocks per ray to this; the stream is a free parameter of the path (the reference draws from
 * xoroshiro128+, sart_oracle_nim_rand_* below: parity with it is statistical), and this function IS its definition:
 * the HIP kernel (sart_kernels.hip: uniforms_of) and this restatement must change together.
 * ---------------------------------------------------------------------------------------- */
#define SART_PHILOX_ROUNDS 7
static inline void philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                              uint32_t k0, uint32_t k1, uint32_t out[4]) {
  for (int round = 0; round < SART_PHILOX_ROUNDS; ++round) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static inline double fill52(uint32_t hi, uint32_t lo) {
  /* 52 random mantissa bits (hi and the upper 20 bits of lo) under the exponent of 1.0, minus 1.0: Nim's std/random
   * rand(1.0) */
  uint64_t bits = ((uint64_t)hi << 32) | (uint64_t)lo;
  union { uint64_t i; double d; } cv;
  cv.i = 0x3FF0000000000000ull | (bits >> 12);
  return cv.d - 1.0;
}

void sart_oracle_uniforms(uint64_t seed, uint64_t ray_id, double u[6]) {
  /* One counter block (x, y, z, w) + one word s of the shared stream = 160 random bits for the six uniforms, none used twice:
   * the two CDF draws (u2 radius, u5 energy) and the disc radius (u3) are 32-bit fractions, the two angles of the solar point
   * (u0, u1) 21-bit ones, the disc angle (u4) a 22-bit one made of what the other two leave over in their words. */
  uint32_t b[4];
  philox4x32((uint32_t)ray_id, (uint32_t)(ray_id >> 32), 0u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), b);
  /* The word behind u3 (the uniform behind the radius of the point on the bore exit, :418) is word `ray_id` of a
   * word stream with random access: stream[n] = word (n & 3) of the block with counter (n >> 2, 3, 0).  Four
   * consecutive rays share that block, which is what lets the HIP kernel's first stage (rays that this word alone
   * proves dead) cost a quarter of a Philox block per ray. */
  uint32_t sh[4];
  const uint64_t g = ray_id >> 2;
  philox4x32((uint32_t)g, (uint32_t)(g >> 32), 3u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), sh);
  u[2] = fill52(b[0], 0u);
  u[5] = fill52(b[2], 0u);
  u[0] = fill52(b[1] & 0xFFFFF800u, 0u);
  u[1] = fill52(b[3] & 0xFFFFF800u, 0u);
  u[4] = fill52((b[1] << 21) | ((b[3] & 0x7FFu) << 10), 0u);
  u[3] = fill52(sh[ray_id & 3u], 0u);
}

/* ------------------------------------------------------------------------------------------
 * std / third-party pieces
 * ---------------------------------------------------------------------------------------- */
/* std/algorithm.lowerBound: first index i with a[i] >= key (n if none). */
static int64_t lower_bound_r(const double* a, int64_t n, real key) {
  int64_t first = 0, count = n;
  while (count > 0) {
    int64_t step = count / 2;
    int64_t pos = first + step;
    if (a[pos] < key) { first = pos + 1; count -= step + 1; }
    else count = step;
  }
  return first;
}

/* std/math.almostEqual(x, y, unitsInLastPlace = 4), used at raytracer.nim:2055. */
static int almost_equal_r(real x, real y) {
  if (x == y) return 1;
  real diff = M_FABS(x - y);
  return (diff <= 2.220446049250313e-16 * M_FABS(x + y) * 4.0) || (diff < 2.2250738585072014e-308);
}

/* numericalnim newBilinearSpline / eval (call sites raytracer.nim:1181, 1204, 1226, 1567-1568,
 * 1577-1578): uniform grid given by (min,max) and the tensor shape; z[i][j], i <-> x. */
static real bilinear_r(const double* z, int32_t nx, int32_t ny, real xmin, real xmax,
                            real ymin, real ymax, real x, real y) {
  real dx = (xmax - xmin) / (real)(nx - 1);
  real dy = (ymax - ymin) / (real)(ny - 1);
  long i = (long)M_FLOOR((x - xmin) / dx);
  long j = (long)M_FLOOR((y - ymin) / dy);
  if (i > nx - 2) i = nx - 2;
  if (j > ny - 2) j = ny - 2;
  /* Below the grid the library indexes out of bounds (IndexDefect, or garbage with -d:danger) - the
   * BabyIAXO test source's 0.021 keV (raytracer.nim:1377) is such a point.  Undefined in the
   * reference; defined here (and in the HIP path) as extrapolation from the first cell. */
  if (i < 0) i = 0;
  if (j < 0) j = 0;
  real xCorner = xmin + (real)i * dx;
  real yCorner = ymin + (real)j * dy;
  real xUnit = (x - xCorner) / dx;
  real yUnit = (y - yCorner) / dy;
  real f00 = z[(size_t)i * ny + j];
  real f10 = z[(size_t)(i + 1) * ny + j];
  real f01 = z[(size_t)i * ny + j + 1];
  real f11 = z[(size_t)(i + 1) * ny + j + 1];
  real a00 = f00;
  real a10 = f10 - f00;
  real a01 = f01 - f00;
  real a11 = f11 - f10 - f01 + f00;
  return a00 + a10 * xUnit + a01 * yUnit + a11 * xUnit * yUnit;
}

/* numericalnim newLinear1D / eval (raytracer.nim:1522-1527, 2170, 2179, 2190): binary search
 * for the interval, y0 + (x - x0) * (y1 - y0) / (x1 - x0).  The library raises outside
 * [x0, xN]; the path never leaves the tables (0..15 keV), so the ends clamp to the end
 * intervals here. */
static real linear1d_r(const double* xs, const double* ys, int32_t n, real x) {
  /* n_int = last index with xs[idx] <= x, clamped to [0, n-2] */
  int64_t lo = 0, hi = n - 1;
  while (hi - lo > 1) {
    int64_t mid = (lo + hi) / 2;
    if (xs[mid] <= x) lo = mid; else hi = mid;
  }
  real x0 = xs[lo], x1 = xs[lo + 1], y0 = ys[lo], y1 = ys[lo + 1];
  return y0 + (x - x0) * (y1 - y0) / (x1 - x0);
}

/* ------------------------------------------------------------------------------------------
 * rotations, raytracer.nim:334-361
 * ---------------------------------------------------------------------------------------- */
static inline v3 rotate_in_x(v3 v, real angle, real off) { /* :338-345 */
  v.z += -off;
  v3 r = V(v.x * M_COS(angle) + v.z * M_SIN(angle), v.y, v.z * M_COS(angle) - v.x * M_SIN(angle));
  r.z += off;
  return r;
}
static inline v3 rotate_in_y(v3 v, real angle, real off) { /* :347-354 */
  v.z += -off;
  v3 r = V(v.x, v.y * M_COS(angle) - v.z * M_SIN(angle), v.z * M_COS(angle) + v.y * M_SIN(angle));
  r.z += off;
  return r;
}
static inline v3 rotate_around_z(v3 v, real angle) { /* :356-361 */
  return V(v.x * M_COS(angle) + v.y * M_SIN(angle), v.y * M_COS(angle) - v.x * M_SIN(angle), v.z);
}

/* conversionProb, raytracer.nim:363-365, with unchained's natural-unit conversions
 * (T -> eV^2: 195.353; m -> eV^-1: 1/1.97327e-7; GeV^-1 -> eV^-1: 1e-9). */
static real conversion_prob_r(real B_tesla, real g_agamma_inv_gev, real length_mm) {
  real L_m = length_mm * 1e-3;
  real B_nat = B_tesla * 195.353;
  real L_nat = L_m / 1.97327e-7;
  real g_nat = g_agamma_inv_gev * 1e-9;
  return M_POW(g_nat * B_nat * L_nat / 2.0, 2.0);
}

/* ------------------------------------------------------------------------------------------
 * gas phase, axionMassforMagnet.nim:4-113
 * ---------------------------------------------------------------------------------------- */
static real gas_density(real p, real temp) { /* :4-15 and :17-28 (identical bodies) */
  const real gasConstant = 8.314, M = 4.002602;
  real pressure = p * 1e2;
  real result = pressure * M / (gasConstant * temp * 1000.0);
  return result / 1000.0;
}
static real num_density(real c) { return 2 * 6.022e23 * c; } /* :30-33 */
static real eff_photon_mass(real ne) { /* :35-40 */
  const real alpha = 1.0 / 137.0, me = 511e3;
  return M_SQRT(M_POW(1.97e-7, 3) * 4 * SART_PI * alpha * ne / me);
}
static real molar_amount(real p, real vol, real temp) { /* :42-49 */
  const real gasConstant = 8.314;
  real pressure = p * 1e2;
  return pressure * vol / (gasConstant * temp);
}
static real eff_photon_mass2_r(real p, real length, real radBore, real temp) { /* :51-61 */
  real vol = length * (SART_PI * M_POW(radBore, 2));
  real amountMol = molar_amount(p, vol, temp);
  real numPerMol = num_density(amountMol / vol);
  return eff_photon_mass(numPerMol);
}
static real momentum_transfer(real m_gamma, real m_a, real E_a_kev) { /* :63-68 */
  return M_FABS((m_gamma * m_gamma - m_a * m_a) / (2 * (E_a_kev * 1000.0)));
}
static real log_mass_attenuation(real e) { /* :70-73 */
  return -1.5832 + 5.9195 * M_EXP(-0.353808 * e) + 4.03598 * M_EXP(-0.970557 * e);
}
static real axion_conversion_prob2_r(real m_a, real energyAx, real pressure, real temp,
                                          real length, real radBore, real g_agamma,
                                          real B) { /* :75-101 */
  real gamma = 1.97e-7 * 100.0 * gas_density(pressure, temp) * M_EXP(log_mass_attenuation(energyAx));
  real m_gamma = eff_photon_mass2_r(pressure, length, radBore, temp);
  real L = length / 1.97e-7;
  real g_agammaEV = g_agamma * 1e-9;
  real beV = B * 1e3 / 1.444;
  real q = momentum_transfer(m_gamma, m_a, energyAx);
  real term1 = M_POW(g_agammaEV * beV / 2.0, 2);
  real term2 = 1.0 / (q * q + gamma * gamma / 4);
  real term3 = 1.0 + M_EXP(-gamma * L) - 2 * M_EXP(-gamma * L / 2) * M_COS(q * L);
  return term1 * term2 * term3;
}
static real intensity_suppression2_r(real energy, real distanceMagnet, real distancePipe,
                                          real pressure, real tempMagnet,
                                          real tempPipe) { /* :103-113 */
  real massAtt = M_EXP(log_mass_attenuation(energy));
  real rhoMagnet = gas_density(pressure, tempMagnet);
  real rhoPipe = gas_density(pressure, tempPipe);
  return M_EXP(-massAtt * rhoPipe * distancePipe * 100) * M_EXP(-massAtt * rhoMagnet * distanceMagnet * 100);
}

/* ------------------------------------------------------------------------------------------
 * sampling, raytracer.nim:412-471
 * ---------------------------------------------------------------------------------------- */
static v3 get_random_point_on_disk(v3 center, real radius, real u_r, real u_phi) { /* :412-422 */
  real r = radius * M_SQRT(u_r);
  real angle = 360 * u_phi;
  real x = M_COS(deg_to_rad(angle)) * r;
  real y = M_SIN(deg_to_rad(angle)) * r;
  return vadd(V(x, y, 0.0), center);
}

static v3 get_random_point_from_solar_model(v3 center, real radius, const double* cdf, int32_t n,
                                            real u_a1, real u_a2, real u_r) { /* :425-442 */
  real angle1 = 360 * u_a1;
  real angle2 = 180 * u_a2; /* QUIRK: uniform in theta, not in M_COS(theta) */
  int64_t rIdx = lower_bound_r(cdf, n, u_r);
  real r = (0.0015 + (real)rIdx * 0.0005) * radius;
  real x = M_COS(deg_to_rad(angle1)) * M_SIN(deg_to_rad(angle2)) * r;
  real y = M_SIN(deg_to_rad(angle1)) * M_SIN(deg_to_rad(angle2)) * r;
  real z = M_COS(deg_to_rad(angle2)) * r;
  return vadd(V(x, y, z), center);
}

static real get_random_energy_from_solar_model(v3 vectorInSun, v3 center, real radius,
                                                 const sart_oracle_tables_t* t, real u, int* idx_out) { /* :444-471 */
  real rad = vlength(vsub(vectorInSun, center));
  real r = rad / radius;
  real indexRad = (r - 0.0015) / 0.0005;
  int64_t iRad;
  if (indexRad - 0.5 > M_FLOOR(indexRad)) iRad = (int64_t)M_CEIL(indexRad);
  else iRad = (int64_t)M_FLOOR(indexRad);
  /* The reference would raise an IndexDefect outside the table; never reached because
   * r comes from the same grid (:438).  Clamp so the oracle cannot read out of bounds. */
  if (iRad < 0) iRad = 0;
  if (iRad > t->n_radii - 1) iRad = t->n_radii - 1;
  const double* cdfEmRate = t->diff_flux_cdfs + (size_t)iRad * (size_t)t->n_energies;
  int64_t idx = lower_bound_r(cdfEmRate, t->n_energies, u);
  if (idx > t->n_energies - 1) idx = t->n_energies - 1; /* CDF ends at exactly 1.0 (:2675-2677) */
  real energy = t->energies_kev[idx];
  *idx_out = (int)idx;
  return M_FMAX(0.03, energy); /* QUIRK: discrete energies, clamped to >= 0.03 keV (:470-471) */
}

/* ------------------------------------------------------------------------------------------
 * line / aperture intersections, raytracer.nim:481-616
 * ---------------------------------------------------------------------------------------- */
static int line_intersects_circle(v3 p1, v3 p2, v3 center, real radius) { /* :481-492 */
  v3 vector = vsub(p2, p1);
  real lambda1 = (center.z - p1.z) / vector.z;
  v3 intersect = vsub(vadd(p1, vscale(lambda1, vector)), center);
  real r_xy = M_SQRT(intersect.x * intersect.x + intersect.y * intersect.y);
  return r_xy < radius;
}

static int line_intersects_object(int kind, v3 p1, v3 p2, v3 center, real radius) { /* :494-527 */
  v3 vector = vsub(p2, p1);
  real lambda1 = (center.z - p1.z) / vector.z;
  v3 intersect = vsub(vadd(p1, vscale(lambda1, vector)), center);
  real r_xy = M_SQRT(intersect.x * intersect.x + intersect.y * intersect.y);
  real tx = intersect.x / M_SQRT(2.0) - intersect.y / M_SQRT(2.0);
  real ty = intersect.x / M_SQRT(2.0) + intersect.y / M_SQRT(2.0);
  real ax = M_FABS(intersect.x), ay = M_FABS(intersect.y), atx = M_FABS(tx), aty = M_FABS(ty);
  switch (kind) {
    case SART_HT_CIRCLE: return r_xy < radius;
    case SART_HT_CROSS:
      return (ax < radius && ay < radius * 16.0) || (ay < radius && ax < radius * 16.0);
    case SART_HT_STAR:
      return (ax < radius && ay < radius * 16.0) || (ay < radius && ax < radius * 16.0) ||
             (atx < radius && aty < radius * 16.0) || (aty < radius && atx < radius * 16.0);
    case SART_HT_SQUARE: return ax < radius && ay < radius;
    case SART_HT_DIAMOND: return atx < radius && aty < radius;
    default: return 0; /* htNone */
  }
}

static v3 get_intersect_line_intersects_circle(v3 p1, v3 p2, v3 center) { /* :529-534 */
  v3 vector = vsub(p2, p1);
  real lambda1 = (center.z - p1.z) / vector.z;
  return vadd(p1, vscale(lambda1, vector));
}

typedef struct { v3 i1, i2; int v1, v2; } cyl_result;

static cyl_result line_intersects_cylinder(v3 point_1, v3 point_2, v3 cb, v3 ce, real radius) { /* :538-588 */
  real alpha_x = M_ASIN((ce.x - cb.x) / M_FABS(cb.z - ce.z));
  real alpha_y = M_ASIN((ce.y - cb.y) / M_FABS(cb.z - ce.z));
  real offset_x, offset_y;
  if (M_FABS(ce.x) <= M_FABS(cb.x)) offset_x = ce.x; else offset_x = cb.x;
  if (M_FABS(ce.y) <= M_FABS(cb.y)) offset_y = ce.y; else offset_y = cb.y;
  v3 off = V(offset_x, offset_y, 0.0);
  v3 p_1 = vsub(rotate_in_y(rotate_in_x(point_1, alpha_x, 0.0), alpha_y, 0.0), off);
  v3 p_2 = vsub(rotate_in_y(rotate_in_x(point_2, alpha_x, 0.0), alpha_y, 0.0), off);
  v3 vector = vsub(p_2, p_1);
  real lambda_dummy = (-1000.0 - p_1.z) / vector.z;
  v3 dummy = vadd(p_1, vscale(lambda_dummy, vector));
  v3 vector_dummy = vsub(p_2, dummy);
  real factor = vector_dummy.x * vector_dummy.x + vector_dummy.y * vector_dummy.y;
  real p = 2.0 * (dummy.x * vector_dummy.x + dummy.y * vector_dummy.y) / factor;
  real q = (dummy.x * dummy.x + dummy.y * dummy.y - (radius * radius)) / factor;
  real lambda_1 = -p / 2.0 + M_SQRT(p * p / 4.0 - q);
  real lambda_2 = -p / 2.0 - M_SQRT(p * p / 4.0 - q);
  v3 intersect_1 = vadd(dummy, vscale(lambda_1, vector_dummy));
  v3 intersect_2 = vadd(dummy, vscale(lambda_2, vector_dummy));
  intersect_1 = vadd(intersect_1, off);
  intersect_1 = rotate_in_y(rotate_in_x(intersect_1, -alpha_x, 0.0), -alpha_y, 0.0);
  intersect_2 = vadd(intersect_2, off);
  intersect_2 = rotate_in_y(rotate_in_x(intersect_2, -alpha_x, 0.0), -alpha_y, 0.0);
  cyl_result r;
  r.i1 = intersect_1; r.i2 = intersect_2;
  r.v1 = (intersect_1.z > cb.z) && (intersect_1.z < ce.z);
  r.v2 = (intersect_2.z > cb.z) && (intersect_2.z < ce.z);
  return r;
}

static int line_intersects_cylinder_once(v3 p1, v3 p2, v3 cb, v3 ce, real radius) { /* :591-604 */
  cyl_result r = line_intersects_cylinder(p1, p2, cb, ce, radius);
  if ((r.v1 && r.v2) || (!r.v1 && !r.v2)) return 0;
  return 1;
}

static v3 get_intersect_line_intersects_cylinder_once(v3 p1, v3 p2, v3 cb, v3 ce, real radius) { /* :608-616 */
  cyl_result r = line_intersects_cylinder(p1, p2, cb, ce, radius);
  return r.v1 ? r.i1 : r.i2;
}

static v3 get_pixel_value(v3 intersects) { /* :618-623 */
  const real sizeViewfield = 48.0;
  v3 pix = V(0.0, 0.0, 0.0);
  pix.x = M_FLOOR(intersects.x / (sizeViewfield / 1400.0)) + 700;
  pix.y = M_FLOOR(intersects.y / (sizeViewfield / 1400.0)) + 700;
  return pix;
}

/* ------------------------------------------------------------------------------------------
 * mirrors, raytracer.nim:628-795
 * ---------------------------------------------------------------------------------------- */
static inline real cot_(real x) { return 1.0 / M_TAN(x); } /* std/math.cot */

static v3 pick_root(v3 point, v3 direc, real a, real half_b, real c, real distMirr,
                    real lMirror, real angle) { /* common tail :646-658 / :678-690 / :717-729 */
  real s;
  real root1 = (-half_b - M_SQRT(half_b * half_b - a * c)) / a;
  real root2 = (-half_b + M_SQRT(half_b * half_b - a * c)) / a;
  if (point.z + root1 * direc.z > distMirr && point.z + root1 * direc.z < distMirr + lMirror * M_COS(angle))
    s = root1;
  else if (point.z + root2 * direc.z > distMirr && point.z + root2 * direc.z < distMirr + lMirror * M_COS(angle))
    s = root2;
  else
    s = 0.0; /* miss (also when the discriminant is negative: NaN compares false) */
  return vadd(point, vscale(s, direc));
}

static v3 find_pos_cone(v3 pointXRT, v3 pointCB, real r1, real angle, real lMirror,
                        real distMirr) { /* :628-658 */
  v3 point = pointCB, direc = vsub(pointXRT, pointCB);
  real k = M_TAN(angle) * M_TAN(angle);
  real a = direc.x * direc.x + direc.y * direc.y - k * direc.z * direc.z;
  real b = 2.0 * (point.x * direc.x + point.y * direc.y + r1 * M_TAN(angle) * direc.z -
                    k * (point.z - distMirr) * direc.z);
  real half_b = b / 2.0;
  real c = point.x * point.x + point.y * point.y - r1 * r1 +
             2.0 * r1 * M_TAN(angle) * (point.z - distMirr) -
             k * (point.z - distMirr) * (point.z - distMirr);
  return pick_root(point, direc, a, half_b, c, distMirr, lMirror, angle);
}

static v3 find_pos_parabolic(v3 pointXRT, v3 pointCB, real r1, real angle, real lMirror,
                             real distMirr) { /* :660-690 */
  v3 point = pointCB, direc = vsub(pointXRT, pointCB);
  real r3 = -M_TAN(angle) * lMirror + M_SQRT(M_TAN(angle) * lMirror * M_TAN(angle) * lMirror + r1 * r1);
  real e = 2.0 * r3 * M_TAN(angle);
  real a = direc.x * direc.x + direc.y * direc.y; /* QUIRK: a == 0 for an axis-parallel ray => miss */
  real b = 2.0 * (point.x * direc.x + point.y * direc.y) + e * direc.z;
  real half_b = b / 2.0;
  real c = point.x * point.x + point.y * point.y - r3 * r3 - e * lMirror + e * point.z;
  return pick_root(point, direc, a, half_b, c, distMirr, lMirror, angle);
}

static v3 find_pos_hyperbolic(v3 pointXRT, v3 pointCB, real r1, real angle, real lMirror,
                              real distMirr, real focalLength) { /* :692-729; angle = 3*beta */
  v3 point = pointCB, direc = vsub(pointXRT, pointCB);
  real r3 = -M_TAN(angle / 3.0) * lMirror +
              M_SQRT(M_TAN(angle / 3.0) * lMirror * M_TAN(angle / 3.0) * lMirror + r1 * r1);
  real f = focalLength;
  real e = 2.0 * r3 * M_TAN(angle);
  real g = 2.0 * r3 * M_TAN(angle) / (f + r3 * cot_(2.0 * angle / 3.0));
  real a = direc.x * direc.x + direc.y * direc.y - g * direc.z * direc.z;
  real b = 2.0 * (point.x * direc.x + point.y * direc.y + g * direc.z * lMirror -
                    g * direc.z * point.z) + e * direc.z;
  real half_b = b / 2.0;
  real c = point.x * point.x + point.y * point.y - r3 * r3 - e * lMirror + e * point.z -
             g * lMirror * lMirror + 2.0 * g * point.z * lMirror - g * point.z * point.z;
  return pick_root(point, direc, a, half_b, c, distMirr, lMirror, angle);
}

static v3 calc_normal_vec(v3 pm, real angle, real r1, real lMirror, real focalLength,
                          int shape) { /* :731-759 */
  v3 n = V(pm.x, pm.y, 0.0);
  if (shape == 0) {
    n.z = M_TAN(angle) * M_SQRT(pm.x * pm.x + pm.y * pm.y);
  } else if (shape == 1) {
    real r3 = -M_TAN(angle) * lMirror + M_SQRT(M_TAN(angle) * lMirror * M_TAN(angle) * lMirror + r1 * r1);
    real m = 1.0 / (r3 * M_TAN(angle) / M_SQRT(r3 * r3 + r3 * 2.0 * M_TAN(angle) * (lMirror - pm.z)));
    real nn = M_SQRT(pm.x * pm.x + pm.y * pm.y) - m * pm.z;
    n.z = pm.z - (-nn / m);
  } else {
    real r3 = -M_TAN(angle / 3.0) * lMirror +
                M_SQRT(M_TAN(angle / 3.0) * lMirror * M_TAN(angle / 3.0) * lMirror + r1 * r1);
    real f = focalLength;
    real alpha = angle / 3.0;
    real z = pm.z;
    real m = 1.0 /
               (r3 * M_TAN(angle) * (1.0 + 2.0 * (lMirror - z) / (f + r3 * cot_(2.0 * alpha))) /
                M_SQRT(r3 * r3 + r3 * 2.0 * M_TAN(angle) * (lMirror - z) *
                                   (1.0 + (lMirror - z) / (f + r3 * cot_(2.0 * alpha)))));
    real nn = M_SQRT(pm.x * pm.x + pm.y * pm.y) - m * z;
    n.z = pm.z - (-nn / m);
  }
  return n;
}

static v3 get_vector_after_mirror(v3 pointXRT, v3 pointCB, v3 pointMirror, real angle, real r1,
                                  real lMirror, real focalLength, int shape) { /* :762-780 */
  v3 normalVec = calc_normal_vec(pointMirror, angle, r1, lMirror, focalLength, shape);
  v3 vectorBeforeMirror = vnormalize(vsub(pointXRT, pointCB));
  v3 vectorAxis = vnormalize(vcross(normalVec, vectorBeforeMirror));
  real alphaMirror = M_ASIN(M_FABS(vdot(normalVec, vectorBeforeMirror) / vlength(normalVec)));
  v3 vecBeforeAxis = vcross(vectorBeforeMirror, vectorAxis);
  return vsub(vscale(M_COS(2.0 * alphaMirror), vectorBeforeMirror),
              vscale(M_SIN(2.0 * alphaMirror), vecBeforeAxis));
}

static real get_mirror_angle(v3 pointXRT, v3 pointCB, v3 pointMirror, real angle, real r1,
                               real lMirror, real focalLength, int shape) { /* :782-795 */
  v3 normalVec = calc_normal_vec(pointMirror, angle, r1, lMirror, focalLength, shape);
  v3 vectorBeforeMirror = vnormalize(vsub(pointXRT, pointCB));
  real alphaMirror = M_ASIN(M_FABS(vdot(normalVec, vectorBeforeMirror) / vlength(normalVec)));
  return rad_to_deg(alphaMirror);
}

static v3 get_point_detector_window(v3 pointMirror2, v3 pointAfterMirror2, real distDet,
                                    real dCBXray, real pipeAngleDeg) { /* :797-814 */
  real pipeRad = deg_to_rad(pipeAngleDeg);
  v3 shift = V(dCBXray, 0.0, 0.0);
  v3 pointMirror2Turned = vsub(rotate_in_x(pointMirror2, pipeRad, 0.0), shift);
  v3 pointAfterMirror2Turned = vsub(rotate_in_x(pointAfterMirror2, pipeRad, 0.0), shift);
  v3 vectorAfterMirror2 = vsub(pointAfterMirror2Turned, pointMirror2Turned);
  real dd = distDet / M_COS(pipeRad);
  real n = (dd - pointMirror2Turned.z) / vectorAfterMirror2.z;
  return vadd(pointMirror2Turned, vscale(n, vectorAfterMirror2));
}

/* exported wrappers for known-answer tests */
void sart_oracle_find_pos(int shape, const double pxrt[3], const double pcb[3], double r1,
                          double angle, double lMirror, double distMirr, double focal, double out[3]) {
  v3 a = V(pxrt[0], pxrt[1], pxrt[2]), b = V(pcb[0], pcb[1], pcb[2]), r;
  if (shape == 0) r = find_pos_cone(a, b, r1, angle, lMirror, distMirr);
  else if (shape == 1) r = find_pos_parabolic(a, b, r1, angle, lMirror, distMirr);
  else r = find_pos_hyperbolic(a, b, r1, angle, lMirror, distMirr, focal);
  out[0] = (double)r.x; out[1] = (double)r.y; out[2] = (double)r.z;
}
void sart_oracle_vector_after_mirror(const double pxrt[3], const double pcb[3], const double pm[3],
                                     double angle, double r1, double lMirror, double focal,
                                     int shape, double out[3]) {
  v3 r = get_vector_after_mirror(V(pxrt[0], pxrt[1], pxrt[2]), V(pcb[0], pcb[1], pcb[2]),
                                 V(pm[0], pm[1], pm[2]), angle, r1, lMirror, focal, shape);
  out[0] = (double)r.x; out[1] = (double)r.y; out[2] = (double)r.z;
}
double sart_oracle_mirror_angle_deg(const double pxrt[3], const double pcb[3], const double pm[3],
                                    double angle, double r1, double lMirror, double focal, int shape) {
  return (double)get_mirror_angle(V(pxrt[0], pxrt[1], pxrt[2]), V(pcb[0], pcb[1], pcb[2]),
                          V(pm[0], pm[1], pm[2]), angle, r1, lMirror, focal, shape);
}

/* lengthTelescope, raytracer.nim:1883-1884 */
static real length_telescope_r(const sart_setup_t* s) {
  real a0 = deg_to_rad(s->all_angles_deg[0]);
  return (s->l_mirror + 0.5 * s->all_xsep[0]) * M_COS(a0) +
         (s->l_mirror + 0.5 * s->all_xsep[0]) * M_COS(3.0 * a0);
}

/* ------------------------------------------------------------------------------------------
 * weights, raytracer.nim:1533-1625
 * ---------------------------------------------------------------------------------------- */
static void compute_reflectivity(const sart_setup_t* s, const sart_oracle_tables_t* t, real energy,
                                 int64_t hitLayer, real transmissionMagnet, real p, real ya,
                                 real alpha1, real alpha2, uint32_t flags, real* reflect,
                                 real* weight) { /* :1533-1580 */
  (void)p; (void)ya;
  if (flags & SART_CF_IGNORE_REFLECTION) {
    *reflect = 1.0;
    *weight = transmissionMagnet;
    return;
  }
  size_t plane = (size_t)t->refl_n_angles * (size_t)t->refl_n_energies;
  if (s->reflectivity_kind == SART_RK_SINGLE_COATING) { /* :1563-1570 */
    real p1 = bilinear_r(t->refl_data, t->refl_n_angles, t->refl_n_energies, t->refl_angle_min,
                                     t->refl_angle_max, t->refl_energy_min, t->refl_energy_max, alpha1, energy);
    real p2 = bilinear_r(t->refl_data, t->refl_n_angles, t->refl_n_energies, t->refl_angle_min,
                                     t->refl_angle_max, t->refl_energy_min, t->refl_energy_max, alpha2, energy);
    *reflect = p1 * p2;
    *weight = *reflect * transmissionMagnet;
  } else { /* rkMultiCoating :1571-1580 */
    /* QUIRK: layers.lowerBound(hitLayer) on [2,5,9,14] maps shells 0-2 -> 0, 3-5 -> 1,
     * 6-9 -> 2, 10-13 -> 3 (off by one against the comment at :78-80). */
    int64_t layerIdx = 0;
    {
      int64_t first = 0, count = s->n_coatings;
      while (count > 0) {
        int64_t step = count / 2, pos = first + step;
        if ((int64_t)s->coating_layers[pos] < hitLayer) { first = pos + 1; count -= step + 1; }
        else count = step;
      }
      layerIdx = first;
    }
    if (layerIdx > t->refl_n_coatings - 1) layerIdx = t->refl_n_coatings - 1; /* IndexDefect in the reference */
    const double* z = t->refl_data + (size_t)layerIdx * plane;
    real p1 = bilinear_r(z, t->refl_n_angles, t->refl_n_energies, t->refl_angle_min,
                                     t->refl_angle_max, t->refl_energy_min, t->refl_energy_max, alpha1, energy);
    real p2 = bilinear_r(z, t->refl_n_angles, t->refl_n_energies, t->refl_angle_min,
                                     t->refl_angle_max, t->refl_energy_min, t->refl_energy_max, alpha2, energy);
    *reflect = p1 * p2;
    *weight = *reflect * transmissionMagnet;
  }
}

static real compute_magnet_transmission(const sart_setup_t* s, real energy, real distancePipe_m,
                                          real pathCB_mm, real ya, uint32_t flags) { /* :1582-1625 */
  if (s->stage == SART_SK_VACUUM) {
    real prob = (flags & SART_CF_IGNORE_CONV_PROB)
                      ? 1.0
                      : conversion_prob_r(s->magnet_B, s->g_agamma, pathCB_mm);
    return M_COS(ya) * prob; /* QUIRK: ya is in degrees, M_COS() takes it as radians (:1598) */
  }
  /* QUIRK: pGas is a bar quantity handed to functions documented in mbar (:1601-1621);
   * pathCB is used as the gas-column length. */
  real pGas = s->magnet_pGasRoom / s->room_temp * s->magnet_tGas;
  real pathCB_m = pathCB_mm * 1e-3;
  real radiusCB_m = s->magnet_radiusCB * 1e-3;
  real prob = (flags & SART_CF_IGNORE_CONV_PROB)
                    ? 1.0
                    : axion_conversion_prob2_r(s->m_axion, energy, pGas, s->magnet_tGas, pathCB_m,
                                                         radiusCB_m, s->g_agamma, s->magnet_B);
  real absorb = intensity_suppression2_r(energy, pathCB_m, distancePipe_m, pGas, s->magnet_tGas,
                                                     s->room_temp);
  return M_COS(ya) * prob * absorb;
}

static void radius_and_phi(v3 v, real* radius, real* phi) { /* :1627-1633 */
  *radius = M_SQRT(v.x * v.x + v.y * v.y);
  *phi = rad_to_deg(M_ACOS(v.x / *radius));
}

static int line_intersects_opaque_telescope_structures(const sart_setup_t* s, real radialDist,
                                                       v3 testVector, v3 vectorXRT, v3 pointExitCB,
                                                       v3 pointEntranceXRT) { /* :1635-1704 */
  int result = 0;
  switch (s->telescope_kind) {
    case SART_TK_LLNL:
      /* QUIRK: `return` without a value => false: the graphite block never blocks (:1646) */
      return 0;
    case SART_TK_ABRIXAS: {
      real factorSpider = (-35.0 - pointExitCB.z) / vectorXRT.z;
      v3 pointEntranceSpider = vadd(pointExitCB, vscale(factorSpider, vectorXRT));
      real radius, phiFlat, radiusSpider, phiFlatSpider;
      radius_and_phi(pointEntranceXRT, &radius, &phiFlat);
      radius_and_phi(pointEntranceSpider, &radiusSpider, &phiFlatSpider);
      if (radialDist < 37.5) result = 1;
      else {
        for (int i = 0; i <= 6; ++i) {
          if ((phiFlat >= (-3.75 + 60.0 * i) && phiFlat <= (3.75 + 60.0 * i)) ||
              (phiFlatSpider >= (-3.75 + 60.0 * i) && phiFlatSpider <= (3.75 + 60.0 * i))) {
            result = 1;
            break;
          }
        }
      }
      return result;
    }
    case SART_TK_XMM: {
      real factorSpider = (-85.0 - pointExitCB.z) / vectorXRT.z;
      v3 pointEntranceSpider = vadd(pointExitCB, vscale(factorSpider, vectorXRT));
      real radius, phiFlat, radiusSpider, phiFlatSpider;
      radius_and_phi(pointEntranceXRT, &radius, &phiFlat);
      radius_and_phi(pointEntranceSpider, &radiusSpider, &phiFlatSpider);
      if (radialDist <= 64.7) {
        int nHoles = s->number_of_holes;
        int lim = nHoles - (int)M_CEIL((real)nHoles / 2.0);
        for (int l = -lim; l <= lim; ++l) {
          v3 centerHole = testVector;
          if (l != 0) {
            if (abs(l) % 2 == 0) centerHole.y += 2.0 * (real)l * s->hole_in_optics;
            else centerHole.x += 2.0 * ((real)l + ((real)l / (real)abs(l))) * s->hole_in_optics;
          }
          if (line_intersects_object(s->hole_type, pointExitCB, pointEntranceXRT, centerHole, s->hole_in_optics)) {
            result = 0;
            break;
          } else
            result = 1;
        }
      } else if (radialDist < 151.6 && radialDist > (151.6 - 20.9)) {
        result = 1;
      } else if (radialDist > 64.7) {
        for (int i = 0; i <= 16; ++i) {
          if ((phiFlat >= (-1.145 + 22.5 * i) && phiFlat <= (1.145 + 22.5 * i)) ||
              (phiFlatSpider >= (-1.145 + 22.5 * i) && phiFlatSpider <= (1.145 + 22.5 * i))) {
            result = 1;
            break;
          }
        }
      }
      return result;
    }
    default:
      return 0; /* the reference asserts (:1703); sart_set_setup rejects these kinds */
  }
}

static int line_hits_nickel(const sart_setup_t* s, real alpha1_deg, real r1, int64_t hitLayer,
                            v3 pointMirror1) { /* :1706-1734 */
  if (hitLayer > 0) {
    real tanAlpha = M_TAN(deg_to_rad(alpha1_deg));
    int64_t hL = hitLayer - 1;
    real compVal = (r1 - (s->all_r1[hL] + s->all_thickness[hL])) / (s->l_mirror - pointMirror1.z);
    return tanAlpha > compVal;
  }
  return 0;
}

/* ------------------------------------------------------------------------------------------
 * traceAxion, raytracer.nim:1736-2221
 * ---------------------------------------------------------------------------------------- */
/* `stage` (oracle-only bookkeeping for the SART_ACC_N_REACHED_TELESCOPE / _SHELL_SELECTED
 * counters): 1 once the ray has passed bore + pipes, 2 once a shell has been selected. */
static void trace_axion_impl(sart_axion_t* res, const sart_setup_t* s, const sart_oracle_tables_t* t,
                             uint32_t flags, const double u[6], int* stage, int* e_idx_out) {
  /* centre vectors, initCenterVectors :278-320 */
  const v3 c_sun = V(0.0, -(0.0 * 1.33e10), -s->distance_sun_earth);
  const v3 c_entranceCB = V(0.0, -0.0, 0.0);
  const v3 c_exitCBMagneticField = V(0.0, 0.0, s->magnet_lengthB);
  const v3 c_exitCB = V(0.0, -0.0, s->magnet_lengthColdbore);
  const v3 c_exitPipeCBVT3 = V(0.0, 0.0, s->magnet_lengthColdbore + s->pipe_cb_vt3_length);
  const v3 c_exitPipeVT3XRT =
      V(0.0, 0.0, s->magnet_lengthColdbore + s->pipe_cb_vt3_length + s->pipe_vt3_xrt_length);
  const v3 c_xraySource = V(s->test_off_axis_left, s->test_off_axis_up, -(s->test_distance));
  const v3 c_collimator = V(s->test_off_axis_left, s->test_off_axis_up, -(s->test_distance) + s->test_length_col);
  const real ChipCenterX = s->chip_x_max / 2.0, ChipCenterY = s->chip_y_max / 2.0;

  const int testXray = s->test_active; /* :1746 */
  v3 rayOrigin, pointExitCBMagneticField = V(0.0, 0.0, 0.0);
  real energyAx;
  if (!testXray) { /* :1751-1764; draw order u0..u5 */
    rayOrigin = get_random_point_from_solar_model(c_sun, s->radius_sun, t->flux_radius_cdf, t->n_radii, u[0], u[1], u[2]);
    pointExitCBMagneticField = get_random_point_on_disk(c_exitCBMagneticField, s->magnet_radiusCB, u[3], u[4]);
    energyAx = get_random_energy_from_solar_model(rayOrigin, c_sun, s->radius_sun, t, u[5], e_idx_out);
  } else { /* :1765-1806 */
    rayOrigin = get_random_point_on_disk(c_xraySource, s->test_radius, u[0], u[1]);
    energyAx = s->test_energy;
    if (s->test_parallel) { /* rand(0.5) = 0.5 * rand(1.0) */
      pointExitCBMagneticField.x = rayOrigin.x + (u[2] * 0.5) - 0.25;
      pointExitCBMagneticField.y = rayOrigin.y + (u[3] * 0.5) - 0.25;
      pointExitCBMagneticField.z = s->magnet_lengthB;
    } else {
      pointExitCBMagneticField = get_random_point_on_disk(c_exitCBMagneticField, s->magnet_radiusCB, u[2], u[3]);
    }
    if (!line_intersects_circle(rayOrigin, pointExitCBMagneticField, c_collimator, s->test_radius)) return; /* :1800 */
  }

  int intersectsEntranceCB =
      line_intersects_circle(rayOrigin, pointExitCBMagneticField, c_entranceCB, s->magnet_radiusCB); /* :1813 */
  int intersectsCB = 0;
  res->emratesPre = 1.0; /* :1818 */
  res->energiesPre = energyAx;
  if (!intersectsEntranceCB)
    intersectsCB = line_intersects_cylinder_once(rayOrigin, pointExitCBMagneticField, c_entranceCB, c_exitCB,
                                                 s->magnet_radiusCB); /* :1822 */
  if (!intersectsEntranceCB && !intersectsCB) return; /* :1825 */

  v3 intersect;
  if (!intersectsEntranceCB)
    intersect = get_intersect_line_intersects_cylinder_once(rayOrigin, pointExitCBMagneticField, c_entranceCB,
                                                            c_exitCB, s->magnet_radiusCB); /* :1829 */
  else
    intersect = get_intersect_line_intersects_circle(rayOrigin, pointExitCBMagneticField, c_entranceCB); /* :1836 */

  real pathCB = vlength(vsub(pointExitCBMagneticField, intersect)); /* :1843 */

  if (!line_intersects_circle(rayOrigin, pointExitCBMagneticField, c_exitCB, s->magnet_radiusCB)) return; /* :1846 */

  v3 d0 = vsub(pointExitCBMagneticField, rayOrigin);
  v3 pointExitCB = vadd(rayOrigin, vscale((c_exitCB.z - rayOrigin.z) / d0.z, d0)); /* :1850-1853 */

  if (!line_intersects_circle(pointExitCBMagneticField, pointExitCB, c_exitPipeCBVT3, s->pipe_cb_vt3_radius))
    return; /* :1856 */

  v3 d1 = vsub(pointExitCB, pointExitCBMagneticField);
  v3 pointExitPipeCBVT3 =
      vadd(pointExitCBMagneticField, vscale((c_exitPipeCBVT3.z - pointExitCBMagneticField.z) / d1.z, d1)); /* :1860-1863 */

  /* QUIRK: the VT3->XRT cut uses coldBoreToVT3.radius again (:1866-1867) */
  if (!line_intersects_circle(pointExitCB, pointExitPipeCBVT3, c_exitPipeVT3XRT, s->pipe_cb_vt3_radius)) return;

  v3 d2 = vsub(pointExitPipeCBVT3, pointExitCB);
  v3 pointExitPipeVT3XRT = vadd(pointExitCB, vscale((c_exitPipeVT3XRT.z - pointExitCB.z) / d2.z, d2)); /* :1870-1872 */

  v3 vectorBeforeXRT = vsub(pointExitPipeVT3XRT, pointExitCB); /* :1874 */
  *stage = 1;

  /* telescope frame, :1878-1899 */
  v3 vectorXRT;
  real turnedX = deg_to_rad(s->telescope_turned_x_deg);
  real turnedY = deg_to_rad(s->telescope_turned_y_deg);
  real lengthTelescope = length_telescope_r(s);
  v3 entranceXY = V(s->optics_entrance[0], s->optics_entrance[1], 0.0);

  pointExitCB.z -= c_exitPipeVT3XRT.z;
  pointExitCB = vsub(rotate_in_y(rotate_in_x(pointExitCB, turnedX, lengthTelescope / 2), turnedY, lengthTelescope / 2),
                     entranceXY);
  pointExitPipeVT3XRT.z -= c_exitPipeVT3XRT.z;
  pointExitPipeVT3XRT = vsub(
      rotate_in_y(rotate_in_x(pointExitPipeVT3XRT, turnedX, lengthTelescope / 2), turnedY, lengthTelescope / 2),
      entranceXY);
  vectorXRT = vsub(pointExitPipeVT3XRT, pointExitCB);
  real factor = (0.0 - pointExitCB.z) / vectorXRT.z;
  v3 pointEntranceXRT = vadd(pointExitCB, vscale(factor, vectorXRT));
  vectorBeforeXRT = vectorXRT;

  real radialDist, phi_unused;
  radius_and_phi(pointEntranceXRT, &radialDist, &phi_unused); /* :1905 */

  if (line_intersects_opaque_telescope_structures(s, radialDist, V(0.0, 0.0, 0.0), vectorXRT, pointExitCB,
                                                  pointEntranceXRT))
    return; /* :1910-1914 */

  /* shell selection, :1918-1957 */
  real minDist = INFINITY;
  real r1 = 0.0, r2 = 0.0, r3 = 0.0, r4 = 0.0, r5 = 0.0, beta = 0.0, xSep = 0.0;
  int64_t hitLayer = 0;
  const int nS = s->n_shells;
  if (radialDist > s->all_r1[nS - 1]) return; /* :1934 */
  for (int j = 0; j < nS; ++j) {
    if (radialDist > s->all_r1[j] && radialDist < (s->all_r1[j] + s->all_thickness[j])) return; /* :1942-1944 */
    real dist = s->all_r1[j] - radialDist;
    if (dist > 0.0 && dist < minDist) {
      minDist = dist;
      hitLayer = j;
      r1 = s->all_r1[j];
      beta = deg_to_rad(s->all_angles_deg[j]);
      xSep = s->all_xsep[j];
      r2 = r1 - s->l_mirror * M_SIN(beta);
      r3 = r2 - 0.5 * xSep * M_TAN(beta);
      r4 = r3 - 0.5 * xSep * M_TAN(3.0 * beta);
      r5 = r4 - s->l_mirror * M_SIN(3.0 * beta);
    }
  }
  (void)r5;
  *stage = 2;

  real beta3 = 3.0 * beta;
  real distanceMirrors = M_COS(beta) * (xSep + s->l_mirror); /* :1973 */
  v3 pointMirror1, vectorAfterMirror1, pointAfterMirror1, pointMirror2, vectorAfterMirrors, pointAfterMirror2;
  real alpha1, alpha2;
  const real lM = s->l_mirror, fL = s->distance_detector_xrt;
  if (s->telescope_kind == SART_TK_XMM || s->telescope_kind == SART_TK_ABRIXAS) { /* :1984-2010 */
    pointMirror1 = find_pos_parabolic(pointEntranceXRT, pointExitCB, r1, beta, lM, 0.0);
    vectorAfterMirror1 = get_vector_after_mirror(pointEntranceXRT, pointExitCB, pointMirror1, beta, r1, lM, fL, 1);
    pointAfterMirror1 = vadd(pointMirror1, vscale(200.0, vectorAfterMirror1));
    pointMirror2 = find_pos_hyperbolic(pointAfterMirror1, pointMirror1, r1, beta3, lM, distanceMirrors, fL);
    vectorAfterMirrors = get_vector_after_mirror(pointAfterMirror1, pointMirror1, pointMirror2, beta3, r1, lM, fL, 2);
    pointAfterMirror2 = vadd(pointMirror2, vscale(200.0, vectorAfterMirrors));
    alpha1 = get_mirror_angle(pointEntranceXRT, pointExitCB, pointMirror1, beta, r1, lM, fL, 1);
    alpha2 = get_mirror_angle(pointAfterMirror1, pointMirror1, pointMirror2, beta3, r1, lM, fL, 2);
  } else { /* :2011-2037 */
    pointMirror1 = find_pos_cone(pointEntranceXRT, pointExitCB, r1, beta, lM, 0.0);
    vectorAfterMirror1 = get_vector_after_mirror(pointEntranceXRT, pointExitCB, pointMirror1, beta, r1, lM, fL, 0);
    pointAfterMirror1 = vadd(pointMirror1, vscale(200.0, vectorAfterMirror1));
    pointMirror2 = find_pos_cone(pointAfterMirror1, pointMirror1, r4, beta3, lM, distanceMirrors);
    vectorAfterMirrors = get_vector_after_mirror(pointAfterMirror1, pointMirror1, pointMirror2, beta3, r1, lM, fL, 0);
    pointAfterMirror2 = vadd(pointMirror2, vscale(200.0, vectorAfterMirrors));
    alpha1 = get_mirror_angle(pointEntranceXRT, pointExitCB, pointMirror1, beta, r1, lM, fL, 0);
    alpha2 = get_mirror_angle(pointAfterMirror1, pointMirror1, pointMirror2, beta3, r1, lM, fL, 0);
  }

  /* QUIRK: nickel test before the no-hit test (:2040-2046 vs :2055) */
  res->hitNickel = (uint8_t)line_hits_nickel(s, alpha1, r1, hitLayer, pointMirror1);
  if (res->hitNickel) return;

  real z0 = pointExitCB.z, z1 = pointMirror1.z, z2 = pointMirror2.z;
  if (almost_equal_r(z1, z2) || almost_equal_r(z1, z0)) return; /* :2051-2057 */

  /* detector plane, :2064-2094. QUIRK: hard-coded shell index 8 in allXsep[8] */
  real distDet = distanceMirrors - 0.5 * s->all_xsep[8] * M_COS(beta) + s->distance_detector_xrt -
                   s->distance_window_focal_plane;
  real d = -s->optics_entrance[0];
  v3 pointDetectorWindow = get_point_detector_window(pointMirror2, pointAfterMirror2, distDet, d, s->pipes_turned_deg);
  v3 pointEndDetector =
      get_point_detector_window(pointMirror2, pointAfterMirror2, (distDet + s->depth_det), d, s->pipes_turned_deg);

  res->deviationDet = M_SQRT(M_POW((pointEndDetector.x - pointDetectorWindow.x), 2.0) +
                           M_POW((pointEndDetector.y - pointDetectorWindow.y), 2.0));

  v3 valuesPix = get_pixel_value(pointEntranceXRT);
  res->pointdataXBefore = pointEntranceXRT.x;
  res->pointdataYBefore = pointEntranceXRT.y;
  res->pixvalsX = valuesPix.x;
  res->pixvalsY = valuesPix.y;

  /* pitch / yaw, :2101-2116 */
  vectorBeforeXRT = vscale(-1.0, vectorBeforeXRT);
  real vecLength = vlength(vectorBeforeXRT);
  real polar1 = rad_to_deg(M_ACOS(vectorBeforeXRT.x / vecLength));
  real polar2 = rad_to_deg(M_ATAN2(vectorBeforeXRT.z, vectorBeforeXRT.y));
  real p = polar1 - 90.0;
  real ya = polar2 + 90.0;
  real distancePipe = (pointDetectorWindow.z - pointExitCB.z) * 1e-3; /* mm -> m */

  res->transmissionMagnet = compute_magnet_transmission(s, energyAx, distancePipe, pathCB, ya, flags); /* :2120 */
  res->yawAngles = ya;

  real weight = 1.0, reflect = 0.0;
  compute_reflectivity(s, t, energyAx, hitLayer, res->transmissionMagnet, p, ya, alpha1, alpha2, flags, &reflect,
                       &weight); /* :2126 */
  res->reflect = reflect;

  if (testXray && minDist > 100.0) { /* :2130-2132 */
    v3 dd = vsub(pointEntranceXRT, pointExitCB);
    real n = (distDet - pointExitCB.z) / dd.z;
    pointDetectorWindow = vadd(pointExitCB, vscale(n, dd));
  }
  pointDetectorWindow.x -= s->lateral_shift;
  pointDetectorWindow.y -= s->transversal_shift;
  if (weight != 0) res->passedTillWindow = 1; /* :2135 */

  /* QUIRK: window cut, else chip cut (:2139-2147) */
  if (!(flags & SART_CF_IGNORE_DET_WINDOW) &&
      M_SQRT(pointDetectorWindow.x * pointDetectorWindow.x + pointDetectorWindow.y * pointDetectorWindow.y) >
          s->radius_window) {
    return;
  } else {
    if (M_FABS(pointDetectorWindow.x) > ChipCenterX || M_FABS(pointDetectorWindow.y) > ChipCenterY) return;
  }

  v3 turned = rotate_around_z(pointDetectorWindow, s->theta_rad); /* :2149-2153 */
  real y = turned.y;

  /* window strips, :2162-2187 */
  real stripDist = s->strip_dist_window, stripWidth = s->strip_width_window;
  real transWindow = 0.0;
  int nHalf = (int)M_ROUND((real)s->number_of_strips / 2.0);
  for (int i = 0; i <= nHalf - 1; ++i) {
    if (M_FABS(y) > (1.0 * i + 0.5) * stripDist + i * stripWidth &&
        M_FABS(y) < (1.0 * i + 0.5) * stripDist + (i + 1.0) * stripWidth) {
      transWindow = linear1d_r(t->strongback_x, t->strongback_y, t->n_strongback, energyAx);
      res->transProbWindow = transWindow;
      res->transProbDetector = transWindow;
      res->energiesAxAll = energyAx;
      res->energiesAxWindow = energyAx;
      res->kinds = SART_MK_SI;
      res->kindsWindow = SART_MK_SI;
      break;
    } else {
      transWindow = linear1d_r(t->window_x, t->window_y, t->n_window, energyAx);
      res->transProbWindow = transWindow;
      res->transProbDetector = transWindow;
      res->energiesAxAll = energyAx;
      res->energiesAxWindow = energyAx;
      res->kinds = SART_MK_SI3N4;
      res->kindsWindow = SART_MK_SI3N4;
    }
  }
  if (!(flags & SART_CF_IGNORE_DET_WINDOW)) weight *= transWindow;

  real absGasDet = linear1d_r(t->gas_abs_x, t->gas_abs_y, t->n_gas_abs, energyAx); /* :2190 */
  if (!(flags & SART_CF_IGNORE_GAS_ABS)) weight *= absGasDet;
  res->transProbArgon = absGasDet;
  res->transProbDetector = absGasDet;
  res->energiesAxAll = energyAx;
  res->kinds = SART_MK_AR;
  res->energiesAx = energyAx;
  res->shellNumber = hitLayer;

  real pointRadialComponent =
      M_SQRT(pointDetectorWindow.x * pointDetectorWindow.x + pointDetectorWindow.y * pointDetectorWindow.y);
  res->pointdataR = pointRadialComponent;
  pointDetectorWindow.x = -pointDetectorWindow.x + ChipCenterX; /* :2203-2204 */
  pointDetectorWindow.y = pointDetectorWindow.y + ChipCenterY;

  if (!(flags & SART_CF_XRAY_TEST)) { /* :2207-2212 */
    if (s->experiment == SART_ES_CAST) weight *= 3.585e3 * 3600.0 * 1.5 * 90.0;
    else weight *= 9.5e6 * 3600.0 * 12.0 * 90.0;
  }
  res->pointdataX = pointDetectorWindow.x;
  res->pointdataY = pointDetectorWindow.y;
  res->weights = weight;
  res->weightsAll = weight;
  if (weight != 0) res->passed = 1; /* :2220 */
}

void sart_oracle_trace_axion(sart_axion_t* res, const sart_setup_t* s, const sart_oracle_tables_t* t,
                             uint32_t flags, const double u[6]) {
  int stage = 0, e_idx = 0;
  trace_axion_impl(res, s, t, flags, u, &stage, &e_idx);
}

/* ------------------------------------------------------------------------------------------
 * exported f64 wrappers of the pieces above (known-answer tests)
 * ---------------------------------------------------------------------------------------- */
int64_t sart_oracle_lower_bound(const double* a, int64_t n, double key) { return lower_bound_r(a, n, key); }
int sart_oracle_almost_equal(double x, double y) { return almost_equal_r(x, y); }
double sart_oracle_bilinear(const double* z, int32_t nx, int32_t ny, double xmin, double xmax, double ymin,
                            double ymax, double x, double y) {
  return (double)bilinear_r(z, nx, ny, xmin, xmax, ymin, ymax, x, y);
}
double sart_oracle_linear1d(const double* xs, const double* ys, int32_t n, double x) {
  return (double)linear1d_r(xs, ys, n, x);
}
double sart_oracle_conversion_prob(double B, double g, double length_mm) { return (double)conversion_prob_r(B, g, length_mm); }
double sart_oracle_eff_photon_mass2(double p, double length, double radBore, double temp) {
  return (double)eff_photon_mass2_r(p, length, radBore, temp);
}
double sart_oracle_axion_conversion_prob2(double m_a, double e, double p, double temp, double length, double radBore,
                                          double g, double B) {
  return (double)axion_conversion_prob2_r(m_a, e, p, temp, length, radBore, g, B);
}
double sart_oracle_intensity_suppression2(double e, double dm, double dp, double p, double tm, double tp) {
  return (double)intensity_suppression2_r(e, dm, dp, p, tm, tp);
}
double sart_oracle_length_telescope(const sart_setup_t* s) { return (double)length_telescope_r(s); }
/* exp(logMassAttenuation(E)) in cm^2 / g, axionMassforMagnet.nim:70-73 (a fit to mass_attenuation_nist_data.txt) */
double sart_oracle_mass_attenuation(double energy_kev) { return (double)M_EXP(log_mass_attenuation((real)energy_kev)); }

/* ------------------------------------------------------------------------------------------
 * traceAxionWrapper (:2223-2244) and the accumulation that follows (:818-842, :2252-2257, :2800)
 * ---------------------------------------------------------------------------------------- */
static int resolve_threads(int n_threads) {
#ifdef _OPENMP
  if (n_threads <= 0) n_threads = omp_get_max_threads();
  return n_threads;
#else
  (void)n_threads;
  return 1;
#endif
}

int sart_oracle_trace_records(const sart_setup_t* setup, const sart_oracle_tables_t* tables,
                              const sart_trace_params_t* params, sart_axion_t* ax_buf, int n_threads) {
  int nt = resolve_threads(n_threads);
  int64_t n = (int64_t)params->n_rays;
#pragma omp parallel for num_threads(nt) schedule(static, 4096)
  for (int64_t i = 0; i < n; ++i) {
    double u[6];
    sart_axion_t res;
    memset(&res, 0, sizeof res); /* newSeq[Axion] zero-initialises (:2760) */
    sart_oracle_uniforms(params->seed, params->ray_id_offset + (uint64_t)i, u);
    sart_oracle_trace_axion(&res, setup, tables, params->flags, u);
    ax_buf[i] = res;
  }
  return nt;
}

/* The records of n rays whose six uniforms are GIVEN (u[n][6], draw order of SURVEY App. B: :433, :434, :436, :418, :419,
 * :464) instead of drawn: what tests/golden/uniform_keyed_*.npz store, so that those fixtures pin the physics of
 * traceAxion whatever maps (seed, ray id) to uniforms. */
int sart_oracle_trace_records_uniforms(const sart_setup_t* setup, const sart_oracle_tables_t* tables, uint32_t flags,
                                       const double* u, int64_t n, sart_axion_t* ax_buf, int n_threads) {
  int nt = resolve_threads(n_threads);
#pragma omp parallel for num_threads(nt) schedule(static, 4096)
  for (int64_t i = 0; i < n; ++i) {
    sart_axion_t res;
    memset(&res, 0, sizeof res);
    sart_oracle_trace_axion(&res, setup, tables, flags, u + 6 * i);
    ax_buf[i] = res;
  }
  return nt;
}

/* ---- Nim std/random (lib/pure/random.nim): xoroshiro128+ ------------------------------------------------------- */
static inline uint64_t nim_rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
uint64_t sart_oracle_nim_rand_next(sart_oracle_nim_rand_t* r) {
  const uint64_t s0 = r->a0;
  uint64_t s1 = r->a1;
  const uint64_t result = s0 + s1;
  s1 ^= s0;
  r->a0 = nim_rotl(s0, 55) ^ s1 ^ (s1 << 14);
  r->a1 = nim_rotl(s1, 36);
  return result;
}
static void nim_skip_random_numbers(sart_oracle_nim_rand_t* s) { /* the generator's 2^64 jump */
  static const uint64_t helper[2] = {0xbeac0467eba5facbull, 0xd86b048b86aa9922ull};
  uint64_t s0 = 0, s1 = 0;
  for (int i = 0; i < 2; ++i)
    for (int b = 0; b < 64; ++b) {
      if (helper[i] & (1ull << b)) { s0 ^= s->a0; s1 ^= s->a1; }
      (void)sart_oracle_nim_rand_next(s);
    }
  s->a0 = s0;
  s->a1 = s1;
}
void sart_oracle_nim_rand_init(sart_oracle_nim_rand_t* r, int64_t seed, int init_variant) {
  r->a0 = (uint64_t)(seed >> 16);
  r->a1 = (uint64_t)(seed & 0xffff);
  if (init_variant == 1) nim_skip_random_numbers(r);
  (void)sart_oracle_nim_rand_next(r);
}
double sart_oracle_nim_rand_float(sart_oracle_nim_rand_t* r) {
  const uint64_t x = sart_oracle_nim_rand_next(r);
  const uint64_t u = (0x3FFull << 52) | (x >> 12);
  double d;
  memcpy(&d, &u, sizeof d);
  return d - 1.0;
}

int sart_oracle_trace_records_nim_stream(const sart_setup_t* setup, const sart_oracle_tables_t* tables,
                                         const sart_trace_params_t* params, sart_axion_t* ax_buf, int init_variant) {
  sart_oracle_nim_rand_t rng;
  sart_oracle_nim_rand_init(&rng, (int64_t)params->seed, init_variant);
  const int draws = setup->test_active ? 4 : 6; /* SURVEY App. B */
  for (uint64_t k = 0; k < params->ray_id_offset * (uint64_t)draws; ++k) (void)sart_oracle_nim_rand_next(&rng);
  for (uint64_t i = 0; i < params->n_rays; ++i) {
    double u[6] = {0, 0, 0, 0, 0, 0};
    for (int k = 0; k < draws; ++k) u[k] = sart_oracle_nim_rand_float(&rng);
    sart_axion_t res;
    memset(&res, 0, sizeof res);
    sart_oracle_trace_axion(&res, setup, tables, params->flags, u);
    ax_buf[i] = res;
  }
  return 1;
}

/* One record into the fused accumulator (layout of include/sart.h SART_ACC_*). */
static void accumulate_record(const sart_axion_t* r, int stage, const sart_trace_params_t* p, double* acc,
                              int n_energies, int e_idx) {
  size_t nimg = (size_t)p->image_nx * (size_t)p->image_ny;
  double* sc = acc + nimg;
  sc[SART_ACC_N_RAYS] += 1.0;
  if (stage >= 1) sc[SART_ACC_N_REACHED_TELESCOPE] += 1.0;
  if (stage >= 2) sc[SART_ACC_N_SHELL_SELECTED] += 1.0;
  if (r->hitNickel) sc[SART_ACC_N_HIT_NICKEL] += 1.0;
  if (r->passedTillWindow) sc[SART_ACC_N_PASSED_TILL_WINDOW] += 1.0;
  if (r->passed) {
    sc[SART_ACC_N_PASSED] += 1.0;
    sc[SART_ACC_SUM_WEIGHTS] += r->weights;
    sc[SART_ACC_SUM_WEIGHTS_SQ] += r->weights * r->weights;
    sc[SART_ACC_SUM_X] += r->pointdataX;
    sc[SART_ACC_SUM_Y] += r->pointdataY;
    sc[SART_ACC_SUM_R] += r->pointdataR;
    /* prepareHeatmap :827-842 with norm = 1 */
    real stepsize_X = (p->image_x_max - p->image_x_min) / (real)p->image_nx;
    real stepsize_Y = (p->image_y_max - p->image_y_min) / (real)p->image_ny;
    real cx = M_FLOOR((r->pointdataX - p->image_x_min) / stepsize_X);
    real cy = M_FLOOR((r->pointdataY - p->image_y_min) / stepsize_Y);
    if (cx >= 0.0 && cx < (real)p->image_nx && cy >= 0.0 && cy < (real)p->image_ny)
      acc[(size_t)cy * (size_t)p->image_nx + (size_t)cx] += 1 * r->weights / 1.0;
    else
      sc[SART_ACC_N_OUTSIDE_IMAGE] += 1.0; /* IndexDefect in the reference */
    if (p->spectra) { /* histograms of generateResultPlots, layout of sart_accumulator_len_spectra */
      double* rad = sc + SART_ACC_COUNT;
      double* en = rad + 2 * (size_t)p->n_radial_bins;
      size_t ne1 = (size_t)n_energies + 1;
      int rb = (int)(r->pointdataR * ((double)p->n_radial_bins / p->radial_max));
      if (rb > p->n_radial_bins - 1) rb = p->n_radial_bins - 1;
      rad[rb] += 1.0;
      rad[(size_t)p->n_radial_bins + rb] += r->weights;
      en[e_idx] += 1.0;
      en[ne1 + e_idx] += r->weights;
      en[2 * ne1 + e_idx] += r->reflect;
    }
  }
}

int sart_oracle_trace_histogram(const sart_setup_t* setup, const sart_oracle_tables_t* tables,
                                const sart_trace_params_t* params, double* accumulator, int n_threads) {
  int nt = resolve_threads(n_threads);
  size_t len = params->spectra ? sart_accumulator_len_spectra(params->image_nx, params->image_ny, params->n_radial_bins,
                                                              tables->n_energies)
                               : sart_accumulator_len(params->image_nx, params->image_ny);
  if (!params->accumulate) memset(accumulator, 0, len * sizeof(double));
  int64_t n = (int64_t)params->n_rays;
#pragma omp parallel num_threads(nt)
  {
    double* local = (double*)calloc(len, sizeof(double));
#pragma omp for schedule(static, 4096)
    for (int64_t i = 0; i < n; ++i) {
      double u[6];
      sart_axion_t res;
      memset(&res, 0, sizeof res);
      sart_oracle_uniforms(params->seed, params->ray_id_offset + (uint64_t)i, u);
      int stage = 0;
      int e_idx = tables->n_energies; /* energies are discrete (:470); the test source uses the extra slot */
      trace_axion_impl(&res, setup, tables, params->flags, u, &stage, &e_idx);
      accumulate_record(&res, stage, params, local, tables->n_energies, e_idx);
    }
#pragma omp critical
    {
      for (size_t k = 0; k < len; ++k) accumulator[k] += local[k];
    }
    free(local);
  }
  return nt;
}
