/*
 * sart_oracle.h — CPU oracle (TEST INFRASTRUCTURE, not product code).
 *
 * A plain-C restatement of the reference's per-ray path `traceAxion`
 * (src/raytracer.nim:1736-2221) and of the accumulation that follows it.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library; the product path (libsart.so) never does.
 *
 * It consumes the same PODs as the C-ABI (include/sart.h) so that the HIP path
 * and the oracle are driven with byte-identical inputs.
 */
#ifndef SART_ORACLE_H
#define SART_ORACLE_H

#include "../include/sart.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Host-memory views of the tables traceAxion captures. Nothing is copied. */
typedef struct sart_oracle_tables_t {
  /* solar tables, FullRaytraceSetup raytracer.nim:237-241 */
  const double* flux_radius_cdf;  /* [n_radii]              */
  const double* diff_flux_cdfs;   /* [n_radii][n_energies]  */
  const double* energies_kev;     /* [n_energies]           */
  int32_t n_radii, n_energies;
  /* reflectivity grids, raytracer.nim:1160-1231 */
  const double* refl_data;        /* [n_coatings][n_angles][n_energies] */
  int32_t refl_n_coatings, refl_n_angles, refl_n_energies;
  int32_t _pad;
  double refl_angle_min, refl_angle_max, refl_energy_min, refl_energy_max;
  /* 1-D tables, raytracer.nim:1522-1527 */
  const double* strongback_x; const double* strongback_y; int32_t n_strongback; int32_t _pad1;
  const double* window_x; const double* window_y; int32_t n_window; int32_t _pad2;
  const double* gas_abs_x; const double* gas_abs_y; int32_t n_gas_abs; int32_t _pad3;
} sart_oracle_tables_t;

/* The six uniforms of ray `ray_id` (Philox4x32-7, key = seed, counter = (id, 0, 0): ONE block (x, y, z, w) + word `ray_id` s of a
 * word stream shared by consecutive rays - counter = (id >> 2, 3, 0), word id & 3 -: u2 = x / 2^32, u5 = z / 2^32 (the CDF draws),
 * u3 = s / 2^32 (disc radius), u0 = (y >> 11) / 2^21, u1 = (w >> 11) / 2^21 (solar point), u4 = the low 11 bits of y and w as a
 * 22-bit fraction (disc angle).  This is the stream's definition; the HIP kernel's uniforms_of computes the same bits. */
void sart_oracle_uniforms(uint64_t seed, uint64_t ray_id, double u[6]);

/* traceAxion for one ray given its uniforms; *res must be zero-initialised by the caller
 * exactly as `newSeq[Axion]` does (raytracer.nim:2760). */
void sart_oracle_trace_axion(sart_axion_t* res, const sart_setup_t* setup,
                             const sart_oracle_tables_t* tables, uint32_t flags,
                             const double u[6]);

/* traceAxionWrapper (raytracer.nim:2223-2244): n_rays records into ax_buf. n_threads<=0: all cores. */
int sart_oracle_trace_records(const sart_setup_t* setup, const sart_oracle_tables_t* tables,
                              const sart_trace_params_t* params, sart_axion_t* ax_buf,
                              int n_threads);

/* Records of n rays with GIVEN uniforms u[n][6] (draw order of SURVEY App. B). Returns threads used. */
int sart_oracle_trace_records_uniforms(const sart_setup_t* setup, const sart_oracle_tables_t* tables, uint32_t flags,
                                       const double* u, int64_t n, sart_axion_t* ax_buf, int n_threads);

/* ---- the reference's own random stream (for fixtures produced by a Nim build of the reference) ----------------------
 * Nim std/random = xoroshiro128+ (rotations 55 / 14 / 36), ONE global stream seeded by `randomize(299792458)`
 * (raytracer.nim:276); rand(1.0) = 52 mantissa bits of next() under the exponent of 1.0, minus 1.0.
 * init_variant 0: initRand of Nim < 1.4 (a0 = seed shr 16, a1 = seed and 0xffff, one discarded draw);
 * init_variant 1: Nim >= 1.4 (the same, with skipRandomNumbers — the 2^64 jump — before the discarded draw). */
typedef struct sart_oracle_nim_rand_t { uint64_t a0, a1; } sart_oracle_nim_rand_t;
void sart_oracle_nim_rand_init(sart_oracle_nim_rand_t* r, int64_t seed, int init_variant);
uint64_t sart_oracle_nim_rand_next(sart_oracle_nim_rand_t* r);
double sart_oracle_nim_rand_float(sart_oracle_nim_rand_t* r); /* rand(1.0) */

/* traceAxionWrapper with the reference's stream instead of the per-ray Philox blocks: ONE thread, rays in index order,
 * every ray takes its draws from the running stream in the reference's order (six per ray from the Sun —
 * theta1, theta2, radius CDF, disk radius, disk angle, energy CDF —, four per ray of the X-ray test source).  This is what
 * a single-threaded run of the reference (WEAVE_NUM_THREADS=1) computes; with more threads the reference's shared stream
 * is racy and no run is reproducible.  params->seed is the randomize() seed; ray_id_offset rays are skipped first. */
int sart_oracle_trace_records_nim_stream(const sart_setup_t* setup, const sart_oracle_tables_t* tables,
                                         const sart_trace_params_t* params, sart_axion_t* ax_buf, int init_variant);

/* trace + prepareHeatmap(norm=1) + flux sum + counters into `accumulator`
 * (sart_accumulator_len doubles, layout of include/sart.h). Returns threads used. */
int sart_oracle_trace_histogram(const sart_setup_t* setup, const sart_oracle_tables_t* tables,
                                const sart_trace_params_t* params, double* accumulator,
                                int n_threads);

/* Pieces exported for known-answer tests. */
double sart_oracle_conversion_prob(double B_tesla, double g_agamma_inv_gev, double length_mm);
double sart_oracle_eff_photon_mass2(double p_mbar, double length_m, double rad_bore_m, double temp_k);
double sart_oracle_axion_conversion_prob2(double m_a, double energy_kev, double pressure,
                                          double temp, double length_m, double rad_bore_m,
                                          double g_agamma, double B);
double sart_oracle_intensity_suppression2(double energy_kev, double distance_magnet_m,
                                          double distance_pipe_m, double pressure,
                                          double temp_magnet, double temp_pipe);
double sart_oracle_mass_attenuation(double energy_kev);
double sart_oracle_bilinear(const double* z, int32_t nx, int32_t ny, double xmin, double xmax,
                            double ymin, double ymax, double x, double y);
double sart_oracle_linear1d(const double* xs, const double* ys, int32_t n, double x);
int64_t sart_oracle_lower_bound(const double* a, int64_t n, double key);
int sart_oracle_almost_equal(double x, double y);
/* Mirror helpers: shape 0 cone, 1 parabolic, 2 hyperbolic (MirrorShapeKind :20-21). */
void sart_oracle_find_pos(int shape, const double point_xrt[3], const double point_cb[3],
                          double r1, double angle_rad, double l_mirror, double dist_mirr,
                          double focal_length, double out[3]);
void sart_oracle_vector_after_mirror(const double point_xrt[3], const double point_cb[3],
                                     const double point_mirror[3], double angle_rad, double r1,
                                     double l_mirror, double focal_length, int shape,
                                     double out[3]);
double sart_oracle_mirror_angle_deg(const double point_xrt[3], const double point_cb[3],
                                    const double point_mirror[3], double angle_rad, double r1,
                                    double l_mirror, double focal_length, int shape);
double sart_oracle_length_telescope(const sart_setup_t* setup);

#ifdef __cplusplus
}
#endif
#endif
