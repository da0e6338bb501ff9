/*
 * trace_axion_wrapper.c - the drop-in boundary used from plain C, with nothing but the headers under include/ and the two shared
 * libraries (no Python, no torch): what `calculateFluxFractions` (raytracer.nim:2755-2776) does around
 * `traceAxionWrapper` (:2223-2244), and what `generateResultPlots` does first with its output (:2252-2257, :838-842).
 *
 *   gcc -std=c11 -O2 -I include integration/trace_axion_wrapper.c -L solaraxionraytracing_amd -lsart_host -lsart -lm \
 *       -Wl,-rpath,$PWD/solaraxionraytracing_amd -o trace_axion_wrapper && ./trace_axion_wrapper [n_rays]
 *
 * 1. newExperimentSetup + newDetectorSetup (:1411-1423, :1464-1528) through sart_host_new_full_setup: BabyIAXO, InGridIAXO,
 *    vacuum, XMM = config/config_default.toml:19-22.
 * 2. The captures of traceAxionWrapper.  The reference reads them from files it does not ship; this example makes small
 *    analytic ones (flat emission table -> sart_host_build_cdfs = the CDF loops of initFullSetup :2670-2705; reflectivity
 *    R = exp(-angle / 0.5 deg) exp(-E / 10 keV); constant transmissions).
 * 3. sart_trace_records into a caller-owned array of 208-byte Axion records  = traceAxionWrapper(axBuf, bufLen, ...).
 * 4. sart_trace_histogram over the same ray ids = the same rays without the records: prepareHeatmap + flux sum + counters.
 * 5. The two must agree: counters exactly, image and flux to rounding (the histogram adds in another order).
 * 6. The fused axion-mass scan (sart_trace_mass_scan): the same setup in the gas stage, every ray traced once and weighed for
 *    five axion masses - against the reference-shaped loop (set mAxion, :255, re-trace, sum the weights of the records).
 * 7. The fused angular scan (sart_trace_angular_scan): back in the vacuum stage, every ray sampled and cut once and turned through
 *    four telescope angles, against one traceAxionWrapper per angle (performAngularScan's shape, raytracer.nim:2791-2800).
 * Exit code 0 when everything agrees.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "sart_host.h"

#define CHECK(call)                                                                                  \
  do {                                                                                               \
    int rc_ = (call);                                                                                \
    if (rc_ != SART_OK) {                                                                            \
      fprintf(stderr, "%s failed (%d): %s | %s\n", #call, rc_, sart_last_error(), sart_host_last_error()); \
      return 2;                                                                                      \
    }                                                                                                \
  } while (0)

enum { N_RADII = 64, N_ENERGIES = 50, N_ANGLES = 200, N_REFL_E = 200, N_DET = 100, IMG = 256 };

int main(int argc, char** argv) {
  const uint64_t n_rays = argc > 1 ? strtoull(argv[1], NULL, 10) : 200000;

  /* 1. the setup */
  sart_setup_t setup;
  CHECK(sart_host_new_full_setup(SART_ES_BABYIAXO, SART_DK_INGRIDIAXO, SART_SK_VACUUM, SART_TK_XMM, 0u, NULL, NULL, NULL, &setup));

  /* 2. the tables */
  static double em[N_RADII * N_ENERGIES], radii[N_RADII], energies[N_ENERGIES], rcdf[N_RADII], ecdf[N_RADII * N_ENERGIES];
  for (int r = 0; r < N_RADII; ++r) radii[r] = 0.0015 + 0.0005 * r;                           /* :438 */
  for (int e = 0; e < N_ENERGIES; ++e) energies[e] = 1e-3 + (15.0 - 1e-3) * e / (N_ENERGIES - 1); /* readOpacityFile.nim:609 */
  for (int i = 0; i < N_RADII * N_ENERGIES; ++i) em[i] = 1.0;
  CHECK(sart_host_build_cdfs(em, radii, energies, N_RADII, N_ENERGIES, rcdf, ecdf));
  static double refl[N_ANGLES * N_REFL_E];
  const double a_min = 0.0, a_max = 1.5, e_min = 0.03, e_max = 15.0;                           /* the H5 grid, SURVEY 8(b) */
  for (int a = 0; a < N_ANGLES; ++a)
    for (int e = 0; e < N_REFL_E; ++e)
      refl[a * N_REFL_E + e] = exp(-(a_min + (a_max - a_min) * a / (N_ANGLES - 1)) / 0.5) * exp(-(e_min + (e_max - e_min) * e / (N_REFL_E - 1)) / 10.0);
  static double det_x[N_DET], strongback[N_DET], window[N_DET], gas_abs[N_DET];
  for (int i = 0; i < N_DET; ++i) {
    det_x[i] = 16.0 * i / (N_DET - 1);
    strongback[i] = 0.3;
    window[i] = 0.8;
    gas_abs[i] = 0.9;
  }

  sart_context* ctx = NULL;
  CHECK(sart_create(0, &ctx));
  CHECK(sart_set_setup(ctx, &setup));
  CHECK(sart_set_solar_tables(ctx, rcdf, ecdf, energies, N_RADII, N_ENERGIES));
  CHECK(sart_set_reflectivity(ctx, 1, N_ANGLES, N_REFL_E, a_min, a_max, e_min, e_max, refl));
  CHECK(sart_set_detector_tables(ctx, det_x, strongback, N_DET, det_x, window, N_DET, det_x, gas_abs, N_DET));

  sart_trace_params_t p;
  memset(&p, 0, sizeof p);
  p.n_rays = n_rays;
  p.seed = 299792458;   /* randomize(299792458), :276 */
  p.ray_id_offset = 0;
  p.flags = 0;
  p.image_nx = p.image_ny = IMG;
  p.image_x_min = p.image_y_min = 0.0;
  p.image_x_max = p.image_y_max = setup.chip_x_max > 0.0 ? setup.chip_x_max : 14.0;   /* ChipXMax, :260-272 */

  /* 3. the drop-in call */
  sart_axion_t* ax_buf = calloc(n_rays, sizeof *ax_buf);   /* newSeq[Axion](bufLen), :2760 */
  if (!ax_buf) return 2;
  CHECK(sart_host_trace_axion_wrapper(ctx, ax_buf, (int64_t)n_rays, p.seed, p.ray_id_offset, p.flags));

  /* what generateResultPlots starts with: filterIt(it.passed) (:2252-2257) and the heat map (:838-842) */
  static double img_rec[IMG * IMG];
  uint64_t n_passed = 0, n_till = 0, n_nickel = 0;
  double flux = 0.0;
  const double inv_step = IMG / (p.image_x_max - p.image_x_min);   /* the library bins with the reciprocal of the step */
  for (uint64_t i = 0; i < n_rays; ++i) {
    const sart_axion_t* a = &ax_buf[i];
    n_till += a->passedTillWindow;
    n_nickel += a->hitNickel;
    if (!a->passed) continue;
    ++n_passed;
    flux += a->weights;
    const double fx = (a->pointdataX - p.image_x_min) * inv_step, fy = (a->pointdataY - p.image_y_min) * inv_step;
    if (fx >= 0.0 && fx < IMG && fy >= 0.0 && fy < IMG) img_rec[(int)fy * IMG + (int)fx] += a->weights;
  }

  /* 3b. the passed rays only (sart_trace_records_passed): the records `filterIt(it.passed)` keeps, in the same order, and the
   *     three counts - with a buffer a quarter over the number that passed */
  int passed_ok = 0;
  {
    const uint64_t cap = n_passed + n_passed / 4 + 1;
    sart_axion_t* only = malloc(cap * sizeof *only);
    sart_record_counts_t rc;
    if (!only) return 2;
    CHECK(sart_trace_records_passed(ctx, &p, only, cap, &rc));
    passed_ok = rc.n_rays == n_rays && rc.n_passed == n_passed && rc.n_passed_till_window == n_till && rc.n_hit_nickel == n_nickel;
    uint64_t k = 0;
    for (uint64_t i = 0; i < n_rays && passed_ok; ++i)
      if (ax_buf[i].passed) passed_ok = memcmp(&only[k++], &ax_buf[i], sizeof *only) == 0;
    passed_ok = passed_ok && k == n_passed;
    free(only);
    /* the record entries keep their device scratch in the context (it only grows); a host that is done with records hands it
     * back (ABI 5) - the histogram and scan entries below do not use it */
    CHECK(sart_release_scratch(ctx));
  }

  /* 4. the same rays through the fused histogram entry */
  static double image[IMG * IMG];
  sart_summary_t s;
  CHECK(sart_trace_histogram(ctx, &p, image, &s));

  /* 5. compare */
  double max_diff = 0.0, img_max = 0.0;
  for (int i = 0; i < IMG * IMG; ++i) {
    const double d = fabs(image[i] - img_rec[i]);
    if (d > max_diff) max_diff = d;
    if (image[i] > img_max) img_max = image[i];
  }
  const int ok = s.v[SART_ACC_N_PASSED] == (double)n_passed && s.v[SART_ACC_N_PASSED_TILL_WINDOW] == (double)n_till &&
                 s.v[SART_ACC_N_HIT_NICKEL] == (double)n_nickel && s.v[SART_ACC_N_RAYS] == (double)n_rays &&
                 fabs(s.v[SART_ACC_SUM_WEIGHTS] - flux) <= 1e-11 * flux && max_diff <= 1e-11 * img_max && n_passed > 0;
  printf("{\"abi\": %d, \"build\": \"%s\", \"n_rays\": %llu, \"passed\": %llu, \"passed_till_window\": %llu, \"hit_nickel\": %llu, "
         "\"flux_records\": %.17g, \"flux_histogram\": %.17g, \"image_max_abs_diff\": %.3g, \"passed_only_records_agree\": %s, \"agree\": %s}\n",
         sart_abi_version(), sart_build_id(), (unsigned long long)n_rays, (unsigned long long)n_passed, (unsigned long long)n_till,
         (unsigned long long)n_nickel, flux, s.v[SART_ACC_SUM_WEIGHTS], max_diff, passed_ok ? "true" : "false", (ok && passed_ok) ? "true" : "false");
  /* 6. gas stage: one pass over the rays for all masses vs one traceAxionWrapper per mass */
  enum { N_MASSES = 5 };
  const double masses[N_MASSES] = {0.0, 0.004, 0.008235, 0.012, 0.05};   /* eV; m_gamma = 0.008235 eV for this magnet (literal units) */
  sart_setup_t gas = setup;
  gas.stage = SART_SK_GAS;
  CHECK(sart_set_setup(ctx, &gas));
  double scan[(N_MASSES + 1) * SART_SCAN_ROW];
  CHECK(sart_trace_mass_scan(ctx, &p, masses, N_MASSES, scan));
  int scan_ok = scan[N_MASSES * SART_SCAN_ROW + SART_SCAN_N_RAYS] == (double)n_rays;
  double scan_max_rel = 0.0;
  for (int k = 0; k < N_MASSES; ++k) {
    CHECK(sart_set_axion_mass(ctx, masses[k]));
    CHECK(sart_host_trace_axion_wrapper(ctx, ax_buf, (int64_t)n_rays, p.seed, p.ray_id_offset, p.flags));
    double f = 0.0;
    uint64_t np = 0;
    for (uint64_t i = 0; i < n_rays; ++i)
      if (ax_buf[i].passed) { f += ax_buf[i].weights; ++np; }
    const double* row = scan + k * SART_SCAN_ROW;
    const double rel = fabs(row[SART_SCAN_SUM_WEIGHTS] - f) / f;
    if (rel > scan_max_rel) scan_max_rel = rel;
    scan_ok = scan_ok && row[SART_SCAN_N_PASSED] == (double)np && rel <= 1e-9 && np > 0;
  }
  scan_ok = scan_ok && scan[2 * SART_SCAN_ROW + SART_SCAN_SUM_WEIGHTS] > scan[0 * SART_SCAN_ROW + SART_SCAN_SUM_WEIGHTS] &&
            scan[2 * SART_SCAN_ROW + SART_SCAN_SUM_WEIGHTS] > scan[4 * SART_SCAN_ROW + SART_SCAN_SUM_WEIGHTS];   /* the resonance */
  printf("{\"mass_scan\": {\"masses\": %d, \"flux_on_resonance\": %.17g, \"max_rel_diff_to_per_mass_records\": %.3g, \"agree\": %s}}\n",
         N_MASSES, scan[2 * SART_SCAN_ROW + SART_SCAN_SUM_WEIGHTS], scan_max_rel, scan_ok ? "true" : "false");
  /* 7. vacuum stage again: one pass over the rays for all telescope angles (sart_trace_angular_scan) vs performAngularScan's
   *    shape - set the angle, one traceAxionWrapper, sum the weights of the passed rays (raytracer.nim:2791-2800) */
  enum { N_ANGLES = 4 };
  const double angles[N_ANGLES] = {0.005, 0.01, 0.02, 0.04};   /* degrees; 0.04 deg walks the spot 5 mm across the 14 mm chip (f = 7.5 m) */
  CHECK(sart_set_setup(ctx, &setup));
  double ascan[(N_ANGLES + 1) * SART_ASCAN_ROW];
  CHECK(sart_trace_angular_scan(ctx, &p, angles, N_ANGLES, ascan));
  int ascan_ok = ascan[N_ANGLES * SART_ASCAN_ROW + SART_ASCAN_N_RAYS] == (double)n_rays;
  double ascan_max_rel = 0.0;
  for (int k = 0; k < N_ANGLES; ++k) {
    CHECK(sart_set_telescope_angles(ctx, NAN, angles[k]));
    CHECK(sart_host_trace_axion_wrapper(ctx, ax_buf, (int64_t)n_rays, p.seed, p.ray_id_offset, p.flags));
    double f = 0.0;
    uint64_t np = 0, nn = 0;
    for (uint64_t i = 0; i < n_rays; ++i) {
      if (ax_buf[i].passed) { f += ax_buf[i].weights; ++np; }
      if (ax_buf[i].hitNickel) ++nn;
    }
    const double* row = ascan + k * SART_ASCAN_ROW;
    const double rel = fabs(row[SART_ASCAN_SUM_WEIGHTS] - f) / f;
    if (rel > ascan_max_rel) ascan_max_rel = rel;
    const int row_ok = row[SART_ASCAN_N_PASSED] == (double)np && row[SART_ASCAN_N_HIT_NICKEL] == (double)nn && rel <= 1e-9 && np > 0;
    if (!row_ok)
      fprintf(stderr, "angle %g deg: scan passed %.0f nickel %.0f flux %.17g | records passed %llu nickel %llu flux %.17g\n", angles[k],
              row[SART_ASCAN_N_PASSED], row[SART_ASCAN_N_HIT_NICKEL], row[SART_ASCAN_SUM_WEIGHTS], (unsigned long long)np, (unsigned long long)nn, f);
    ascan_ok = ascan_ok && row_ok;
  }
  printf("{\"angular_scan\": {\"angles\": %d, \"flux_at_first_angle\": %.17g, \"max_rel_diff_to_per_angle_records\": %.3g, \"agree\": %s}}\n",
         N_ANGLES, ascan[SART_ASCAN_SUM_WEIGHTS], ascan_max_rel, ascan_ok ? "true" : "false");
  free(ax_buf);
  CHECK(sart_destroy(ctx));
  return (ok && passed_ok && scan_ok && ascan_ok) ? 0 : 1;
}
