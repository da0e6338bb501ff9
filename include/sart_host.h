/*
 * sart_host.h — C bindings of the host-side mirror of the reference's setup / driver layer
 * for the per-ray hot path (libsart_host.so, C++ above the sart.h C-ABI).
 *
 * The reference's host is Nim; no Nim toolchain exists in the build image, so the host
 * layer that sits directly above the hot path is written in C++ with the reference's
 * names, argument meaning and error behaviour.  Nothing here touches the GPU except the
 * two drivers at the bottom, which only call the sart.h entry points.
 */
#ifndef SART_HOST_H
#define SART_HOST_H

#include "sart.h"
#include "sart_emission.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Optional `[Magnet]`, `[TestXraySource]`, `[DetectorInstallation]` blocks of config.toml
 * (config/config_default.toml:24-50; parsers raytracer.nim:1032-1096).  A NULL pointer means
 * "useConfig = false and the corresponding flag not given". */
typedef struct sart_magnet_config_t {
  double B, radiusCB, lengthColdbore, lengthB, pGasRoom, tGas;
} sart_magnet_config_t;
typedef struct sart_test_source_config_t {
  int32_t active, parallel;
  double energy, distance, radius, offAxisUp, offAxisLeft, activity, lengthCol;
} sart_test_source_config_t;
typedef struct sart_detector_install_config_t {
  double distanceDetectorXRT, distanceWindowFocalPlane, lateralShift, transversalShift;
} sart_detector_install_config_t;

const char* sart_host_last_error(void);

/* newExperimentSetup (raytracer.nim:1411-1423) = initMagnet :1098 + initTelescope :1251 +
 * initTestXraySource :1350 + initPipes :1125 + initDetectorInstallation :1381, followed by
 * newDetectorSetup (:1464-1528, without the file reads) and the module constants :248-272.
 * Fails (like the reference's doAssert) for tkCustomBabyIAXO / tkOther. */
int sart_host_new_full_setup(int32_t experiment, int32_t detector, int32_t stage, int32_t telescope,
                             uint32_t flags, const sart_magnet_config_t* magnet_cfg,
                             const sart_test_source_config_t* source_cfg,
                             const sart_detector_install_config_t* install_cfg, sart_setup_t* out);

/* calcWindowVals, raytracer.nim:1431-1462 */
int sart_host_calc_window_vals(double radius_window, int32_t number_of_strips,
                               double open_aperture_ratio, double* width, double* dist);

/* The CDF construction of initFullSetup (raytracer.nim:2670-2705):
 * diffFlux[r][e] = emRate[r][e] * E^2 * r^2; per-radius cumulative sum / last; radius cumulative / last. */
int sart_host_build_cdfs(const double* em_rates, const double* radii, const double* energies_kev,
                         int32_t n_radii, int32_t n_energies, double* flux_radius_cdf_out,
                         double* diff_flux_cdfs_out);

/* The three 1-D tables of newDetectorSetup (:1509-1527) from the raw TSV columns:
 * strongback = T_Si * T_Al, window = T_Si3N4 * T_Al, gas absorption = 1 - T_Ar; x = eV / 1000. */
int sart_host_detector_tables(const double* energy_ev, const double* t_si3n4, const double* t_si,
                              const double* t_al, int32_t n, const double* argon_energy_ev,
                              const double* t_argon, int32_t n_argon, double* x_kev_out,
                              double* strongback_out, double* window_out, double* argon_x_kev_out,
                              double* gas_abs_out);

/* traceAxionWrapper, raytracer.nim:2223-2244: identical to sart_trace_records (kept so that
 * host code reads like the reference). */
int sart_host_trace_axion_wrapper(sart_context* ctx, sart_axion_t* ax_buf, int64_t buf_len,
                                  uint64_t seed, uint64_t ray_id_offset, uint32_t flags);

/* performAngularScan, raytracer.nim:2778-2802, in the reference's shape - a host loop: for each angle set
 * telescope_turned_y, trace n_rays_per_angle FRESH rays (flux-only launches: no image is accumulated) and sum the weights
 * of the passed rays; fluxes_out[i] holds the raw sums, rel_fluxes_out[i] the max-normalised curve (:2801-2802).  Angle i
 * uses ray ids [ray_id_offset + i*n_rays_per_angle, ...): independent samples per angle, like the reference's running
 * random stream.  The context's setup has its own angle back afterwards (the reference scans a copy, :2794-2797). */
int sart_host_perform_angular_scan(sart_context* ctx, const double* angles_deg, int32_t n_angles,
                                   uint64_t n_rays_per_angle, uint64_t seed, uint64_t ray_id_offset,
                                   uint32_t flags, double* fluxes_out, double* rel_fluxes_out);

/* The same curve through the fused scan kernel (sart_trace_angular_scan, include/sart.h): ONE pass over the ray ids
 * [ray_id_offset, ray_id_offset + n_rays) - every ray is sampled and taken through bore and pipes once and turned through
 * every angle; the same rays for all angles (common random numbers).  flux_sq_out[i] (may be NULL) = sum of the squared
 * weights (Monte-Carlo error of the flux: sqrt), n_passed_out[i] (may be NULL) the passed rays.  The setup is not touched. */
int sart_host_angular_scan(sart_context* ctx, const double* angles_deg, int32_t n_angles, uint64_t n_rays, uint64_t seed,
                           uint64_t ray_id_offset, uint32_t flags, double* fluxes_out, double* rel_fluxes_out,
                           double* flux_sq_out, double* n_passed_out);

/* Axion-mass scan (BASELINE configs[4]; the reference only has the constant mAxion, raytracer.nim:255).
 * Gas stage: ONE pass over the ray ids [ray_id_offset, ray_id_offset + n_rays) through the fused scan kernel
 * (sart_trace_mass_scan, include/sart.h): every ray is traced once and weighed for every mass - the same rays for all masses
 * (common random numbers).  Vacuum stage: the conversion probability (:363-365) has no m_a in it - one launch, the same flux
 * for every mass.  fluxes_out[i] = sum of the weights of the passed rays for masses_ev[i]; flux_sq_out[i] (may be NULL) the sum
 * of their squares (Monte-Carlo error of the flux: sqrt(flux_sq)); n_passed_out[i] (may be NULL) the number of passed rays. */
int sart_host_axion_mass_scan(sart_context* ctx, const double* masses_ev, int32_t n_masses, uint64_t n_rays, uint64_t seed,
                              uint64_t ray_id_offset, uint32_t flags, double* fluxes_out, double* flux_sq_out,
                              double* n_passed_out);
/* The scan of rounds 1-3 under its old name, with its old meaning: a host loop in performAngularScan's shape - mass i on its OWN
 * block of fresh rays, ids [ray_id_offset + i n_rays_per_mass, ...) - through flux-only launches.  (Round 4 had routed this name
 * through the fused kernel: same signature, correlated scan points, other numbers for the same seed.  The fused scan is
 * sart_host_axion_mass_scan above and nowhere else.)  The context's own axion mass is put back afterwards. */
int sart_host_perform_axion_mass_scan(sart_context* ctx, const double* masses_ev, int32_t n_masses,
                                      uint64_t n_rays_per_mass, uint64_t seed, uint64_t ray_id_offset,
                                      uint32_t flags, double* fluxes_out);

/* HDF5 reflectivity files read by initReflectivity (raytracer.nim:1174-1186, :1196-1209; written by
 * tools/llnl_layer_reflectivity.nim:62-80 and tools/convert_reflectivities_to_h5.nim:29-48): datasets `/Energy` (nE,1) keV,
 * `/Angles` (nA,1) deg, and `/Reflectivity` (single coating) or `/Reflectivity0..N-1` (LLNL), f64, declared shape (nE, nA)
 * but laid out [angle][energy] row-major.  Only min/max of the axes are used (uniform grid).  libhdf5 is loaded at run
 * time (dlopen); without it these calls return SART_ERR_UNSUPPORTED. */
int sart_host_h5_reflectivity_info(const char* path, int32_t* n_coatings, int32_t* n_angles, int32_t* n_energies,
                                   double* angle_min, double* angle_max, double* energy_min, double* energy_max);
int sart_host_h5_read_reflectivity(const char* path, double* data_out /* [n_coatings][n_angles][n_energies] */);
int sart_host_h5_write_reflectivity(const char* path, int32_t n_coatings, int32_t n_angles, int32_t n_energies,
                                    const double* angles_deg, const double* energies_kev, const double* data);

/* Containment radii of generateResultPlots (raytracer.nim:2459-2527) from the radial histograms of a spectra trace
 * (bin k = [k, k+1) * radial_max / n_bins):
 *   r_sigma1 / r_sigma2     radii holding round(0.68 n) / round(0.955 n) of the passed rays (pointR[sigma1 - 1], :2472-2473)
 *   r_sigma1_w / r_sigma2_w weighted versions: last radius whose cumulative weight stays below 0.68 / 0.955 of the
 *                           total (rSigma1W / rSigma2W, :2511-2524)
 * each reported as the upper edge of the bin in which the threshold is crossed (resolution radial_max / n_bins). */
int sart_host_containment_radii(const double* radial_counts, const double* radial_weights, int32_t n_bins,
                                double radial_max, double* r_sigma1, double* r_sigma2, double* r_sigma1_w,
                                double* r_sigma2_w);

/* plotHeatmap's CSV (raytracer.nim:866-921): `axion_image_{year}{suffix}.csv` with the columns
 * x, y, photon flux, yr0, yr02, x-position [mm], y-position [mm], xr, xrneg, yr, xr2, xrneg2, yr2
 * for a width x width image over 0 .. chip_max mm.  Returns the total flux (the `echo` of :886) in *flux_out. */
int sart_host_write_image_csv(const char* path, const double* image, int32_t width, double chip_max,
                              double r_sigma1, double r_sigma2, double* flux_out);

/* First loop of calculateOpacities (readOpacityFile.nim:655-705): the per-radius plasma quantities the emission-table
 * producer (sart_emission.h) needs, from the columns of the solar-model file (AGSS09_solar_model_stripped.dat):
 * temp_K[n] = Temp, rho[n] = Rho, mass_fractions[n][29] = H1 He4 He3 C12 C13 N14 N15 O16 O17 O18 Ne Na Mg Al Si P S Cl Ar K
 * Ca Sc Ti V Cr Mn Fe Co Ni (the order of `elements`, :125-128). */
int sart_host_solar_zones(const double* temp_K, const double* rho, const double* mass_fractions, int32_t n_radii,
                          sart_solar_zone_t* zones_out);

/* n_Z of the same loop (:655-679): number densities per PROTON NUMBER in 1/cm^3, n_z_out[n_radii][29].  [1] hydrogen, [2]
 * helium, [6] carbon, [7] nitrogen, [8] oxygen (isotopes joined with their mean mass), [10]..[28] neon .. nickel; the
 * entries the reference never writes ([0], [3], [4], [5], [9]) are 0. */
int sart_host_solar_number_densities(const double* rho, const double* mass_fractions, int32_t n_radii, double* n_z_out);

/* ---- OPCD 3.3 monochromatic opacity files (readOpacityFile.nim:146-296, :731-745) -------------------------------------
 * `<opcd_path>/OPCD_3.3/mono/fmZZ.TTT`: one header line, then per electron-density index a table = header line 1 (starts
 * with the density index), header line 2 (unused), a line with the number of table lines (0 = 10000), the table lines
 * (`opacity`, or `abscissa opacity`; a 10000-line table is indexed by its line count 1..10000).  `fm01.mesh`: blank-separated
 * columns with a header line, column `u`, 10001 lines.  The data is not part of the reference repository (OPCD licence). */

/* Column u of a mesh file.  u_out may be NULL (then only *n_out is set). */
int sart_host_opcd_read_mesh(const char* path, double* u_out, int32_t capacity, int32_t* n_out);
/* Element and temperature index from the file name, the density indices and line counts of every table in file order. */
int sart_host_opcd_file_info(const char* path, int32_t* element, int32_t* temp_index, int32_t* n_tables, int32_t* densities_out,
                             int32_t* lengths_out, int32_t capacity);
/* One density table of a file: abscissae (line numbers 1..n for a 10000-line table) and opacities. */
int sart_host_opcd_read_table(const char* path, int32_t density, double* x_out, double* y_out, int32_t capacity, int32_t* n_out);

/* Everything sart_emission_abs_coefs needs for the zones of one solar model: the mesh and, for every temperature index of
 * the zones, the files of the 17 elements the reference looks up (:827-831; a missing file or density table fails like the
 * reference's KeyError), parsed on n_threads threads (0 = up to 16).  Only the density tables the zones use are converted. */
typedef struct sart_opcd_set sart_opcd_set;
int sart_host_opcd_load(const char* opcd_path, const sart_solar_zone_t* zones, int32_t n_radii, int32_t n_threads, sart_opcd_set** out);
const sart_opacity_tables_t* sart_host_opcd_tables(const sart_opcd_set* set);   /* valid until sart_host_opcd_free */
int sart_host_opcd_slot(const sart_opcd_set* set, int32_t slot, int32_t* temp_index, int32_t* ne_index);
void sart_host_opcd_free(sart_opcd_set* set);

/* getFluxFractionR (readOpacityFile.nim:535-584): differential flux at Earth in 1/(keV y m^2) per energy, summed over the
 * radial zones of an emission table [n_radii][n_energies] (zone r at 0.0015 + 0.0005 r solar radii). */
int sart_host_flux_spectrum(const double* em_rates, int32_t n_radii, const double* energies_kev, int32_t n_energies,
                            double* diff_flux_out);

#ifdef __cplusplus
}
#endif
#endif
