/*
 * sart_emission.h — C-ABI of the solar emission-table producer.
 *
 * Replaces the cell loop of `calculateOpacities` (src/readOpacityFile.nim:745-860): for every solar
 * radius R (1968 zones of the AGSS09 model) and every axion energy (1500 values) the emission rate
 * emRate[R][E] = Compton + term1 + ee-bremsstrahlung + free-free + Primakoff + longitudinal and
 * transverse plasmon conversion + 57Fe line (:849), all in keV units as in the reference.  The cells
 * are independent; two of the terms hold the integral `fNew` (:312-326).  The result is the
 * `emRates` column of `solar_model_dataframe.csv` that `initFullSetup` reads (raytracer.nim:2647-2668)
 * and turns into the sampling CDFs (sart_host_build_cdfs).
 *
 * The absorption coefficient absCoef[R][E] (:790-823) is interpolated by the reference from the OPCD 3.3
 * monochromatic opacity files, which are not redistributable and not part of the reference repository.
 * A caller that has them builds the table with sart_emission_abs_coefs* below (files read by sart_host_opcd_load,
 * sart_host.h) and passes it on (keV, same shape as the output); NULL means zero, which is also what the reference
 * computes outside 0.0732 < w < 20 (:801-808).
 */
#ifndef SART_EMISSION_H
#define SART_EMISSION_H

#include "sart.h"

#ifdef __cplusplus
extern "C" {
#endif

/* One radial zone of the solar model, as the first loop of calculateOpacities derives it from the model
 * file (readOpacityFile.nim:655-705).  Built by sart_host_solar_zones (sart_host.h). */
typedef struct sart_solar_zone_t {
  double radius_frac; /* 0.0015 + 0.0005 i, fraction of the solar radius (:701, :751)                    */
  double temp_K;      /* column Temp (:692)                                                              */
  double rho;         /* column Rho, g/cm^3                                                              */
  double n_e;         /* electrons / cm^3, full ionisation, sum over the 29 species (:682-684)           */
  double n_H;         /* n_Z[R][1], 1/cm^3 (:661)                                                        */
  double n_He;        /* n_Z[R][2], 1/cm^3, He4 + He3 with their mean mass (:662-670)                    */
  int32_t temp_index; /* OPCD temperature grid point 140 + 2 i nearest to log10(T)/0.025 (:687-691); the
                         energy variable w = E / (10^(0.025 temp_index) * 8.617e-8) uses it (:768, :793)  */
  int32_t ne_index;   /* OPCD electron-density grid point 74 + 2 i (:693-697); informational              */
} sart_solar_zone_t;

/* Term selection (bit set) for sart_emission_table; the reference sums all of them (:849). */
enum {
  SART_EM_COMPTON = 1u << 0,       /* comptonEmrate :360-362 */
  SART_EM_TERM1 = 1u << 1,         /* term1 :369-371 (needs abs_coefs; ff + fb + bb processes) */
  SART_EM_EE_BREMS = 1u << 2,      /* bremsEmrate :364-367 */
  SART_EM_FREE_FREE = 1u << 3,     /* freefreeEmrate :378-381 */
  SART_EM_PRIMAKOFF = 1u << 4,     /* primakoff :394-418 */
  SART_EM_LONG_PLASMON = 1u << 5,  /* longPlasmon :420-437 */
  SART_EM_TRANS_PLASMON = 1u << 6, /* transPlasmon :439-453 */
  SART_EM_IRON57 = 1u << 7,        /* iron :455-468 */
  SART_EM_ALL = 0xFFu,
  SART_EM_N_TERMS = 8
};

typedef struct sart_emission_params_t {
  double g_ae;      /* axion-electron coupling; reference 1e-13 (:640)       */
  double g_agamma;  /* axion-photon coupling in GeV^-1; reference 1e-12 (:641) */
  double g_anuclei; /* axion-nucleon coupling; reference 1e-15 (:643)        */
  uint32_t terms;   /* SART_EM_* bit set; SART_EM_ALL = the reference's total */
  uint32_t _pad;
} sart_emission_params_t;

/* Fills *p with the reference's constants (readOpacityFile.nim:638-643) and SART_EM_ALL. */
void sart_emission_default_params(sart_emission_params_t* p);

/* emRates[R][E] for all cells (row-major [n_radii][n_energies], host memory).
 *   zones        [n_radii]
 *   energies_kev [n_energies]   (the reference: linspace(1e-3, 15, 1500), :612-613)
 *   abs_coefs    [n_radii][n_energies] in keV, or NULL (= 0)
 *   components_out  NULL, or [SART_EM_N_TERMS][n_radii][n_energies]: the single terms in the bit order above
 *                   (all eight are written whatever `terms` selects)
 * Returns SART_OK or a negative SART_ERR_* (sart_last_error() has the text). */
int sart_emission_table(sart_context* ctx, const sart_solar_zone_t* zones, int32_t n_radii,
                        const double* energies_kev, int32_t n_energies, const double* abs_coefs,
                        const sart_emission_params_t* params, double* em_rates_out, double* components_out);

/* Same with device-resident output (and optional device-resident abs_coefs): for callers that keep the table
 * on the GPU.  Runs on the context's stream; returns after the kernel has finished. */
int sart_emission_table_device(sart_context* ctx, const sart_solar_zone_t* zones, int32_t n_radii,
                               const double* energies_kev, int32_t n_energies, const double* abs_coefs_dev,
                               const sart_emission_params_t* params, double* em_rates_dev, double* components_dev);

/* The whole front end of BASELINE configs[4] on the device: emission table of the solar model (as above, kept on the
 * GPU) -> sart_set_solar_tables_device (fluxRadiusCDF, diffFluxCDFs and their guide tables, raytracer.nim:2670-2705).
 * Replaces calculateOpacities -> solar_model_dataframe.csv -> initFullSetup's readCsv + CDF loops for the context's
 * sampling tables; radii = zones[i].radius_frac.  Blocking. */
int sart_emission_to_solar_tables(sart_context* ctx, const sart_solar_zone_t* zones, int32_t n_radii,
                                  const double* energies_kev, int32_t n_energies, const double* abs_coefs_dev,
                                  const sart_emission_params_t* params);

/* ---- absorption coefficients from the OPCD 3.3 monochromatic opacities (readOpacityFile.nim:790-823) ----------------
 *
 * absCoef[R][E] = sum over the metals Z of n_Z[R][Z] * opacity_Z(T_R, n_e,R; E) * 1.97327e-8 * (0.528e-8)^2 * (1 - exp(-E / T_R))
 * for 0.0732 < w = E / T_table(R) < 20, else 0 (:801-808).  opacity_Z is read from the density table `ne_index` of the
 * file `fmZZ.<temp_index>` at the abscissa `table = mesh(w)`, the line number that the mesh file `fm01.mesh` assigns to w
 * (both linear interpolations of numericalnim's newLinear1D, :217, :296, :825, :831).
 *
 * The tables the zones of one solar model need, flattened: one slot per distinct (temp_index, ne_index) pair, one table
 * per slot and element.  sart_host_opcd_load (sart_host.h) builds it from an OPCD directory; all pointers are HOST
 * pointers, the calls below copy what they need to the device. */
typedef struct sart_opacity_tables_t {
  const double* u_mesh;        /* [n_mesh] column `u` of fm01.mesh, strictly ascending; entry i belongs to line number i (:284-296) */
  int32_t n_mesh;              /* 10001 in OPCD 3.3 (:281) */
  int32_t n_slots;             /* distinct (temp_index, ne_index) pairs of the zones */
  const int32_t* slot_of_zone; /* [n_radii] */
  const int32_t* element_z;    /* [n_elements] proton numbers in the order of the reference's sum (ascending, :829-834) */
  int32_t n_elements;          /* the 15 metals 6 7 8 10 11 12 13 14 16 18 20 24 25 26 28 for the reference's element list */
  int32_t _pad;
  const int64_t* table_y_begin; /* [n_slots][n_elements] first opacity of the table in table_y */
  const int64_t* table_x_begin; /* [n_slots][n_elements] first abscissa in table_x, or -1: the abscissae are the line numbers
                                   1 .. len (the 10000-line tables, :206-208) */
  const int32_t* table_len;     /* [n_slots][n_elements] points of the table (>= 2) */
  const double* table_x;        /* pool of abscissae (tables that carry their own first column) */
  const double* table_y;        /* pool of opacities in atomic units (a0^2) */
  int64_t n_table_x, n_table_y; /* pool lengths */
} sart_opacity_tables_t;

/* absCoef[n_radii][n_energies] in keV, as sart_emission_table* take it.
 *   n_z  [n_radii][29]: number densities per proton number in 1/cm^3 (sart_host_solar_number_densities)
 * An abscissa outside a table (numericalnim raises there) fails with SART_ERR_INVALID_ARGUMENT; nothing is clamped. */
int sart_emission_abs_coefs(sart_context* ctx, const sart_solar_zone_t* zones, int32_t n_radii, const double* n_z,
                            const double* energies_kev, int32_t n_energies, const sart_opacity_tables_t* tables,
                            double* abs_coefs_out);
/* Same with the result left on the device (for sart_emission_table_device / sart_emission_to_solar_tables). */
int sart_emission_abs_coefs_device(sart_context* ctx, const sart_solar_zone_t* zones, int32_t n_radii, const double* n_z,
                                   const double* energies_kev, int32_t n_energies, const sart_opacity_tables_t* tables,
                                   double* abs_coefs_dev);

/* The same front end as sart_emission_to_solar_tables with the absorption coefficients of the OPCD files in it:
 * absCoef (above) -> emission table -> fluxRadiusCDF / diffFluxCDFs / guide tables, all on the device.  Blocking. */
int sart_emission_to_solar_tables_opcd(sart_context* ctx, const sart_solar_zone_t* zones, int32_t n_radii, const double* n_z,
                                       const double* energies_kev, int32_t n_energies, const sart_opacity_tables_t* tables,
                                       const sart_emission_params_t* params);

/* Duration in ms of the last absorption-coefficient kernel this process launched (HIP events on the launch stream). */
double sart_emission_abs_coefs_last_kernel_ms(void);

/* Duration in ms of the last emission-table kernel this process launched (HIP events on the launch stream). */
double sart_emission_last_kernel_ms(void);

#ifdef __cplusplus
}
#endif
#endif
