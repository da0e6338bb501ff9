/*
 * sart_emission_oracle.c — CPU oracle of the solar emission-table producer (TEST INFRASTRUCTURE, not product code).
 *
 * Plain-C, f64 restatement of the per-cell arithmetic of `calculateOpacities`
 * (/root/reference/src/readOpacityFile.nim:745-860) and of the functions it calls (:296-468), plus the first loop
 * that derives the per-radius plasma quantities from the solar-model file (:655-705) and the flux spectrum of
 * `getFluxFractionR` (:535-584).  Only tests/ and the emission benchmark's CPU leg may load this library.
 *
 * Third-party arithmetic restated from its published form (parity unpinned at this boundary): numericalnim's
 * `adaptiveGauss` (global adaptive Gauss-Kronrod G10/K21, error estimate |K21 - G10|, worst interval bisected until the
 * summed estimate is below the tolerance 1e-8).  The reference has no test or golden value for any of these functions;
 * tests/test_emission.py pins this file against independent evaluations (scipy quadrature, the literature value of the
 * solar Primakoff flux).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/sart_emission.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ---- constants of calculateOpacities, readOpacityFile.nim:637-651 ---- */
static const double kAlpha = 1.0 / 137.0;
static const double kMeKeV = 510.998;
static const double kAmu = 1.6605e-24;

/* const atomicMass / charges, readOpacityFile.nim:120-132 (order of the model-file columns H1 He4 He3 C12 ... Ni) */
static const double kAtomicMass[29] = {1.0078,  4.0026,  3.0160,  12.0000, 13.0033, 14.0030, 15.0001, 15.9949, 16.9991, 17.9991,
                                       20.1797, 22.9897, 24.3055, 26.9815, 28.085,  30.9737, 32.0675, 35.4515, 39.8775, 39.0983,
                                       40.078,  44.9559, 47.867,  50.9415, 51.9961, 54.9380, 55.845,  58.9331, 58.6934};
static const double kCharges[29] = {1,  2,  2,  6,  6,  7,  7,  8,  8,  8,  10, 11, 12, 13, 14,
                                    15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28};

/* ---- first loop of calculateOpacities, :655-705 ---------------------------------------------------------------- */
/* temp[n], rho[n], mass_fractions[n][29]; zones_out[n].  `temperature` and `n_eInt` keep their previous value when no
 * grid point is within one step (they are variables of the enclosing scope, :622-629) — QUIRK kept. */
void sart_emission_oracle_zones(const double* temp, const double* rho, const double* mass_fractions, int32_t n,
                                sart_solar_zone_t* zones_out) {
  int temperature = 0, n_e_int = 0;
  for (int32_t i = 0; i < n; ++i) {
    const double* e = mass_fractions + (size_t)i * 29;
    sart_solar_zone_t* z = &zones_out[i];
    z->n_H = (e[0] / kAtomicMass[0]) * (rho[i] / kAmu); /* :661 */
    /* :662-667, iZmult = 1: He4 + He3 over their abundance-weighted mean mass */
    z->n_He = (e[1] + e[2]) / ((kAtomicMass[1] * e[1] + kAtomicMass[2] * e[2]) / (e[1] + e[2])) * rho[i] / kAmu;
    double n_e = 0.0; /* :681-683 */
    for (int k = 0; k < 29; ++k) n_e += (rho[i] / kAmu) * kCharges[k] * e[k] / kAtomicMass[k];
    z->n_e = n_e;
    for (int it = 0; it <= 90; ++it) { /* :686-689 */
      const double dist = log10(temp[i]) / 0.025 - (double)(140 + 2 * it);
      if (fabs(dist) <= 1.0) temperature = 140 + 2 * it;
    }
    for (int in = 0; in <= 17; ++in) { /* :692-695 */
      const double dist = log10(n_e) / 0.25 - (double)(74 + in * 2);
      if (fabs(dist) <= 1.0) n_e_int = 74 + in * 2;
    }
    z->temp_index = temperature;
    z->ne_index = n_e_int;
    z->temp_K = temp[i];
    z->rho = rho[i];
    z->radius_frac = (double)i * 0.0005 + 0.0015; /* :698 */
  }
}

/* n_Z[iRadius][.] of the same loop (:655-679): out[n][29], indexed as the reference indexes it (by proton number). */
void sart_emission_oracle_number_densities(const double* rho, const double* mass_fractions, int32_t n, double* out) {
  for (int32_t iRadius = 0; iRadius < n; ++iRadius) {
    const double* e = mass_fractions + (size_t)iRadius * 29;
    double* n_Z = out + (size_t)iRadius * 29;
    memset(n_Z, 0, 29 * sizeof(double)); /* newSeqWith(nRadius, newSeq[float](29)) :619 */
#define N_OF(idx) ((e[idx] / kAtomicMass[idx]) * (rho[iRadius] / kAmu)) /* template n :658-659 */
    n_Z[1] = N_OF(0); /* :661 */
    for (int iZmult = 1; iZmult <= 3; ++iZmult) { /* :662-670 */
      const int iz = iZmult * 2;
      const double nVal = (e[iz - 1] + e[iz]) / ((kAtomicMass[iz - 1] * e[iz - 1] + kAtomicMass[iz] * e[iz]) / (e[iz - 1] + e[iz])) *
                          rho[iRadius] / kAmu;
      if (iZmult == 1) n_Z[iz] = nVal;
      else n_Z[iZmult + 4] = nVal;
    }
    n_Z[8] = (e[7] + e[8] + e[9]) / ((e[7] * kAtomicMass[7] + e[8] * kAtomicMass[8] + e[9] * kAtomicMass[9]) / (e[7] + e[8] + e[9])) *
             rho[iRadius] / kAmu; /* :672-675 */
    for (int iZ = 10; iZ < 29; ++iZ) n_Z[iZ] = N_OF(iZ); /* :676-677 */
#undef N_OF
  }
}

/* ---- absorption coefficients, :790-823 ----------------------------------------------------------------------------
 * numericalnim's Linear1D as the reference uses it (newLinear1D :217, :296; eval :825, :831): the interval by binary search,
 * y0 + (x - x0) (y1 - y0) / (x1 - x0); the library raises outside [x0, xN] -> *outside is set. */
static double linear1d_strict(const double* xs, const double* ys, int64_t n, double x, int* outside) {
  if (!(x >= xs[0]) || !(x <= xs[n - 1])) {
    *outside = 1;
    return 0.0;
  }
  int64_t lo = 0, hi = n - 1;
  while (hi - lo > 1) {
    const int64_t mid = (lo + hi) / 2;
    if (xs[mid] <= x) lo = mid; else hi = mid;
  }
  return ys[lo] + (x - xs[lo]) * (ys[lo + 1] - ys[lo]) / (xs[lo + 1] - xs[lo]);
}

/* absCoefs[R, iEindex] for all cells.  `t` holds what the reference keeps in `spline` (mesh -> line number) and in
 * opElements[temperature][Z].densityTab[n_eInt] for the (temperature, density) slot of every zone; the loop over the
 * elements is the reference's: all of ElementKind (:28-52) without noElement (:636), every one evaluated, Z > 2 summed.
 * Hydrogen and helium carry no table here (their value is discarded, :833).  Returns the number of cells whose
 * evaluation leaves a table (the reference raises at the first one); those cells are NaN. */
int64_t sart_emission_oracle_abs_coefs(const sart_solar_zone_t* zones, int32_t n_radii, const double* n_z, const double* energies_kev,
                                       int32_t n_energies, const sart_opacity_tables_t* t, double* out) {
  static const int element_kinds[] = {1, 2, 6, 7, 8, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28};
  static const int no_element[] = {3, 4, 5, 9, 15, 17, 19, 21, 22, 23, 27};
  int64_t n_outside = 0;
  double* lines = (double*)malloc((size_t)t->n_mesh * sizeof(double)); /* linspace(0.0, 10000.0, 10001) :286 */
  double* line_numbers = (double*)malloc(10000 * sizeof(double));      /* energy = float lineCnt + 1 :207 */
  for (int32_t i = 0; i < t->n_mesh; ++i) lines[i] = (double)i;
  for (int i = 0; i < 10000; ++i) line_numbers[i] = (double)(i + 1);
  for (int32_t R = 0; R < n_radii; ++R) {
    const sart_solar_zone_t* z = &zones[R];
    const double temp = (double)z->temp_index;
    const double temp_keVTable = pow(10.0, (temp * 0.025)) * 8.617e-8; /* :758 */
    const double temp_keV = z->temp_K * 8.617e-8;                      /* :759-760 */
    const double* n_Z = n_z + (size_t)R * 29;
    const size_t row = (size_t)t->slot_of_zone[R] * (size_t)t->n_elements;
    for (int32_t iE = 0; iE < n_energies; ++iE) {
      const double energy_keV = energies_kev[iE];
      double sum = 0.0;
      const double w = energy_keV / temp_keVTable; /* :793 */
      double absCoef;
      if (w >= 20.0 || w <= 0.0732) { /* :801-808 */
        for (size_t k = 0; k < sizeof(element_kinds) / sizeof(int); ++k) {
          const int Z = element_kinds[k];
          int skip = 0;
          for (size_t j = 0; j < sizeof(no_element) / sizeof(int); ++j) skip |= (no_element[j] == Z);
          if (skip) continue;
          sum = sum + n_Z[Z] * 0.0;
        }
        absCoef = sum * 1.97327e-8 * 0.528e-8 * 0.528e-8 * (1.0 - exp(-energy_keV / temp_keV));
      } else {
        int outside = 0;
        const double table = linear1d_strict(t->u_mesh, lines, t->n_mesh, w, &outside); /* :825 */
        for (size_t k = 0; k < sizeof(element_kinds) / sizeof(int) && !outside; ++k) {
          const int Z = element_kinds[k];
          int skip = 0;
          for (size_t j = 0; j < sizeof(no_element) / sizeof(int); ++j) skip |= (no_element[j] == Z);
          if (skip) continue;
          if (!(Z > 2)) continue; /* evaluated and dropped in the reference (:831-833) */
          int col = -1;
          for (int32_t c = 0; c < t->n_elements; ++c)
            if (t->element_z[c] == Z) col = c;
          if (col < 0) { outside = 1; break; } /* KeyError of :831 */
          const int64_t yb = t->table_y_begin[row + (size_t)col], xb = t->table_x_begin[row + (size_t)col];
          const int64_t len = t->table_len[row + (size_t)col];
          const double* xs = xb < 0 ? line_numbers : t->table_x + xb;
          if (xb < 0 && len != 10000) { outside = 1; break; }
          const double opacity = linear1d_strict(xs, t->table_y + yb, len, table, &outside);
          sum += n_Z[Z] * opacity; /* :834 */
        }
        absCoef = sum * 1.97327e-8 * 0.528e-8 * 0.528e-8 * (1.0 - exp(-energy_keV / temp_keV)); /* :838 */
        if (outside) {
          absCoef = NAN;
          ++n_outside;
        }
      }
      out[(size_t)R * n_energies + iE] = absCoef;
    }
  }
  free(lines);
  free(line_numbers);
  return n_outside;
}

/* ---- fNew and its integrand, :296-326 --------------------------------------------------------------------------- */
static double inner_integral(double t, double y) { /* :296-297 */
  return (1.0 / 2.0) * (((y * y) / (t * t + y * y)) + log(t * t + y * y));
}
static double outer(double x, double w, double y) { /* :299-310 */
  const double coeff = x * exp(-x * x);
  const double frm = sqrt(x * x + w) - x;
  const double to = sqrt(x * x + w) + x;
  return coeff * (inner_integral(to, y) - inner_integral(frm, y));
}
static double fn_to_int(double t, double w, double y) { /* :319-324 */
  if (t != 0.0) return outer((1.0 - t) / t, w, y) / (t * t);
  return outer((1.0 - t) / (t + 1e-8), w, y) / (t * t);
}

/* Gauss-Kronrod 21 / Gauss 10 nodes and weights on [-1, 1] (QUADPACK qk21). */
static const double kXgk[11] = {0.995657163025808080735527280689003, 0.973906528517171720077964012084452,
                                0.930157491355708226001207180059508, 0.865063366688984510732096688423493,
                                0.780817726586416897063717578345042, 0.679409568299024406234327365114874,
                                0.562757134668604683339000099272694, 0.433395394129247190799265943165784,
                                0.294392862701460198131126603103866, 0.148874338981631210884826001129720,
                                0.000000000000000000000000000000000};
static const double kWgk[11] = {0.011694638867371874278064396062192, 0.032558162307964727478818972459390,
                                0.054755896574351996031381300244580, 0.075039674810919952767043140916190,
                                0.093125454583697605535065465083366, 0.109387158802297641899210590325805,
                                0.123491976262065851077958109585166, 0.134709217311473325928054001771707,
                                0.142775938577060080797094273138717, 0.147739104901338491374841515972068,
                                0.149445554002916905664936468389821};
static const double kWg[5] = {0.066671344308688137593568809893332, 0.149451349150580593145776339657697,
                              0.219086362515982043995534934228163, 0.269266719309996355091226921569469,
                              0.295524224714752870173815619188769};

static void gk21(double a, double b, double w, double y, double* integral, double* err) {
  const double c = 0.5 * (a + b), h = 0.5 * (b - a);
  const double fc = fn_to_int(c, w, y);
  double rk = kWgk[10] * fc, rg = 0.0;
  for (int j = 0; j < 5; ++j) { /* odd Kronrod indices are the Gauss nodes */
    const double dx = h * kXgk[2 * j + 1];
    const double f1 = fn_to_int(c - dx, w, y), f2 = fn_to_int(c + dx, w, y);
    rg += kWg[j] * (f1 + f2);
    rk += kWgk[2 * j + 1] * (f1 + f2);
  }
  for (int j = 0; j < 5; ++j) {
    const double dx = h * kXgk[2 * j];
    rk += kWgk[2 * j] * (fn_to_int(c - dx, w, y) + fn_to_int(c + dx, w, y));
  }
  *integral = rk * h;
  *err = fabs((rk - rg) * h);
}

/* fNew (:312-326): adaptiveGauss(fnToInt, 0.0, 1.0), default tolerance 1e-8 */
double sart_emission_oracle_fnew(double w, double y) {
  enum { kMaxIntervals = 10000 };
  static __thread double lo[kMaxIntervals], hi[kMaxIntervals], val[kMaxIntervals], er[kMaxIntervals];
  int n = 1;
  lo[0] = 0.0;
  hi[0] = 1.0;
  gk21(0.0, 1.0, w, y, &val[0], &er[0]);
  double total_err = er[0];
  while (total_err > 1e-8 && n < kMaxIntervals - 1) {
    int worst = 0;
    for (int i = 1; i < n; ++i)
      if (er[i] > er[worst]) worst = i;
    const double a = lo[worst], b = hi[worst], m = 0.5 * (a + b);
    gk21(a, m, w, y, &val[worst], &er[worst]);
    hi[worst] = m;
    lo[n] = m;
    hi[n] = b;
    gk21(m, b, w, y, &val[n], &er[n]);
    ++n;
    total_err = 0.0;
    for (int i = 0; i < n; ++i) total_err += er[i];
  }
  double total = 0.0;
  for (int i = 0; i < n; ++i) total += val[i];
  return total;
}

/* ---- emission-rate terms, :328-468 ------------------------------------------------------------------------------- */
static double bfield(double r) { /* :328-352 */
  const double radius_cz = 0.712, size_tach = 0.02, radius_outer = 0.96, size_outer = 0.035;
  const double bfield_rad_T = 3.0e3, bfield_tach_T = 50.0, bfield_outer_T = 4.0;
  const double lambda1 = 10.0 * radius_cz + 1.0;
  const double lambda_factor = (1.0 + lambda1) * pow(1.0 + 1.0 / lambda1, lambda1);
  double b = 0.0;
  if (r < (radius_cz + size_tach)) {
    const double x = pow(r / radius_cz, 2.0);
    if (x < 1.0) b = bfield_rad_T * lambda_factor * x * pow(1.0 - x, lambda1);
    const double y = pow(((r - radius_cz) / size_tach), 2.0);
    if (y < 1.0) b = bfield_tach_T * (1.0 - y);
  } else {
    const double z = pow((r - radius_outer) / size_outer, 2.0);
    if (z < 1.0) b = bfield_outer_T * (1.0 - z);
    else b = 0.0;
  }
  return b / (1.0e6 * 1.4440271 * 1.0e-3 * sqrt(4.0 * M_PI));
}
double sart_emission_oracle_bfield(double r) { return bfield(r); }

static double omega_plasmon_sq(double alpha, double ne, double me) { return 4.0 * alpha * M_PI * ne / me; } /* :356-357 */

static double compton_emrate(double alpha, double gae, double energy, double ne, double me, double temp) { /* :360-362 */
  return (alpha * gae * gae * energy * energy * ne) / (3.0 * pow(me, 4) * (exp(energy / temp) - 1.0));
}
static double brems_emrate(double alpha, double gae, double energy, double ne, double me, double temp, double w, double y) { /* :364-367 */
  return (alpha * alpha * gae * gae * 4.0 * sqrt(M_PI) * ne * ne * exp(-energy / temp) * sart_emission_oracle_fnew(w, sqrt(2.0) * y)) /
         (3.0 * sqrt(temp) * pow(me, 3.5) * energy);
}
static double term1(double gae, double energy, double abscoef, double echarge, double me, double temp) { /* :369-371 */
  return (gae * gae * energy * energy * abscoef) / (2.0 * echarge * echarge * me * me * (exp(energy / temp) - 1.0));
}
static double freefree_emrate(double alpha, double gae, double energy, double ne, double me, double temp, double nzZ2, double w, double y) { /* :378-381 */
  return (sart_emission_oracle_fnew(w, y) * alpha * alpha * gae * gae * 8.0 * sqrt(M_PI) * ne * nzZ2 * exp(-energy / temp)) /
         (3.0 * sqrt(2.0 * temp) * pow(me, 3.5) * energy);
}
static double primakoff_bracket(double t, double u) { /* :384-392 */
  double a = 0.0;
  if (u > 1.0) a += (u * u - 1.0) * log((u - 1.0) / (u + 1.0));
  const double v = u + t;
  if (v > 1.0) a -= (v * v - 1.0) * log((v - 1.0) / (v + 1.0));
  a *= 0.5 / t;
  a -= 1.0;
  return a;
}
static double primakoff(double temp, double energy, double gagamma, double ks2, double alpha, double ne, double me, double n_Z2, double n_Z1) { /* :394-418 */
  const double prefactor6 = gagamma * gagamma * 1e-12 * alpha / 8.0;
  const double omPlSq = omega_plasmon_sq(alpha, ne, me);
  const double z = energy / temp;
  const double om2 = energy * energy;
  const double x = om2 / omPlSq;
  if (x < 1.0 || energy == 0.0) return 0.0;
  const double phase_factor = 2.0 / (sqrt(1.0 - 1.0 / x) * (exp(z) - 1.0));
  const double n_dens = ne + n_Z1 * 7.645e-24 + 4.0 * n_Z2 * 7.645e-24;
  const double s = 2.0 * energy * sqrt(om2 - omPlSq);
  const double t = ks2 / s;
  const double u = (2.0 * om2 - omPlSq) / s;
  return prefactor6 * phase_factor * n_dens * primakoff_bracket(t, u);
}
static double long_plasmon(double energy, double ne, double me, double alpha, double bfieldR, double temp, double opacity, double gagamma) { /* :420-437 */
  const double omPlSq = omega_plasmon_sq(alpha, ne, me);
  const double prefactor = gagamma * gagamma * 1e-12;
  const double om2 = energy * energy;
  const double z = energy / temp;
  double gammaL = (1.0 - exp(-z)) * opacity;
  gammaL = fmax(gammaL, 1e-4);
  const double xi2 = gammaL * energy;
  const double fwhm = sqrt(om2 + xi2) - sqrt(om2 - xi2);
  if (fabs(energy - sqrt(omPlSq)) > 18.0 * fwhm) return 0.0;
  const double average_bfield_sq = bfieldR * bfieldR / 3.0;
  const double fraction = energy * xi2 / (pow(om2 - omPlSq, 2.0) + xi2 * xi2);
  return prefactor * average_bfield_sq * fraction / (exp(z) - 1.0);
}
static double trans_plasmon(double energy, double ne, double me, double alpha, double bfieldR, double temp, double opacity, double gagamma) { /* :439-453 */
  const double geom_factor = 1.0, photon_polarization = 2.0;
  const double omPlSq = omega_plasmon_sq(alpha, ne, me);
  if (omPlSq > energy * energy) return 0.0;
  const double u = energy / temp;
  const double gamma = (1.0 - exp(-u)) * opacity;
  const double deltaPsq = energy * energy * pow(sqrt(1.0 - omPlSq / (energy * energy)) - 1.0, 2.0);
  const double average_b_field_sq = pow(bfieldR, 2.0) / 3.0;
  const double deltaTsq = gagamma * gagamma * 1e-12 * average_b_field_sq / 4.0;
  return geom_factor * photon_polarization * gamma * deltaTsq / ((deltaPsq + pow(0.5 * gamma, 2.0)) * (exp(u) - 1.0));
}
static double iron(double ganuclei, double temp, double energy, double rho) { /* :455-468 */
  const double tau_gamma = 1.3e-6 * 1.519e18;
  const double n = 3.0e17 * 1.7826e-30;
  const double e_gamma = 14.4;
  const double m_Fe = 56.9353928 * 1.6605e-24 * 5.60958616722e29;
  const double u = e_gamma / temp;
  const double w_1 = 4.0 * exp(-u) / (2.0 + 4.0 * exp(-u));
  const double gamma_frac = 1.82 * ganuclei * ganuclei;
  const double sigma = e_gamma * sqrt(temp / m_Fe);
  const double n_a = n * w_1 * gamma_frac / tau_gamma;
  return n_a * exp(-pow(energy - e_gamma, 2.0) / (2.0 * sigma * sigma)) * rho * sqrt(2.0 * M_PI) * M_PI / (sigma * energy * energy);
}

/* ---- the cell loop, :745-860 ------------------------------------------------------------------------------------ */
/* components_out: NULL or [8][n_radii][n_energies] in the bit order of SART_EM_*.  Radii r_stride apart and energies
 * e_stride apart are evaluated (the others are left untouched) so that tests can check a sub-grid in seconds. */
int sart_emission_oracle_table(const sart_solar_zone_t* zones, int32_t n_radii, const double* energies_kev, int32_t n_energies,
                               const double* abs_coefs, const sart_emission_params_t* params, double* em_rates_out,
                               double* components_out, int32_t r_stride, int32_t e_stride, int32_t n_threads) {
  if (!zones || !energies_kev || !params || !em_rates_out || n_radii < 1 || n_energies < 1) return -1;
  if (r_stride < 1) r_stride = 1;
  if (e_stride < 1) e_stride = 1;
  const double e_charge = sqrt(4.0 * M_PI * kAlpha); /* :645 */
  const size_t plane = (size_t)n_radii * n_energies;
  (void)n_threads;
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads > 0 ? n_threads : 1)
  for (int32_t R = 0; R < n_radii; R += r_stride) {
    const sart_solar_zone_t* z = &zones[R];
    const double n_e_keV = z->n_e * 7.683e-24; /* :749, :756 */
    const double temp = (double)z->temp_index;
    const double radius = 0.0015 + (double)R * 0.0005; /* :751 */
    const double bfieldR = bfield(radius);
    const double rho_keV = z->rho * 7.683e-24 * 5.60958616722e29; /* :753 */
    const double temp_keVTable = pow(10.0, (temp * 0.025)) * 8.617e-8; /* :758 */
    const double temp_keV = z->temp_K * 8.617e-8;                      /* :759-760 */
    const double debye_scale_squared = (4.0 * M_PI * kAlpha / temp_keV) * (n_e_keV + z->n_H * 7.645e-24 + 4.0 * z->n_He * 7.645e-24); /* :763-765 */
    const double debye_scale = sqrt(debye_scale_squared);
    const double y = debye_scale / (sqrt(2.0 * kMeKeV * temp_keV)); /* :767 */
    for (int32_t iE = 0; iE < n_energies; iE += e_stride) {
      const double energy_keV = energies_kev[iE];
      const double w = energy_keV / temp_keVTable; /* :774 */
      const size_t cell = (size_t)R * n_energies + iE;
      const double absCoef = abs_coefs ? abs_coefs[cell] : 0.0;
      const double nZZ2 = (z->rho / kAmu) * 7.683e-24; /* :826 */
      const double c[8] = {
          compton_emrate(kAlpha, params->g_ae, energy_keV, n_e_keV, kMeKeV, temp_keV),
          term1(params->g_ae, energy_keV, absCoef, e_charge, kMeKeV, temp_keV),
          brems_emrate(kAlpha, params->g_ae, energy_keV, n_e_keV, kMeKeV, temp_keV, w, y),
          freefree_emrate(kAlpha, params->g_ae, energy_keV, n_e_keV, kMeKeV, temp_keV, nZZ2, w, y),
          primakoff(temp_keV, energy_keV, params->g_agamma, debye_scale_squared, kAlpha, n_e_keV, kMeKeV, z->n_He, z->n_H),
          long_plasmon(energy_keV, n_e_keV, kMeKeV, kAlpha, bfieldR, temp_keV, absCoef, params->g_agamma),
          trans_plasmon(energy_keV, n_e_keV, kMeKeV, kAlpha, bfieldR, temp_keV, absCoef, params->g_agamma),
          iron(params->g_anuclei, temp_keV, energy_keV, rho_keV)};
      /* :849  compton + term1 + term3 + ffterm + transPlas + primakoff + longPlas + iron57 (same order of additions) */
      double total = 0.0;
      const int order[8] = {0, 1, 2, 3, 6, 4, 5, 7};
      for (int k = 0; k < 8; ++k)
        if (params->terms & (1u << order[k])) total += c[order[k]];
      em_rates_out[cell] = total;
      if (components_out)
        for (int k = 0; k < 8; ++k) components_out[(size_t)k * plane + cell] = c[k];
    }
  }
  return 0;
}

/* ---- getFluxFractionR, :535-584: flux spectrum in 1/(keV y m^2) from an emission table ------------------------- */
void sart_emission_oracle_flux_spectrum(const double* em_rates, int32_t n_radii, const double* energies_kev, int32_t n_energies,
                                        double* diff_flux_out) {
  const double r_sun = 6.957e11, r_sunearth = 1.5e14, hbar = 6.582119514e-25, keV2cm = 1.97327e-8;
  const double factor = pow(r_sun * 0.1 / (keV2cm), 3.0) / (pow(0.1 * r_sunearth, 2.0) * (1.0e6 * hbar)) / (3.1709791983765E-8 * 1.0e-4);
  for (int32_t idx = 0; idx < n_energies; ++idx) {
    const double e_keV = energies_kev[idx];
    double diff_flux_r = 0.0, r_last = 0.0;
    for (int32_t r = 0; r < n_radii; ++r) {
      const double r_perc = ((double)r * 0.0005 + 0.0015);
      const double diff_flux = em_rates[(size_t)r * n_energies + idx] * (r_perc - r_last) * r_perc * r_perc * e_keV * e_keV * 0.5 / (M_PI * M_PI);
      diff_flux_r += diff_flux;
      r_last = r_perc;
    }
    diff_flux_out[idx] = diff_flux_r * factor;
  }
}
