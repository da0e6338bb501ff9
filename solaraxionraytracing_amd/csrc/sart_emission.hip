// sart_emission.hip — solar emission-table producer on the GPU (include/sart_emission.h).
//
// One thread per (radius, energy) cell of `calculateOpacities`' second loop (readOpacityFile.nim:745-860): the eight
// emission-rate terms (:360-468) in f64, summed in the reference's order (:849).  The two terms that hold the integral
// `fNew(w, y)` (:312-326, free-free with y and ee-bremsstrahlung with sqrt(2) y) share one fixed 80-node composite
// Gauss-Legendre rule in x (5 panels graded towards x = 0, 16 nodes each; relative error ~1e-14 over the whole (w, y)
// range of the solar model), instead of the reference's adaptive Gauss-Kronrod with tolerance 1e-8.  The kernel is
// f64-VALU bound (≈100 exp/log/sqrt/div evaluations per cell, 24 B of output): no LDS, no MFMA.
//
// There is no CPU fallback: without a context (= without a GPU) nothing here runs.
#include <hip/hip_runtime.h>

#include <cmath>
#include <string>
#include <vector>

#include "../../include/sart_emission.h"
#include "sart_math.h"

namespace sart {
int context_device(sart_context* c);
hipStream_t context_stream(sart_context* c);
int set_error(int code, const std::string& msg);
}  // namespace sart

namespace {

constexpr double kPi = 3.14159265358979323846;
constexpr double kAlpha = 1.0 / 137.0;  // readOpacityFile.nim:639
constexpr double kMe = 510.998;         // :644
constexpr double kAmu = 1.6605e-24;     // :651
constexpr int kPanels = 5, kOrder = 16, kNodes = kPanels * kOrder;

// Per-radius constants of the cell loop (:748-767), computed once on the host.
struct EmZone {
  double n_e;        // keV^3 (:749)
  double temp;       // keV (:759)
  double inv_temp_table;  // 1 / (10^(0.025 temp_index) * 8.617e-8) (:758): w = E * this (:774)
  double ks2;        // debye_scale_squared (:763-765)
  double y;          // :767
  double bfield;     // bfield(radius) in keV^2 (:328-352, :752)
  double rho;        // keV^4 (:753)
  double nzz2;       // (rho / amu) * 7.683e-24 (:826)
  double n_dens;     // ne + n_H 7.645e-24 + 4 n_He 7.645e-24 (primakoff :410)
  double om_pl_sq;   // omegaPlasmonSq (:356-357)
};

struct EmArgs {
  const EmZone* zones;
  const double* energies;
  const double* abs_coefs;  // may be null
  double* out;
  double* components;  // may be null
  int n_radii, n_energies;
  unsigned terms;
  double g_ae, g_agamma, g_anuclei;
};

__constant__ double c_node_x[kNodes];  // quadrature nodes x_k
__constant__ double c_node_w[kNodes];  // weights * x_k * exp(-x_k^2): the `coeff` of outer() (:300) folded in

// fNew(w, y) and fNew(w, sqrt(2) y) in one pass.  outer(x) = x e^{-x^2} [I(to) - I(frm)], I(t) = (y^2/(t^2+y^2) + ln(t^2+y^2))/2
// (:296-310), to = sqrt(x^2+w) + x, frm = sqrt(x^2+w) - x = w / to.
__device__ __forceinline__ void fnew_pair(double w, double y, double& f_y, double& f_y_sqrt2) {
  const double ya = y * y, yb = 2.0 * ya;
  double sa = 0.0, sb = 0.0;
#pragma unroll 4
  for (int k = 0; k < kNodes; ++k) {
    // square root and reciprocals from the hardware seeds + one third-order step (sart_math.h, <= 1 ulp) instead of the
    // IEEE-exact expansions; the two reciprocals of every (hi, lo) pair come from ONE reciprocal of their product
    const double x = c_node_x[k];
    const double s = sart::fsqrt_pos(fma(x, x, w));
    const double to = s + x;
    const double frm = w * sart::frcp(to);
    const double to2 = to * to, fr2 = frm * frm;
    const double a_hi = to2 + ya, a_lo = fr2 + ya;
    const double b_hi = to2 + yb, b_lo = fr2 + yb;
    const double ra = sart::frcp(a_hi * a_lo), rb = sart::frcp(b_hi * b_lo);   // 1 / a_hi = ra a_lo, 1 / a_lo = ra a_hi
    const double ga = fma(ya * ra, a_lo - a_hi, log(a_hi * a_hi * ra));
    const double gb = fma(yb * rb, b_lo - b_hi, log(b_hi * b_hi * rb));
    sa = fma(c_node_w[k], ga, sa);
    sb = fma(c_node_w[k], gb, sb);
  }
  f_y = 0.5 * sa;
  f_y_sqrt2 = 0.5 * sb;
}

__global__ __launch_bounds__(256) void emission_table_kernel(EmArgs A) {
  const int iE = blockIdx.x * blockDim.x + threadIdx.x;
  const int R = blockIdx.y;
  if (iE >= A.n_energies) return;
  const EmZone Z = A.zones[R];
  const double energy = A.energies[iE];
  const size_t cell = (size_t)R * A.n_energies + iE;
  const double abs_coef = A.abs_coefs ? A.abs_coefs[cell] : 0.0;
  const double temp = Z.temp, ne = Z.n_e;
  const double z = energy / temp;
  const double ez = exp(z), emz = exp(-z);
  const double bose = ez - 1.0;  // (exp(energy / temp) - 1.0), as the reference writes it
  const double w = energy * Z.inv_temp_table;
  const double me2 = kMe * kMe, me35 = me2 * kMe * sqrt(kMe);

  double f_y, f_y2;
  fnew_pair(w, Z.y, f_y, f_y2);

  double c[SART_EM_N_TERMS];
  // comptonEmrate :360-362
  c[0] = (kAlpha * A.g_ae * A.g_ae * energy * energy * ne) / (3.0 * me2 * me2 * bose);
  // term1 :369-371, e_charge^2 = 4 pi alpha (:645)
  c[1] = (A.g_ae * A.g_ae * energy * energy * abs_coef) / (2.0 * (4.0 * kPi * kAlpha) * me2 * bose);
  // bremsEmrate :364-367
  c[2] = (kAlpha * kAlpha * A.g_ae * A.g_ae * 4.0 * sqrt(kPi) * ne * ne * emz * f_y2) / (3.0 * sqrt(temp) * me35 * energy);
  // freefreeEmrate :378-381
  c[3] = (f_y * kAlpha * kAlpha * A.g_ae * A.g_ae * 8.0 * sqrt(kPi) * ne * Z.nzz2 * emz) / (3.0 * sqrt(2.0 * temp) * me35 * energy);
  // primakoff :394-418
  {
    const double om2 = energy * energy;
    const double x = om2 / Z.om_pl_sq;
    double r = 0.0;
    if (!(x < 1.0 || energy == 0.0)) {
      const double phase_factor = 2.0 / (sqrt(1.0 - 1.0 / x) * bose);
      const double s = 2.0 * energy * sqrt(om2 - Z.om_pl_sq);
      const double t = Z.ks2 / s;
      const double u = (2.0 * om2 - Z.om_pl_sq) / s;
      double a = 0.0;  // primakoff_bracket :384-392
      if (u > 1.0) a += (u * u - 1.0) * log((u - 1.0) / (u + 1.0));
      const double v = u + t;
      if (v > 1.0) a -= (v * v - 1.0) * log((v - 1.0) / (v + 1.0));
      a *= 0.5 / t;
      a -= 1.0;
      r = (A.g_agamma * A.g_agamma * 1e-12 * kAlpha / 8.0) * phase_factor * Z.n_dens * a;
    }
    c[4] = r;
  }
  // longPlasmon :420-437
  {
    const double om2 = energy * energy;
    double gamma_l = (1.0 - emz) * abs_coef;
    gamma_l = fmax(gamma_l, 1e-4);
    const double xi2 = gamma_l * energy;
    const double fwhm = sqrt(om2 + xi2) - sqrt(om2 - xi2);
    double r = 0.0;
    if (!(fabs(energy - sqrt(Z.om_pl_sq)) > 18.0 * fwhm)) {
      const double d = om2 - Z.om_pl_sq;
      const double fraction = energy * xi2 / (d * d + xi2 * xi2);
      r = (A.g_agamma * A.g_agamma * 1e-12) * (Z.bfield * Z.bfield / 3.0) * fraction / bose;
    }
    c[5] = r;
  }
  // transPlasmon :439-453
  {
    double r = 0.0;
    if (!(Z.om_pl_sq > energy * energy)) {
      const double gamma = (1.0 - emz) * abs_coef;
      const double q = sqrt(1.0 - Z.om_pl_sq / (energy * energy)) - 1.0;
      const double delta_p_sq = energy * energy * (q * q);
      const double delta_t_sq = A.g_agamma * A.g_agamma * 1e-12 * (Z.bfield * Z.bfield / 3.0) / 4.0;
      r = 1.0 * 2.0 * gamma * delta_t_sq / ((delta_p_sq + (0.5 * gamma) * (0.5 * gamma)) * bose);
    }
    c[6] = r;
  }
  // iron :455-468
  {
    const double tau_gamma = 1.3e-6 * 1.519e18;
    const double n = 3.0e17 * 1.7826e-30;
    const double e_gamma = 14.4;
    const double m_fe = 56.9353928 * 1.6605e-24 * 5.60958616722e29;
    const double eu = exp(-(e_gamma / temp));
    const double w_1 = 4.0 * eu / (2.0 + 4.0 * eu);
    const double sigma = e_gamma * sqrt(temp / m_fe);
    const double n_a = n * w_1 * (1.82 * A.g_anuclei * A.g_anuclei) / tau_gamma;
    const double d = energy - e_gamma;
    c[7] = n_a * exp(-(d * d) / (2.0 * sigma * sigma)) * Z.rho * sqrt(2.0 * kPi) * kPi / (sigma * energy * energy);
  }

  // total_emrate :849: compton + term1 + term3 + ffterm + transPlas + primakoff + longPlas + iron57
  const int order[SART_EM_N_TERMS] = {0, 1, 2, 3, 6, 4, 5, 7};
  double total = 0.0;
#pragma unroll
  for (int k = 0; k < SART_EM_N_TERMS; ++k)
    if (A.terms & (1u << order[k])) total += c[order[k]];
  A.out[cell] = total;
  if (A.components) {
    const size_t plane = (size_t)A.n_radii * A.n_energies;
#pragma unroll
    for (int k = 0; k < SART_EM_N_TERMS; ++k) A.components[(size_t)k * plane + cell] = c[k];
  }
}

// bfield, readOpacityFile.nim:328-352 (r in fractions of the solar radius; result in keV^2)
double bfield_of(double r) {
  const double radius_cz = 0.712, size_tach = 0.02, radius_outer = 0.96, size_outer = 0.035;
  const double bfield_rad_T = 3.0e3, bfield_tach_T = 50.0, bfield_outer_T = 4.0;
  const double lambda1 = 10.0 * radius_cz + 1.0;
  const double lambda_factor = (1.0 + lambda1) * std::pow(1.0 + 1.0 / lambda1, lambda1);
  double b = 0.0;
  if (r < (radius_cz + size_tach)) {
    const double x = std::pow(r / radius_cz, 2.0);
    if (x < 1.0) b = bfield_rad_T * lambda_factor * x * std::pow(1.0 - x, lambda1);
    const double y = std::pow((r - radius_cz) / size_tach, 2.0);
    if (y < 1.0) b = bfield_tach_T * (1.0 - y);
  } else {
    const double z = std::pow((r - radius_outer) / size_outer, 2.0);
    b = (z < 1.0) ? bfield_outer_T * (1.0 - z) : 0.0;
  }
  return b / (1.0e6 * 1.4440271 * 1.0e-3 * std::sqrt(4.0 * kPi));
}

// 16-point Gauss-Legendre rule on [-1, 1] (positive half; symmetric).
const double kGlX[8] = {0.0950125098376374401853193, 0.2816035507792589132304605, 0.4580167776572273863424194,
                        0.6178762444026437484466718, 0.7554044083550030338951012, 0.8656312023878317438804679,
                        0.9445750230732325760779884, 0.9894009349916499325961542};
const double kGlW[8] = {0.1894506104550684962853967, 0.1826034150449235888667637, 0.1691565193950025381893121,
                        0.1495959888165767320815017, 0.1246289712555338720524763, 0.0951585116824927848099251,
                        0.0622535239386478928628438, 0.0271524594117540948517806};
const double kPanelEdges[kPanels + 1] = {0.0, 0.05, 0.3, 1.0, 2.2, 6.0};  // e^{-36} = 2e-16 beyond

double g_last_kernel_ms = 0.0;

#define EM_HIP(call)                                                                                                  \
  do {                                                                                                                \
    hipError_t e_ = (call);                                                                                           \
    if (e_ != hipSuccess)                                                                                             \
      return sart::set_error(e_ == hipErrorOutOfMemory ? SART_ERR_OUT_OF_MEMORY : SART_ERR_NO_DEVICE,               \
                             std::string(#call) + ": " + hipGetErrorString(e_));                                      \
  } while (0)

struct Scoped {  // frees device scratch on every exit path
  void* p = nullptr;
  ~Scoped() { if (p) (void)hipFree(p); }
};

int upload_nodes() {
  double xs[kNodes], ws[kNodes];
  int k = 0;
  for (int p = 0; p < kPanels; ++p) {
    const double a = kPanelEdges[p], b = kPanelEdges[p + 1], c = 0.5 * (a + b), h = 0.5 * (b - a);
    for (int j = 7; j >= 0; --j, ++k) { xs[k] = c - h * kGlX[j]; ws[k] = h * kGlW[j]; }
    for (int j = 0; j < 8; ++j, ++k) { xs[k] = c + h * kGlX[j]; ws[k] = h * kGlW[j]; }
  }
  for (int i = 0; i < kNodes; ++i) ws[i] *= xs[i] * std::exp(-xs[i] * xs[i]);
  EM_HIP(hipMemcpyToSymbol(HIP_SYMBOL(c_node_x), xs, sizeof(xs)));
  EM_HIP(hipMemcpyToSymbol(HIP_SYMBOL(c_node_w), ws, sizeof(ws)));
  return 0;
}

int run(sart_context* ctx, const sart_solar_zone_t* zones, int32_t n_radii, const double* energies, int32_t n_energies,
        const double* abs_coefs, bool abs_on_device, const sart_emission_params_t* params, double* out_dev, double* comp_dev) {
  std::vector<EmZone> hz((size_t)n_radii);
  for (int32_t R = 0; R < n_radii; ++R) {
    const sart_solar_zone_t& z = zones[R];
    if (!(z.temp_K > 0.0) || !(z.n_e > 0.0) || !(z.rho > 0.0) || z.temp_index <= 0)
      return sart::set_error(SART_ERR_INVALID_ARGUMENT, "solar zone " + std::to_string(R) + ": temperature, density and electron density must be positive");
    EmZone& e = hz[R];
    e.n_e = z.n_e * 7.683e-24;
    e.temp = z.temp_K * 8.617e-8;
    e.inv_temp_table = 1.0 / (std::pow(10.0, ((double)z.temp_index * 0.025)) * 8.617e-8);
    e.ks2 = (4.0 * kPi * kAlpha / e.temp) * (e.n_e + z.n_H * 7.645e-24 + 4.0 * z.n_He * 7.645e-24);
    e.y = std::sqrt(e.ks2) / (std::sqrt(2.0 * kMe * e.temp));
    e.bfield = bfield_of(0.0015 + (double)R * 0.0005);
    e.rho = z.rho * 7.683e-24 * 5.60958616722e29;
    e.nzz2 = (z.rho / kAmu) * 7.683e-24;
    e.n_dens = e.n_e + z.n_H * 7.645e-24 + 4.0 * z.n_He * 7.645e-24;
    e.om_pl_sq = 4.0 * kAlpha * kPi * e.n_e / kMe;
  }
  for (int32_t i = 0; i < n_energies; ++i)
    if (!(energies[i] > 0.0) || !std::isfinite(energies[i]))
      return sart::set_error(SART_ERR_INVALID_ARGUMENT, "energies must be positive and finite");

  EM_HIP(hipSetDevice(sart::context_device(ctx)));
  hipStream_t stream = sart::context_stream(ctx);
  if (int rc = upload_nodes()) return rc;
  Scoped d_zones, d_energies, d_abs;
  EM_HIP(hipMalloc(&d_zones.p, hz.size() * sizeof(EmZone)));
  EM_HIP(hipMalloc(&d_energies.p, (size_t)n_energies * sizeof(double)));
  EM_HIP(hipMemcpyAsync(d_zones.p, hz.data(), hz.size() * sizeof(EmZone), hipMemcpyHostToDevice, stream));
  EM_HIP(hipMemcpyAsync(d_energies.p, energies, (size_t)n_energies * sizeof(double), hipMemcpyHostToDevice, stream));
  const size_t plane = (size_t)n_radii * n_energies;
  const double* abs_dev = nullptr;
  if (abs_coefs && abs_on_device) abs_dev = abs_coefs;
  else if (abs_coefs) {
    EM_HIP(hipMalloc(&d_abs.p, plane * sizeof(double)));
    EM_HIP(hipMemcpyAsync(d_abs.p, abs_coefs, plane * sizeof(double), hipMemcpyHostToDevice, stream));
    abs_dev = static_cast<const double*>(d_abs.p);
  }
  EmArgs A;
  A.zones = static_cast<const EmZone*>(d_zones.p);
  A.energies = static_cast<const double*>(d_energies.p);
  A.abs_coefs = abs_dev;
  A.out = out_dev;
  A.components = comp_dev;
  A.n_radii = n_radii;
  A.n_energies = n_energies;
  A.terms = params->terms;
  A.g_ae = params->g_ae;
  A.g_agamma = params->g_agamma;
  A.g_anuclei = params->g_anuclei;
  hipEvent_t e0, e1;
  EM_HIP(hipEventCreate(&e0));
  EM_HIP(hipEventCreate(&e1));
  EM_HIP(hipEventRecord(e0, stream));
  hipLaunchKernelGGL(emission_table_kernel, dim3((n_energies + 255) / 256, n_radii), dim3(256), 0, stream, A);
  EM_HIP(hipEventRecord(e1, stream));
  EM_HIP(hipGetLastError());
  // the scratch buffers above die with this scope: wait for the kernel
  EM_HIP(hipStreamSynchronize(stream));
  float ms = 0.f;
  EM_HIP(hipEventElapsedTime(&ms, e0, e1));
  g_last_kernel_ms = ms;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return 0;
}

int check_args(sart_context* ctx, const sart_solar_zone_t* zones, int32_t n_radii, const double* energies, int32_t n_energies,
               const sart_emission_params_t* params, const double* out) {
  if (!ctx) return sart::set_error(SART_ERR_INVALID_ARGUMENT, "ctx is NULL");
  if (!zones || !energies || !params || !out) return sart::set_error(SART_ERR_INVALID_ARGUMENT, "NULL argument");
  if (n_radii < 1 || n_energies < 1 || n_radii > 65535) return sart::set_error(SART_ERR_INVALID_ARGUMENT, "n_radii must be in [1, 65535], n_energies >= 1");
  if ((params->terms & ~SART_EM_ALL) != 0u) return sart::set_error(SART_ERR_INVALID_ARGUMENT, "unknown bit in params.terms");
  return 0;
}

}  // namespace

extern "C" {

void sart_emission_default_params(sart_emission_params_t* p) {
  if (!p) return;
  p->g_ae = 1e-13;       // readOpacityFile.nim:640
  p->g_agamma = 1e-12;   // :641
  p->g_anuclei = 1e-15;  // :643
  p->terms = SART_EM_ALL;
  p->_pad = 0;
}

int sart_emission_table_device(sart_context* ctx, const sart_solar_zone_t* zones, int32_t n_radii, const double* energies_kev,
                               int32_t n_energies, const double* abs_coefs_dev, const sart_emission_params_t* params,
                               double* em_rates_dev, double* components_dev) {
  if (int rc = check_args(ctx, zones, n_radii, energies_kev, n_energies, params, em_rates_dev)) return rc;
  return run(ctx, zones, n_radii, energies_kev, n_energies, abs_coefs_dev, true, params, em_rates_dev, components_dev);
}

int sart_emission_table(sart_context* ctx, const sart_solar_zone_t* zones, int32_t n_radii, const double* energies_kev,
                        int32_t n_energies, const double* abs_coefs, const sart_emission_params_t* params, double* em_rates_out,
                        double* components_out) {
  if (int rc = check_args(ctx, zones, n_radii, energies_kev, n_energies, params, em_rates_out)) return rc;
  EM_HIP(hipSetDevice(sart::context_device(ctx)));
  const size_t plane = (size_t)n_radii * n_energies;
  Scoped d_out, d_comp;
  EM_HIP(hipMalloc(&d_out.p, plane * sizeof(double)));
  if (components_out) EM_HIP(hipMalloc(&d_comp.p, plane * SART_EM_N_TERMS * sizeof(double)));
  if (int rc = run(ctx, zones, n_radii, energies_kev, n_energies, abs_coefs, false, params, static_cast<double*>(d_out.p),
                   static_cast<double*>(d_comp.p)))
    return rc;
  EM_HIP(hipMemcpy(em_rates_out, d_out.p, plane * sizeof(double), hipMemcpyDeviceToHost));
  if (components_out) EM_HIP(hipMemcpy(components_out, d_comp.p, plane * SART_EM_N_TERMS * sizeof(double), hipMemcpyDeviceToHost));
  return 0;
}

/* AGSS09 zones -> emission table -> fluxRadiusCDF / diffFluxCDFs / guide tables, all on the device (BASELINE configs[4]:
 * readOpacityFile.nim:745-860 followed by raytracer.nim:2670-2705 without the solar_model_dataframe.csv in between). */
int sart_emission_to_solar_tables(sart_context* ctx, const sart_solar_zone_t* zones, int32_t n_radii, const double* energies_kev,
                                  int32_t n_energies, const double* abs_coefs_dev, const sart_emission_params_t* params) {
  if (!ctx || !zones || !energies_kev || !params || n_radii < 1 || n_energies < 2)
    return sart::set_error(SART_ERR_INVALID_ARGUMENT, "sart_emission_to_solar_tables: bad argument");
  EM_HIP(hipSetDevice(sart::context_device(ctx)));
  Scoped d_em;
  EM_HIP(hipMalloc(&d_em.p, (size_t)n_radii * n_energies * sizeof(double)));
  if (int rc = sart_emission_table_device(ctx, zones, n_radii, energies_kev, n_energies, abs_coefs_dev, params,
                                          static_cast<double*>(d_em.p), nullptr))
    return rc;
  std::vector<double> radii((size_t)n_radii);
  for (int32_t r = 0; r < n_radii; ++r) radii[r] = zones[r].radius_frac;
  return sart_set_solar_tables_device(ctx, static_cast<const double*>(d_em.p), radii.data(), energies_kev, n_radii, n_energies);
}

/* Duration of the last emission_table_kernel launch of this process in ms (HIP events on the launch stream). */
double sart_emission_last_kernel_ms(void) { return g_last_kernel_ms; }

}  // extern "C"
