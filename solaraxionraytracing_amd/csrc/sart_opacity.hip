// sart_opacity.hip — absorption coefficients of the solar plasma from the OPCD monochromatic opacities, on the GPU
// (include/sart_emission.h; readOpacityFile.nim:790-823).
//
// One thread per (radius, energy) cell: w = E / T_table -> line number of the frequency mesh (binary search in the 10001
// mesh points, an 80 KB table every cell shares) -> one linear interpolation per metal in the table of the zone's
// (temperature, electron density) slot -> sum in the reference's order.  Neighbouring energies of a radius read neighbouring
// table lines, so the 256 threads of a block cover a few KB of each of the 15 tables; the kernel is a gather over tables
// that stay in L2 (15 x 80 KB per slot) plus 8 B of output per cell.  No LDS, no MFMA.
//
// Rounds like the reference's C code: no FMA contraction in this file (flag in the Makefile and the pragma below).
// There is no CPU fallback.
#include <hip/hip_runtime.h>

#include <cmath>
#include <string>
#include <vector>

#include "../../include/sart_emission.h"

#pragma clang fp contract(off)

namespace sart {
int context_device(sart_context* c);
hipStream_t context_stream(sart_context* c);
int set_error(int code, const std::string& msg);
}  // namespace sart

namespace {

struct AbsZone {
  double temp_table;      // 10^(0.025 temp_index) * 8.617e-8 (:758): w = E / temp_table (:793)
  double temp;            // temp_K * 8.617e-8 (:759-760)
  int slot;
  int _pad;
};

struct AbsArgs {
  const AbsZone* zones;
  const double* n_z;       // [n_radii][29]
  const double* energies;
  const double* u_mesh;
  const int* element_z;
  const long long* y_begin;
  const long long* x_begin;
  const int* len;
  const double* table_x;
  const double* table_y;
  double* out;
  unsigned* n_outside;     // cells whose abscissa left a table (numericalnim raises)
  int n_radii, n_energies, n_mesh, n_elements;
};

// numericalnim's Linear1D eval on (xs, ys): the interval by binary search, y0 + (x - x0) * (y1 - y0) / (x1 - x0); false
// outside [xs[0], xs[n - 1]] (the library raises).
__device__ __forceinline__ bool linear1d(const double* __restrict__ xs, const double* __restrict__ ys, int n, double x, double& y) {
  if (!(x >= xs[0]) || !(x <= xs[n - 1])) return false;
  int lo = 0, hi = n - 1;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (xs[mid] <= x) lo = mid; else hi = mid;
  }
  const double x0 = xs[lo], x1 = xs[lo + 1], y0 = ys[lo], y1 = ys[lo + 1];
  y = y0 + (x - x0) * (y1 - y0) / (x1 - x0);
  return true;
}

// The same on the abscissae 1, 2, ..., n (the 10000-line tables: energy = line count + 1, :206-208).
__device__ __forceinline__ bool linear1d_lines(const double* __restrict__ ys, int n, double x, double& y) {
  if (!(x >= 1.0) || !(x <= (double)n)) return false;
  int lo = (int)x - 1;              // last line number <= x, as an index
  lo = lo > n - 2 ? n - 2 : lo;
  const double x0 = (double)(lo + 1), x1 = (double)(lo + 2), y0 = ys[lo], y1 = ys[lo + 1];
  y = y0 + (x - x0) * (y1 - y0) / (x1 - x0);
  return true;
}

__global__ __launch_bounds__(256) void abs_coef_kernel(AbsArgs A) {
  const int iE = blockIdx.x * 256 + threadIdx.x;
  const int R = blockIdx.y;
  if (iE >= A.n_energies) return;
  const AbsZone z = A.zones[R];
  const double energy = A.energies[iE];
  const double w = energy / z.temp_table;                                   // :793
  double abs_coef;
  if (w >= 20.0 || w <= 0.0732) {
    abs_coef = 0.0;   // :801-808: the sum of n_Z * 0.0 times finite positive factors
  } else {
    bool ok = true;
    double table = 0.0;
    {
      // spline.eval(w) (:825): xs = the mesh, ys = the line numbers 0, 1, ... (linspace(0, 10000, 10001), :286)
      const double* __restrict__ u = A.u_mesh;
      const int n = A.n_mesh;
      if (!(w >= u[0]) || !(w <= u[n - 1])) {
        ok = false;
      } else {
        int lo = 0, hi = n - 1;
        while (hi - lo > 1) {
          const int mid = (lo + hi) >> 1;
          if (u[mid] <= w) lo = mid; else hi = mid;
        }
        const double x0 = u[lo], x1 = u[lo + 1], y0 = (double)lo, y1 = (double)(lo + 1);
        table = y0 + (w - x0) * (y1 - y0) / (x1 - x0);
      }
    }
    double sum = 0.0;
    if (ok) {
      const size_t row = (size_t)z.slot * (size_t)A.n_elements;
      const double* nz = A.n_z + (size_t)R * 29;
      for (int k = 0; k < A.n_elements; ++k) {
        const long long yb = A.y_begin[row + k], xb = A.x_begin[row + k];
        const int n = A.len[row + k];
        double opacity = 0.0;
        const bool in = (xb < 0) ? linear1d_lines(A.table_y + yb, n, table, opacity)
                                 : linear1d(A.table_x + xb, A.table_y + yb, n, table, opacity);
        ok = ok && in;
        sum = sum + nz[A.element_z[k]] * opacity;                           // :833-834
      }
    }
    if (!ok) {
      atomicAdd(A.n_outside, 1u);
      abs_coef = nan("");
    } else {
      abs_coef = sum * 1.97327e-8 * 0.528e-8 * 0.528e-8 * (1.0 - exp(-energy / z.temp));   // :838
    }
  }
  A.out[(size_t)R * A.n_energies + iE] = abs_coef;
}

#define OP_HIP(call)                                                                                                   \
  do {                                                                                                                 \
    hipError_t e_ = (call);                                                                                            \
    if (e_ != hipSuccess)                                                                                              \
      return sart::set_error(e_ == hipErrorOutOfMemory ? SART_ERR_OUT_OF_MEMORY : SART_ERR_NO_DEVICE,                 \
                             std::string("sart_emission_abs_coefs: ") + #call + ": " + hipGetErrorString(e_));        \
  } while (0)

double g_last_kernel_ms = 0.0;

struct Scoped {  // frees device scratch on every exit path
  void* p = nullptr;
  ~Scoped() { if (p) (void)hipFree(p); }
};

int invalid(const std::string& msg) { return sart::set_error(SART_ERR_INVALID_ARGUMENT, "sart_emission_abs_coefs: " + msg); }

// Everything the kernel indexes with is checked here: a table description that points outside its pools never reaches the GPU.
int check_tables(const sart_opacity_tables_t& T, int32_t n_radii) {
  if (!T.u_mesh || T.n_mesh < 2) return invalid("the frequency mesh needs at least two points");
  for (int32_t i = 1; i < T.n_mesh; ++i)
    if (!(T.u_mesh[i] > T.u_mesh[i - 1])) return invalid("the frequency mesh must be strictly ascending (line " + std::to_string(i) + ")");
  if (T.n_slots < 1 || T.n_elements < 1 || !T.slot_of_zone || !T.element_z || !T.table_y_begin || !T.table_x_begin || !T.table_len || !T.table_y)
    return invalid("incomplete opacity tables");
  if (T.n_table_y < 2 || T.n_table_x < 0 || (T.n_table_x > 0 && !T.table_x)) return invalid("empty opacity pools");
  for (int32_t r = 0; r < n_radii; ++r)
    if (T.slot_of_zone[r] < 0 || T.slot_of_zone[r] >= T.n_slots) return invalid("slot_of_zone[" + std::to_string(r) + "] outside [0, n_slots)");
  for (int32_t k = 0; k < T.n_elements; ++k) {
    if (T.element_z[k] < 0 || T.element_z[k] > 28) return invalid("element_z must be a proton number in [0, 28]");
    if (k > 0 && T.element_z[k] <= T.element_z[k - 1]) return invalid("element_z must ascend (the order of the reference's sum)");
  }
  const size_t n_tab = (size_t)T.n_slots * (size_t)T.n_elements;
  for (size_t t = 0; t < n_tab; ++t) {
    const int64_t yb = T.table_y_begin[t], xb = T.table_x_begin[t];
    const int64_t n = T.table_len[t];
    const std::string what = "table " + std::to_string(t % (size_t)T.n_elements) + " of slot " + std::to_string(t / (size_t)T.n_elements);
    if (n < 2) return invalid(what + " has fewer than two points");
    if (yb < 0 || yb + n > T.n_table_y) return invalid(what + " lies outside the opacity pool");
    if (xb >= 0) {
      if (xb + n > T.n_table_x) return invalid(what + " lies outside the abscissa pool");
      for (int64_t i = 1; i < n; ++i)
        if (!(T.table_x[xb + i] > T.table_x[xb + i - 1])) return invalid(what + ": abscissae must be strictly ascending");
    } else if (xb != -1) {
      return invalid(what + ": table_x_begin must be an offset or -1");
    }
  }
  return 0;
}

int run(sart_context* ctx, const sart_solar_zone_t* zones, int32_t n_radii, const double* n_z, const double* energies,
        int32_t n_energies, const sart_opacity_tables_t* tables, double* out_dev) {
  if (!ctx) return sart::set_error(SART_ERR_INVALID_ARGUMENT, "ctx is NULL");
  if (!zones || !n_z || !energies || !tables || !out_dev) return invalid("NULL argument");
  if (n_radii < 1 || n_energies < 1 || n_radii > 65535) return invalid("n_radii must be in [1, 65535], n_energies >= 1");
  const sart_opacity_tables_t& T = *tables;
  if (int rc = check_tables(T, n_radii)) return rc;
  std::vector<AbsZone> hz((size_t)n_radii);
  for (int32_t R = 0; R < n_radii; ++R) {
    const sart_solar_zone_t& z = zones[R];
    if (!(z.temp_K > 0.0) || z.temp_index <= 0) return invalid("solar zone " + std::to_string(R) + ": temperature must be positive");
    AbsZone& a = hz[R];
    a.temp_table = std::pow(10.0, ((double)z.temp_index * 0.025)) * 8.617e-8;
    a.temp = z.temp_K * 8.617e-8;
    a.slot = T.slot_of_zone[R];
    a._pad = 0;
  }
  for (int32_t i = 0; i < n_energies; ++i)
    if (!(energies[i] > 0.0) || !std::isfinite(energies[i])) return invalid("energies must be positive and finite");

  OP_HIP(hipSetDevice(sart::context_device(ctx)));
  hipStream_t stream = sart::context_stream(ctx);
  const size_t n_tab = (size_t)T.n_slots * (size_t)T.n_elements;
  Scoped d_zones, d_nz, d_en, d_mesh, d_ez, d_yb, d_xb, d_len, d_tx, d_ty, d_flag;
  auto up = [&](Scoped& d, const void* src, size_t bytes) -> hipError_t {
    if (hipError_t e = hipMalloc(&d.p, bytes ? bytes : 8)) return e;
    return bytes ? hipMemcpyAsync(d.p, src, bytes, hipMemcpyHostToDevice, stream) : hipSuccess;
  };
  OP_HIP(up(d_zones, hz.data(), hz.size() * sizeof(AbsZone)));
  OP_HIP(up(d_nz, n_z, (size_t)n_radii * 29 * sizeof(double)));
  OP_HIP(up(d_en, energies, (size_t)n_energies * sizeof(double)));
  OP_HIP(up(d_mesh, T.u_mesh, (size_t)T.n_mesh * sizeof(double)));
  OP_HIP(up(d_ez, T.element_z, (size_t)T.n_elements * sizeof(int32_t)));
  OP_HIP(up(d_yb, T.table_y_begin, n_tab * sizeof(int64_t)));
  OP_HIP(up(d_xb, T.table_x_begin, n_tab * sizeof(int64_t)));
  OP_HIP(up(d_len, T.table_len, n_tab * sizeof(int32_t)));
  OP_HIP(up(d_tx, T.table_x, (size_t)T.n_table_x * sizeof(double)));
  OP_HIP(up(d_ty, T.table_y, (size_t)T.n_table_y * sizeof(double)));
  OP_HIP(hipMalloc(&d_flag.p, sizeof(unsigned)));
  OP_HIP(hipMemsetAsync(d_flag.p, 0, sizeof(unsigned), stream));
  AbsArgs A;
  A.zones = static_cast<const AbsZone*>(d_zones.p);
  A.n_z = static_cast<const double*>(d_nz.p);
  A.energies = static_cast<const double*>(d_en.p);
  A.u_mesh = static_cast<const double*>(d_mesh.p);
  A.element_z = static_cast<const int*>(d_ez.p);
  A.y_begin = static_cast<const long long*>(d_yb.p);
  A.x_begin = static_cast<const long long*>(d_xb.p);
  A.len = static_cast<const int*>(d_len.p);
  A.table_x = static_cast<const double*>(d_tx.p);
  A.table_y = static_cast<const double*>(d_ty.p);
  A.out = out_dev;
  A.n_outside = static_cast<unsigned*>(d_flag.p);
  A.n_radii = n_radii;
  A.n_energies = n_energies;
  A.n_mesh = T.n_mesh;
  A.n_elements = T.n_elements;
  hipEvent_t e0, e1;
  OP_HIP(hipEventCreate(&e0));
  OP_HIP(hipEventCreate(&e1));
  OP_HIP(hipEventRecord(e0, stream));
  hipLaunchKernelGGL(abs_coef_kernel, dim3((n_energies + 255) / 256, n_radii), dim3(256), 0, stream, A);
  OP_HIP(hipEventRecord(e1, stream));
  OP_HIP(hipGetLastError());
  unsigned n_outside = 0;
  OP_HIP(hipMemcpyAsync(&n_outside, d_flag.p, sizeof(unsigned), hipMemcpyDeviceToHost, stream));
  OP_HIP(hipStreamSynchronize(stream));   // the scratch buffers above die with this scope
  float ms = 0.f;
  OP_HIP(hipEventElapsedTime(&ms, e0, e1));
  g_last_kernel_ms = ms;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  if (n_outside)
    return invalid(std::to_string(n_outside) + " cells evaluate a table outside its abscissae (the reference's interpolator raises there)");
  return 0;
}

}  // namespace

extern "C" {

int sart_emission_abs_coefs_device(sart_context* ctx, const sart_solar_zone_t* zones, int32_t n_radii, const double* n_z,
                                   const double* energies_kev, int32_t n_energies, const sart_opacity_tables_t* tables,
                                   double* abs_coefs_dev) {
  return run(ctx, zones, n_radii, n_z, energies_kev, n_energies, tables, abs_coefs_dev);
}

int sart_emission_abs_coefs(sart_context* ctx, const sart_solar_zone_t* zones, int32_t n_radii, const double* n_z,
                            const double* energies_kev, int32_t n_energies, const sart_opacity_tables_t* tables,
                            double* abs_coefs_out) {
  if (!ctx) return sart::set_error(SART_ERR_INVALID_ARGUMENT, "ctx is NULL");
  if (!abs_coefs_out || n_radii < 1 || n_energies < 1) return invalid("NULL argument");
  OP_HIP(hipSetDevice(sart::context_device(ctx)));
  const size_t plane = (size_t)n_radii * (size_t)n_energies;
  Scoped d_out;
  OP_HIP(hipMalloc(&d_out.p, plane * sizeof(double)));
  if (int rc = run(ctx, zones, n_radii, n_z, energies_kev, n_energies, tables, static_cast<double*>(d_out.p))) return rc;
  OP_HIP(hipMemcpy(abs_coefs_out, d_out.p, plane * sizeof(double), hipMemcpyDeviceToHost));
  return 0;
}

/* calculateOpacities with the OPCD files (:731-860) followed by the CDF loops of initFullSetup (raytracer.nim:2670-2705),
 * nothing leaving the device in between: absCoef -> emission table -> sampling tables of the context. */
int sart_emission_to_solar_tables_opcd(sart_context* ctx, const sart_solar_zone_t* zones, int32_t n_radii, const double* n_z,
                                       const double* energies_kev, int32_t n_energies, const sart_opacity_tables_t* tables,
                                       const sart_emission_params_t* params) {
  if (!ctx) return sart::set_error(SART_ERR_INVALID_ARGUMENT, "ctx is NULL");
  if (n_radii < 1 || n_energies < 1) return invalid("NULL argument");
  OP_HIP(hipSetDevice(sart::context_device(ctx)));
  Scoped d_abs;
  OP_HIP(hipMalloc(&d_abs.p, (size_t)n_radii * (size_t)n_energies * sizeof(double)));
  if (int rc = run(ctx, zones, n_radii, n_z, energies_kev, n_energies, tables, static_cast<double*>(d_abs.p))) return rc;
  return sart_emission_to_solar_tables(ctx, zones, n_radii, energies_kev, n_energies, static_cast<const double*>(d_abs.p), params);
}

/* Duration of the last abs_coef_kernel launch of this process in ms (HIP events on the launch stream). */
double sart_emission_abs_coefs_last_kernel_ms(void) { return g_last_kernel_ms; }

}  // extern "C"
