// sart_api.hip — implementation of the sart.h C-ABI: context, table upload, hoisting of
// per-setup / per-shell / per-energy-index constants, kernel launches, timing.
//
// There is NO CPU fallback in this library: without a HIP device sart_create() fails with
// SART_ERR_NO_DEVICE, and nothing here links or loads oracle/.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>   // declarations only (types, enums, signatures): the library itself is dlopen()ed
#include <dlfcn.h>
#include <sys/mman.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/sart.h"
#include "sart_device.h"

namespace sart {
void launch_trace_histogram(const HotA& H, const HotB& HB, const DevBlob* blob, const TraceArgs& A, double* acc, int n_blocks,
                            hipStream_t stream, int variant, bool fixed);
void launch_finalize_fixed(const void* in, const void* hi, double* out, size_t n_img, int spectra, int n_radial_bins, int n_energies1, double q_w,
                           double q_w2, double q_pos, double q_refl, void* check_dev, hipStream_t stream);
void launch_rollover_fixed(void* acc, void* hi, size_t n, void* check_dev, hipStream_t stream);
size_t fixed_check_bytes();
std::string fixed_check_describe(const void* host_copy);
bool launch_trace_mass_scan(const HotA& H, const HotB& HB, const DevBlob* blob, const TraceArgs& A, const ScanArgs& SC, double* rows,
                            double* shared_row, int n_blocks, hipStream_t stream, int variant, bool fixed);
void launch_finalize_scan(const void* in, double* out, int n_masses, const double* q_w, const double* q_w2, int shared_row,
                          void* check_dev, hipStream_t stream, uint32_t counter_slots);
void launch_trace_angular_scan(const HotA& H, const HotB& HB, const DevBlob* blob, const TraceArgs& A, const AScanArgs& AN, double* rows,
                               double* shared_row, int n_blocks, hipStream_t stream, bool fast, bool fixed);
int angular_scan_blocks_per_cu(bool fast);
int histogram_block_of(int variant);
void launch_trace_records(const HotA& H, const HotB& HB, const DevBlob* blob, const TraceArgs& A, sart_axion_t* out, int n_blocks,
                          hipStream_t stream, const double* uniforms_dev);
int records_block();
int compact_chunk_max();
void launch_compact_records(const sart_axion_t* rec, uint32_t n, sart_axion_t* out, unsigned long long capacity, uint32_t* block_counts,
                            unsigned long long* block_first, unsigned long long* counts, hipStream_t stream);
int histogram_blocks_per_cu(int variant);
void launch_build_solar_tables(const double* em_dev, const double* radii_dev, const double* energies_dev, int n_radii, int n_energies,
                               double* cdf_dev, double* row_sum_dev, double* rcdf_dev, uint16_t* rguide_dev, uint16_t* eguide_dev,
                               uint32_t* status_dev, hipStream_t stream);
void launch_cdf_hi32(const double* cdf_dev, uint32_t* out_dev, size_t n, hipStream_t stream);
}  // namespace sart

using namespace sart;

namespace {

thread_local std::string g_err;
int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
#define SART_HIP(call)                                                                              \
  do {                                                                                              \
    hipError_t e_ = (call);                                                                         \
    if (e_ != hipSuccess)                                                                           \
      return fail(e_ == hipErrorOutOfMemory ? SART_ERR_OUT_OF_MEMORY : SART_ERR_NO_DEVICE,          \
                  std::string(#call) + ": " + hipGetErrorString(e_));                               \
  } while (0)

constexpr double kPi = 3.14159265358979323846;
inline double deg2rad(double d) { return d * (kPi / 180.0); }  // Nim std/math degToRad
inline double cot(double x) { return 1.0 / std::tan(x); }

template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  ~DevBuf() { if (p) (void)hipFree(p); }
  int upload(const T* host, size_t count) {
    if (count != n || !p) {
      if (p) (void)hipFree(p);
      p = nullptr; n = 0;
      SART_HIP(hipMalloc(reinterpret_cast<void**>(&p), count * sizeof(T)));
      n = count;
    }
    SART_HIP(hipMemcpy(p, host, count * sizeof(T), hipMemcpyHostToDevice));
    return 0;
  }
  int resize(size_t count) {
    if (count == n && p) return 0;
    if (p) (void)hipFree(p);
    p = nullptr; n = 0;
    SART_HIP(hipMalloc(reinterpret_cast<void**>(&p), count * sizeof(T)));
    n = count;
    return 0;
  }
  // Grow-only scratch: room for at least `count` elements.  (hipFree synchronises the device: a scratch buffer that followed the
  // size of every call would do that whenever consecutive calls differ in size.)
  int reserve(size_t count) { return (p && n >= count) ? 0 : resize(count); }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr; n = 0;
  }
};

// numericalnim newLinear1D.eval (reference call sites raytracer.nim:1522-1527, 2170-2190)
double linear1d(const std::vector<double>& xs, const std::vector<double>& ys, double x) {
  size_t lo = 0, hi = xs.size() - 1;
  while (hi - lo > 1) {
    const size_t mid = (lo + hi) / 2;
    if (xs[mid] <= x) lo = mid; else hi = mid;
  }
  return ys[lo] + (x - xs[lo]) * (ys[lo + 1] - ys[lo]) / (xs[lo + 1] - xs[lo]);
}

// (cos, sin)(pi k / 64) for k = 0 .. 128, rounded from long double.  Only the first octant is evaluated (arguments
// <= pi/4, where sinl / cosl lose nothing); the rest follows by symmetry, so the zeros and ones at the multiples of
// pi/2 are exact and sin^2 + cos^2 has the same rounding in every octant.
std::vector<double> sincos_table() {
  std::vector<double> t(2 * static_cast<size_t>(kSinCosEntries));
  const long double pi = 3.14159265358979323846264338327950288L;
  for (int k = 0; k < kSinCosEntries; ++k) {
    const int q = k / 32, j = k % 32;          // angle = q pi/2 + j pi/64
    long double c, sn;
    if (j <= 16) { c = cosl(pi * j / 64); sn = sinl(pi * j / 64); }
    else { c = sinl(pi * (32 - j) / 64); sn = cosl(pi * (32 - j) / 64); }
    if (j == 0) { c = 1.0L; sn = 0.0L; }
    double cd = static_cast<double>(c), sd = static_cast<double>(sn);
    switch (q & 3) {                            // rotate by q quarter turns
      case 0: t[2 * k] = cd; t[2 * k + 1] = sd; break;
      case 1: t[2 * k] = -sd; t[2 * k + 1] = cd; break;
      case 2: t[2 * k] = -cd; t[2 * k + 1] = -sd; break;
      default: t[2 * k] = sd; t[2 * k + 1] = -cd; break;
    }
    if (t[2 * k] == 0.0) t[2 * k] = 0.0;        // no negative zeros
    if (t[2 * k + 1] == 0.0) t[2 * k + 1] = 0.0;
  }
  return t;
}

}  // namespace
// the table as the library uploads it (tests/test_gpu_math.py feeds it to the device math under test)
extern "C" __attribute__((visibility("default"))) void sart_internal_sincos_table(double* out) {
  const std::vector<double> t = sincos_table();
  std::memcpy(out, t.data(), t.size() * sizeof(double));
}
namespace {

size_t lower_bound_idx(const double* a, size_t n, double key) {
  return static_cast<size_t>(std::lower_bound(a, a + n, key) - a);
}

}  // namespace

struct sart_context {
  int device = 0;
  int n_cu = 0;
  int blocks_per_cu_hist[7] = {0, 0, 0, 0, 0, 0, 0}, blocks_per_cu_rec = 0, blocks_per_cu_ascan[2] = {0, 0};
  // tuning / experiment knobs, read from the environment once when the context is created
  struct Knobs {
#ifdef SART_DEBUG_KNOBS              // experiment builds only (make DEBUG_KNOBS=1); compiled out of the shipped library
    bool no_image_atomics = false;   // SART_DEBUG_NO_IMAGE_ATOMICS: timing experiment only (results are wrong)
    uint32_t debug_flags = 0;        // SART_DEBUG_FLAGS=<hex>: 0x20000000 stage A0 drops every ray, 0x10000000 stage A1 does
#endif
    bool no_early_reject = false;    // SART_NO_EARLY_REJECT: stage A0 off
    bool no_sure_miss = false;       // SART_NO_SURE_MISS: rays that provably miss the innermost shell's mirror go through phase B
    bool no_image_tile = false;      // SART_NO_IMAGE_TILE: small focal spots go to global atomics only (as before the tile)
    bool no_path_const = false;      // SART_NO_PATH_CONST: never use the constant-path kernel variant (5)
    bool no_host_prefault = false;   // SART_NO_HOST_PREFAULT: sart_trace_records leaves the caller's buffer as it finds it
    int records_chunk = 0;           // SART_RECORDS_CHUNK: records per chunk of sart_trace_records (0 = 1 Mi)
    int records_fail_chunk = 0;      // SART_RECORDS_FAIL_CHUNK=k: test hook - the k-th chunk of sart_trace_records fails instead of being traced
    int prefault_threads = 0;        // SART_PREFAULT_THREADS: host threads that fault the caller's record buffer in (0 = 8)
    bool force_generic = false;      // SART_FORCE_GENERIC: never use the specialised kernel variant
    int image_replicas = 0;          // SART_IMAGE_REPLICAS: 0 = chosen from the plate scale
    int hist_blocks_per_cu = 0;      // SART_HIST_BLOCKS_PER_CU: 0 = occupancy query
  } knobs;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  std::string device_name;

  // host copies of the inputs
  bool have_setup = false, have_solar = false, have_refl = false, have_det = false;
  sart_setup_t setup;
  std::vector<double> energies;   // [nE]
  int n_radii = 0, n_energies = 0;
  std::vector<double> refl_data;  // [nC][nA][nEr]
  int refl_nc = 0, refl_na = 0, refl_ne = 0;
  double refl_amin = 0, refl_amax = 0, refl_emin = 0, refl_emax = 0;
  std::vector<double> sb_x, sb_y, win_x, win_y, gas_x, gas_y;

  // device state
  DevParams params;  // host mirror of the blob's parameter block
  HotA hot;          // passed by value to every launch
  HotB hotb;         // table bases of phase B's gathers, by value too
  DevBuf<DevBlob> d_blob;
  bool blob_dirty = true;
  bool spot_may_have_moved = true;   // geometry, tables or angles changed since the LDS image tile was placed
  std::vector<ShellDev> shells;
  std::vector<uint8_t> shell_lut;
  int radius_span = 0;
  bool path_const = false;     // no survivor of phase A entered through the bore wall (path_is_constant)
  DevBuf<ShellDev> d_shells;
  DevBuf<double> d_sincos;     // (cos, sin)(pi k / 64), k = 0 .. 128 (sampling angles, sart_kernels.hip: sincos_turns)
  DevBuf<uint8_t> d_lut;
  DevBuf<double> d_rcdf, d_ecdf, d_refl;
  DevBuf<uint32_t> d_ecdf_hi32;   // upper 32 of the 52 bits of floor(d_ecdf 2^52): the candidates of the energy draw (HotB::cdf_hi32)
  DevBuf<uint16_t> d_rguide, d_eguide;
  DevBuf<EnergyDev> d_etab;
  DevBuf<double> d_replicas;   // kImageReplicas scratch images (kept zeroed between launches)
  DevBuf<double> d_partials;   // per-workgroup partial sums of the scalars
  DevBuf<double> d_acc;        // scratch accumulator of the blocking convenience call
  bool d_acc_stale = false;    // its contents belong to another accumulation mode: the next call starts from zero
  DevBuf<sart_axion_t> d_rec;  // scratch records of the blocking convenience call
  DevBuf<sart_axion_t> d_rec2; // its second half-buffer (chunked, double-buffered record path)
  // passed-rays-only record path: compacted chunks (two half-buffers), scan scratch, counts {n_rays, n_passed, n_passed_till_window,
  // n_hit_nickel} per half-buffer on the device and in pinned host memory
  DevBuf<sart_axion_t> d_cmp;  // passed records only: second compacted half-buffer (the first is d_rec2, which that path does not use otherwise)
  DevBuf<uint32_t> d_cmp_counts;
  DevBuf<unsigned long long> d_cmp_first, d_cmp_totals;
  unsigned long long* h_cmp_totals = nullptr;
  hipStream_t copy_stream = nullptr;   // D2H of the record chunks, beside the kernels on `stream`
  hipEvent_t rec_traced[2] = {nullptr, nullptr}, rec_copied[2] = {nullptr, nullptr};
  bool derived_dirty = true;
  // LDS image tile (sart_device.h: TraceArgs::tile_*): centre of the focal spot in image pixels, found by a pilot launch
  // for the current setup and image binning
  struct TileCache {
    bool valid = false, in_pilot = false;
    int32_t nx = 0, ny = 0;
    double x_min = 0, x_max = 0, y_min = 0, y_max = 0;
    int32_t x0 = 0, y0 = 0, n = 0;
    bool ring_cells_free = false;   // which of the two tile sizes n was chosen for
  } tile;
  DevBuf<double> d_pilot;
  DevBuf<double> d_pilot_replicas;   // the pilot's own scratch images: the caller's replica buffer keeps its layout

  // accumulation mode (include/sart.h): SART_ACCUM_FIXED64 adds integers; the quanta are frozen by the first launch
  int accum_mode = SART_ACCUM_F64;
  int headroom_bits = 27;
  bool quanta_frozen = false;
  int weight_exp = 0;                // q_w = 2^weight_exp
  int weight_sq_exp = 0;             // q_w2 = 2^weight_sq_exp
  double weight_bound = 0.0;         // the bound the frozen quantum was derived from
  std::vector<double> etab_max;      // max t_window / t_strongback / a_gas over the energy table, max reflectivity (hoist_*)
  std::vector<EnergyDev> etab_host;  // host copy of the per-energy-index table (the gas-stage weight bound follows the axion mass)
  DevBuf<double> d_fin;              // f64 image of d_acc for the blocking host-output calls in FIXED64 mode
  // What the finalize kernels found wrong with a raw FIXED64 accumulator (sart_kernels.hip: kFixedStatus*): OR-ed into d_status,
  // copied to the pinned h_status behind every finalize, read - and turned into an error - by the next synchronising call.
  DevBuf<uint8_t> d_status;          // a FixedCheck (sart_kernels.hip); its first word is the status
  uint32_t* h_status = nullptr;
  bool status_pending = false;
  // fused mass scan: per-workgroup per-mass partial sums, scratch accumulators of the blocking call
  DevBuf<double> d_scan_partials, d_scan, d_scan_fin;
  DevBuf<double> d_ascan_partials;   // fused angular scan: per-workgroup per-angle partial sums

  // timing
  bool timing = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
  size_t events_used = 0;
};

namespace {

// ---- hoisting: sart_setup_t -> DevParams (geometry part) -------------------------------------
int hoist_setup(sart_context* c) {
  const sart_setup_t& s = c->setup;
  DevParams& P = c->params;
  std::memset(&P, 0, sizeof P);
  P.sun_distance = s.distance_sun_earth;
  P.sun_radius = s.radius_sun;
  P.radius_cb = s.magnet_radiusCB;
  P.radius_cb_sq = s.magnet_radiusCB * s.magnet_radiusCB;
  P.length_b = s.magnet_lengthB;
  P.length_coldbore = s.magnet_lengthColdbore;
  P.pipe1_len = s.pipe_cb_vt3_length;
  P.pipe2_len = s.pipe_vt3_xrt_length;
  P.pipe1_radius_sq = s.pipe_cb_vt3_radius * s.pipe_cb_vt3_radius;
  P.n_radii = c->n_radii;
  P.n_energies = c->n_energies;
  P.radius_span = c->radius_span;
  // X-ray test source, initCenterVectors raytracer.nim:305-311
  P.test_active = s.test_active;
  P.test_parallel = s.test_parallel;
  P.test_x = s.test_off_axis_left;
  P.test_y = s.test_off_axis_up;
  P.test_z = -s.test_distance;
  P.test_radius = s.test_radius;
  P.test_radius_sq = s.test_radius * s.test_radius;
  P.test_collimator_z = -s.test_distance + s.test_length_col;
  // telescope frame, :1878-1899
  P.telescope_kind = s.telescope_kind;
  P.telescope_wolter = (s.telescope_kind == SART_TK_XMM || s.telescope_kind == SART_TK_ABRIXAS) ? 1 : 0;
  P.n_shells = s.n_shells;
  P.l_mirror = s.l_mirror;
  const double turnedX = deg2rad(s.telescope_turned_x_deg), turnedY = deg2rad(s.telescope_turned_y_deg);
  P.rotated = (s.telescope_turned_x_deg != 0.0 || s.telescope_turned_y_deg != 0.0) ? 1 : 0;
  P.rx_c = std::cos(turnedX); P.rx_s = std::sin(turnedX);
  P.ry_c = std::cos(turnedY); P.ry_s = std::sin(turnedY);
  {
    const double a0 = deg2rad(s.all_angles_deg[0]);  // lengthTelescope :1883-1884
    const double lt = (s.l_mirror + 0.5 * s.all_xsep[0]) * std::cos(a0) + (s.l_mirror + 0.5 * s.all_xsep[0]) * std::cos(3.0 * a0);
    P.half_length_telescope = lt / 2;
  }
  P.entrance_x = s.optics_entrance[0];
  P.entrance_y = s.optics_entrance[1];
  // opaque structures, :1635-1704
  P.spoke_n = 16;
  P.spoke_cos_thr = 2.0;
  if (s.telescope_kind == SART_TK_XMM) {
    // 17 bands at i * 22.5 deg +- 1.145 deg on phi in [0, 180] (:1695-1699) = 16 spokes around the circle
    P.spider_z = -85.0; P.inner_radius = 64.7; P.ring_lo = 151.6 - 20.9; P.ring_hi = 151.6;
    P.spoke_n = 16; P.spoke_cos_thr = std::cos(deg2rad(16.0 * 1.145));
  } else if (s.telescope_kind == SART_TK_ABRIXAS) {
    // 7 bands at i * 60 deg +- 3.75 deg (:1657-1660) = 6 spokes around the circle
    P.spider_z = -35.0; P.inner_radius = 37.5; P.ring_lo = 0.0; P.ring_hi = 0.0;
    P.spoke_n = 6; P.spoke_cos_thr = std::cos(deg2rad(6.0 * 3.75));
  }
  P.hole_type = s.hole_type;
  P.number_of_holes = s.number_of_holes;
  P.hole_in_optics = s.hole_in_optics;
  P.inner_blocks = (s.hole_type == SART_HT_NONE) ? 1 : -1;
  // detector plane
  const double pipeRad = deg2rad(s.pipes_turned_deg);
  P.pipe_c = std::cos(pipeRad); P.pipe_s = std::sin(pipeRad);
  P.d_cb_xray = -s.optics_entrance[0];  // :2073
  P.lateral_shift = s.lateral_shift;
  P.transversal_shift = s.transversal_shift;
  P.radius_window_sq = s.radius_window * s.radius_window;
  P.chip_cx = s.chip_x_max / 2.0;
  P.chip_cy = s.chip_y_max / 2.0;
  P.theta_c = std::cos(s.theta_rad); P.theta_s = std::sin(s.theta_rad);
  P.n_half_strips = static_cast<int>(std::round(s.number_of_strips / 2.0));  // :2167
  if (P.n_half_strips > kMaxStrips) return fail(SART_ERR_INVALID_ARGUMENT, "number_of_strips too large");
  for (int i = 0; i < P.n_half_strips; ++i) {  // :2168-2169
    P.strip_lo[i] = (1.0 * i + 0.5) * s.strip_dist_window + i * s.strip_width_window;
    P.strip_hi[i] = (1.0 * i + 0.5) * s.strip_dist_window + (i + 1.0) * s.strip_width_window;
  }
  P.stage_gas = (s.stage == SART_SK_GAS) ? 1 : 0;
  // weights
  {
    // conversionProb :363-365 with unchained's natural units: T -> eV^2 (195.353), m -> eV^-1 (1/1.97327e-7)
    const double k = (s.g_agamma * 1e-9) * (s.magnet_B * 195.353) * (1e-3 / 1.97327e-7) / 2.0;
    P.conv_k = k * k;
  }
  P.exposure = (s.experiment == SART_ES_CAST) ? 3.585e3 * 3600.0 * 1.5 * 90.0 : 9.5e6 * 3600.0 * 12.0 * 90.0;  // :2207-2212
  {
    // gas stage, axionMassforMagnet.nim:30-101 (pGas in bar handed to mbar-documented functions — sic, :1601)
    const double pGas = s.magnet_pGasRoom / s.room_temp * s.magnet_tGas;
    const double length = s.magnet_lengthB * 1e-3, radBore = s.magnet_radiusCB * 1e-3;
    const double vol = length * (kPi * std::pow(radBore, 2));
    const double amountMol = (pGas * 1e2) * vol / (8.314 * s.magnet_tGas);
    const double ne = 2 * 6.022e23 * (amountMol / vol);
    const double m_gamma = std::sqrt(std::pow(1.97e-7, 3) * 4 * kPi * (1.0 / 137.0) * ne / 511e3);
    P.gas_m_gamma_sq = m_gamma * m_gamma;
    const double g_ev = s.g_agamma * 1e-9, beV = s.magnet_B * 1e3 / 1.444;
    P.gas_term1 = std::pow(g_ev * beV / 2.0, 2);
    P.gas_inv_hbarc_m = 1e-3 / 1.97e-7;
    P.gas_dm2_abs = std::fabs(P.gas_m_gamma_sq - s.m_axion * s.m_axion);
  }
  // reflectivity
  P.n_coatings = c->refl_nc;
  P.refl_n_angles = c->refl_na;
  if (c->have_refl) {
    P.refl_angle_min = c->refl_amin;
    P.refl_dangle = (c->refl_amax - c->refl_amin) / static_cast<double>(c->refl_na - 1);
    P.refl_inv_dangle = 1.0 / P.refl_dangle;
  }
  // shells
  c->shells.assign(static_cast<size_t>(s.n_shells), ShellDev{});
  for (int j = 0; j < s.n_shells; ++j) {
    ShellDev& sh = c->shells[j];
    sh.coating = 0;
    if (s.reflectivity_kind == SART_RK_MULTI_COATING)  // layers.lowerBound(hitLayer) :1573 (first boundary >= hitLayer)
      sh.coating = static_cast<int>(std::lower_bound(s.coating_layers, s.coating_layers + s.n_coatings, j) - s.coating_layers);
    sh.refl_row0 = sh.coating * (c->n_energies + 1);
    const double r1 = s.all_r1[j], l = s.l_mirror, xSep = s.all_xsep[j];
    const double beta = deg2rad(s.all_angles_deg[j]), beta3 = 3.0 * beta;
    const double distanceMirrors = std::cos(beta) * (xSep + l);  // :1973
    sh.r1 = r1;
    sh.r1_outer = r1 + s.all_thickness[j];
    if (!P.telescope_wolter) {
      // findPosCone :628-658
      const double r2 = r1 - l * std::sin(beta);               // :1954-1957
      const double r3 = r2 - 0.5 * xSep * std::tan(beta);
      const double r4 = r3 - 0.5 * xSep * std::tan(3.0 * beta);
      const double t1 = std::tan(beta), k1 = t1 * t1;
      sh.m1_k = k1; sh.m1_hb = r1 * t1; sh.m1_cc = r1 * r1; sh.m1_zhi = l * std::cos(beta);
      const double t2 = std::tan(beta3), k2 = t2 * t2, zm = distanceMirrors;
      sh.m2_k = k2;
      sh.m2_hb = r4 * t2 + k2 * zm;
      sh.m2_cc = r4 * r4 + 2.0 * r4 * t2 * zm + k2 * zm * zm;
      sh.m2_zlo = zm; sh.m2_zhi = zm + l * std::cos(beta3);
      sh.m2_rc = r4;
      sh.n1_tan = t1; sh.n2_tan = t2;
    } else {
      // findPosParabolic :660-690
      const double tb = std::tan(beta);
      const double r3 = -tb * l + std::sqrt(tb * l * tb * l + r1 * r1);
      const double e = 2.0 * r3 * tb;
      sh.m1_k = 0.0; sh.m1_hb = e / 2.0; sh.m1_cc = r3 * r3 + e * l; sh.m1_zhi = l * std::cos(beta);
      sh.n1_r3t = r3 * tb; sh.n1_r3sq = r3 * r3; sh.n1_e = e; sh.n1_tan = tb;
      // findPosHyperbolic :692-729 (angle = 3 beta)
      const double ta = std::tan(beta3 / 3.0);
      const double r3h = -ta * l + std::sqrt(ta * l * ta * l + r1 * r1);
      const double f = s.distance_detector_xrt;
      const double t3 = std::tan(beta3);
      const double eh = 2.0 * r3h * t3;
      const double F = f + r3h * cot(2.0 * beta3 / 3.0);
      const double g = 2.0 * r3h * t3 / F;
      sh.m2_k = g;
      sh.m2_hb = g * l + eh / 2.0;
      sh.m2_cc = r3h * r3h + eh * l + g * l * l;
      sh.m2_zlo = distanceMirrors; sh.m2_zhi = distanceMirrors + l * std::cos(beta3);
      sh.m2_rc = 0.0;
      sh.n2_r3t = r3h * t3; sh.n2_r3sq = r3h * r3h; sh.n2_e = eh; sh.n2_invF = 1.0 / F; sh.n2_tan = t3;
    }
    // distDet :2070-2072 (hard-coded shell index 8 — sic), getPointDetectorWindow :811
    const double distDet = distanceMirrors - 0.5 * s.all_xsep[8] * std::cos(beta) + s.distance_detector_xrt -
                           s.distance_window_focal_plane;
    sh.dist_det_raw = distDet;
    sh.dist_det = distDet / std::cos(pipeRad);
    sh.dist_det_end = (distDet + s.depth_det) / std::cos(pipeRad);
    sh.nickel_num = (j > 0) ? (r1 - (s.all_r1[j - 1] + s.all_thickness[j - 1])) : 0.0;  // :1722
  }
  // radial look-up table of the shell selection: cell k = [k step, (k+1) step) -> first shell with R1 > k step;
  // step is below the smallest shell spacing, so the answer is lut[k] or lut[k] + 1
  {
    double min_gap = s.all_r1[0];
    for (int j = 1; j < s.n_shells; ++j) min_gap = std::min(min_gap, s.all_r1[j] - s.all_r1[j - 1]);
    const double r_last = s.all_r1[s.n_shells - 1];
    double step = std::min(1.0, 0.5 * min_gap);
    step = std::max(step, r_last / (kShellLutMax - 2));
    if (!(step < min_gap)) return fail(SART_ERR_UNSUPPORTED, "shell radii too dense for the shell look-up table");
    P.r1_last = r_last;
    P.lut_inv_step = 1.0 / step;
    P.lut_n = static_cast<int>(r_last / step) + 2;
    c->shell_lut.assign(static_cast<size_t>(P.lut_n), 0);
    for (int k = 0; k < P.lut_n; ++k) {
      // a little below k*step so that rounding of radial * inv_step can never skip a shell
      const double lo = std::max(0.0, (k - 1e-9) * step);
      int j = 0;
      while (j < s.n_shells && !(s.all_r1[j] > lo)) ++j;
      c->shell_lut[k] = static_cast<uint8_t>(j);
    }
  }
  return 0;
}

// ---- per-energy-index tables -------------------------------------------------------------------
int hoist_energy_tables(sart_context* c) {
  const sart_setup_t& s = c->setup;
  const int nE = c->n_energies;
  std::vector<EnergyDev> tab(static_cast<size_t>(nE) + 1);
  const double pGas = s.magnet_pGasRoom / s.room_temp * s.magnet_tGas;
  auto density = [](double p, double temp) {  // axionMassforMagnet.nim:4-15
    const double pressure = p * 1e2;
    double r = pressure * 4.002602 / (8.314 * temp * 1000.0);
    return r / 1000.0;
  };
  for (int i = 0; i <= nE; ++i) {
    const double E = (i < nE) ? std::max(0.03, c->energies[i]) : s.test_energy;  // :470-471 / :1771
    EnergyDev& e = tab[i];
    e.energy = E;
    e.t_window = linear1d(c->win_x, c->win_y, E);
    e.t_strongback = linear1d(c->sb_x, c->sb_y, E);
    e.a_gas = linear1d(c->gas_x, c->gas_y, E);
    const double massAtt = std::exp(-1.5832 + 5.9195 * std::exp(-0.353808 * E) + 4.03598 * std::exp(-0.970557 * E));  // :70-73
    e.gamma = 1.97e-7 * 100.0 * density(pGas, s.magnet_tGas) * massAtt;  // :84
    e.inv_two_e_ev = 1.0 / (2 * (E * 1000.0));                             // :68 (the kernel multiplies)
    e.mu_pipe = massAtt * density(pGas, s.room_temp) * 100;                // :109-113
    e.mu_magnet = massAtt * density(pGas, s.magnet_tGas) * 100;
  }
  // bounds for the FIXED64 weight quantum: the solar source draws indices < nE ([0..3]: maxima over those rows), the X-ray test
  // source uses row nE alone ([4..7]: its values - at BabyIAXO's 0.021 keV they are orders of magnitude below the maxima)
  c->etab_max.assign(8, 0.0);
  for (int i = 0; i < nE; ++i) {
    const EnergyDev& e = tab[i];
    c->etab_max[0] = std::max(c->etab_max[0], std::fabs(e.t_window));
    c->etab_max[1] = std::max(c->etab_max[1], std::fabs(e.t_strongback));
    c->etab_max[2] = std::max(c->etab_max[2], std::fabs(e.a_gas));
  }
  c->etab_max[4] = std::fabs(tab[nE].t_window);
  c->etab_max[5] = std::fabs(tab[nE].t_strongback);
  c->etab_max[6] = std::fabs(tab[nE].a_gas);
  c->etab_host = tab;
  return c->d_etab.upload(tab.data(), tab.size());
}

// Reflectivity re-tabulated per energy index: for each coating and each energy index the bilinear
// interpolation (numericalnim eval_bilinear; call sites :1567-1568, :1577-1578) is carried out along
// the energy axis, leaving g[i] with  R(alpha, E_idx) = g[i] + xUnit (g[i+1] - g[i]).
int hoist_reflectivity(sart_context* c) {
  const int nE = c->n_energies, nA = c->refl_na, nEr = c->refl_ne, nC = c->refl_nc;
  // the kernel addresses the re-tabulated grid with 32-bit byte offsets built from 24-bit multiplicands
  if (nA >= (1 << 24) || static_cast<size_t>(nC) * (nE + 1) * nA >= (size_t(1) << 29))
    return fail(SART_ERR_UNSUPPORTED, "reflectivity grid too large: n_coatings * (n_energies + 1) * n_angles must stay below 2^29");
  std::vector<double> out(static_cast<size_t>(nC) * (nE + 1) * nA);
  const double dy = (c->refl_emax - c->refl_emin) / static_cast<double>(nEr - 1);
  for (int e = 0; e <= nE; ++e) {
    const double E = (e < nE) ? std::max(0.03, c->energies[e]) : c->setup.test_energy;
    long j = static_cast<long>(std::floor((E - c->refl_emin) / dy));
    j = std::min<long>(j, nEr - 2);
    if (j < 0) j = 0;  // below the grid the reference would index out of bounds; hold the first cell
    const double yCorner = c->refl_emin + static_cast<double>(j) * dy;
    const double yUnit = (E - yCorner) / dy;
    for (int cc = 0; cc < nC; ++cc) {
      const double* z = c->refl_data.data() + static_cast<size_t>(cc) * nA * nEr;
      double* g = out.data() + (static_cast<size_t>(cc) * (nE + 1) + e) * nA;
      for (int i = 0; i < nA; ++i) {
        const double f0 = z[static_cast<size_t>(i) * nEr + j], f1 = z[static_cast<size_t>(i) * nEr + j + 1];
        g[i] = f0 + (f1 - f0) * yUnit;
      }
    }
  }
  {
    // |R| <= max |g| for every angle inside the grid (the kernel clamps the cell; outside it extrapolates from the edge cell);
    // [3]: the rows of the solar energies, [7]: the rows of the test source's energy (index nE of every coating)
    double m = 0.0, mt = 0.0;
    for (int cc = 0; cc < nC; ++cc)
      for (int e = 0; e <= nE; ++e) {
        const double* g = out.data() + (static_cast<size_t>(cc) * (nE + 1) + e) * nA;
        double& dst = (e < nE) ? m : mt;
        for (int i = 0; i < nA; ++i) dst = std::max(dst, std::fabs(g[i]));
      }
    c->etab_max[3] = m;
    c->etab_max[7] = mt;
  }
  return c->d_refl.upload(out.data(), out.size());
}

int refresh_derived(sart_context* c) {
  if (!c->derived_dirty) return 0;
  if (!(c->have_setup && c->have_solar && c->have_refl && c->have_det))
    return fail(SART_ERR_NOT_READY, "setup, solar tables, reflectivity and detector tables must all be set before tracing");
  const sart_setup_t& s = c->setup;
  if (s.reflectivity_kind == SART_RK_MULTI_COATING && c->refl_nc < s.n_coatings)
    return fail(SART_ERR_INVALID_ARGUMENT, "multi-coating telescope needs one reflectivity grid per coating");
  // Launches still running on the context's stream read the tables that are overwritten below (blocking copies on the
  // null stream do not wait for a non-blocking stream).
  SART_HIP(hipStreamSynchronize(c->stream));
  if (int rc = hoist_setup(c)) return rc;
  if (int rc = c->d_shells.upload(c->shells.data(), c->shells.size())) return rc;
  if (int rc = c->d_lut.upload(c->shell_lut.data(), c->shell_lut.size())) return rc;
  if (int rc = hoist_energy_tables(c)) return rc;
  if (int rc = hoist_reflectivity(c)) return rc;
  c->derived_dirty = false;
  c->blob_dirty = true;
  c->spot_may_have_moved = true;
  return 0;
}

int make_args(sart_context* c, const sart_trace_params_t* p, TraceArgs& a) {
  if (!p) return fail(SART_ERR_INVALID_ARGUMENT, "params is NULL");
  const bool no_image = p->image_nx == 0 && p->image_ny == 0;   // flux-only launch (include/sart.h: sart_trace_params_t)
  if (!no_image && (p->image_nx < 1 || p->image_ny < 1 || !(p->image_x_max > p->image_x_min) || !(p->image_y_max > p->image_y_min)))
    return fail(SART_ERR_INVALID_ARGUMENT, "invalid image specification");
  if (static_cast<int64_t>(p->image_nx) * static_cast<int64_t>(p->image_ny) >= (int64_t(1) << 29))
    return fail(SART_ERR_INVALID_ARGUMENT, "image_nx * image_ny must be below 2^29 (pixel byte offsets are 32-bit on the device)");
  a.replicas = nullptr;
  a.partials = nullptr;
  a.replica_mask = 0u;
  a.replica_stride = 0u;
  a.n_rays = p->n_rays;
  a.ray_id_offset = p->ray_id_offset;
  a.seed_lo = static_cast<uint32_t>(p->seed);
  a.seed_hi = static_cast<uint32_t>(p->seed >> 32);
  a.flags = p->flags;
#ifdef SART_DEBUG_KNOBS
  if (c->knobs.no_image_atomics) a.flags |= 0x40000000u;
  a.flags |= c->knobs.debug_flags & 0x3F000000u;
#endif
  a.image_nx = p->image_nx;
  a.image_ny = p->image_ny;
  a.image_x_min = p->image_x_min;
  a.image_y_min = p->image_y_min;
  a.image_inv_step_x = 1.0 / ((p->image_x_max - p->image_x_min) / static_cast<double>(p->image_nx));  // :828-830
  a.image_inv_step_y = 1.0 / ((p->image_y_max - p->image_y_min) / static_cast<double>(p->image_ny));
  if (no_image) {   // zero columns and rows: the kernel's bounds test `0 <= t < n` fails for every ray - no pixel, no atomic, no tile
    a.image_x_min = a.image_y_min = 0.0;
    a.image_inv_step_x = a.image_inv_step_y = 0.0;
  }
  a.tile_x0 = a.tile_y0 = a.tile_n = a.tile_base = 0;
  a.spectra = p->spectra ? 1 : 0;
  a.n_radial_bins = 0;
  a.radial_inv_bin = 0.0;
  if (p->spectra) {
    if (p->n_radial_bins < 1 || !(p->radial_max > 0.0)) return fail(SART_ERR_INVALID_ARGUMENT, "invalid radial histogram specification");
    a.n_radial_bins = p->n_radial_bins;
    a.radial_inv_bin = static_cast<double>(p->n_radial_bins) / p->radial_max;
  }
  (void)c;
  return 0;
}

size_t acc_len_of(const sart_context* c, const sart_trace_params_t* p) {
  return p->spectra ? sart_accumulator_len_spectra(p->image_nx, p->image_ny, p->n_radial_bins, c->n_energies)
                    : sart_accumulator_len(p->image_nx, p->image_ny);
}

HotA hot_of(const DevParams& P) {
  HotA h;
  std::memset(&h, 0, sizeof h);
  h.sun_distance = P.sun_distance; h.sun_radius = P.sun_radius;
  h.radius_cb = P.radius_cb; h.radius_cb_sq = P.radius_cb_sq; h.length_b = P.length_b; h.length_coldbore = P.length_coldbore;
  h.dz1 = P.length_coldbore - P.length_b;
  h.dz2 = h.dz1 + P.pipe1_len;
  h.dz3 = h.dz2 + P.pipe2_len;
  h.pipe1_radius_sq = P.pipe1_radius_sq;
  h.entrance_x = P.entrance_x; h.entrance_y = P.entrance_y;
  h.r1_last = P.r1_last; h.lut_inv_step = P.lut_inv_step;
  h.spider_z = P.spider_z; h.inner_radius = P.inner_radius;
  // 16 spokes: the kernel compares the scaled Chebyshev value z = T16 / 32768 (sart_kernels.hip: spoke_measure)
  h.spoke_cos_thr = (P.spoke_n == 16) ? P.spoke_cos_thr * (1.0 / 32768.0) : P.spoke_cos_thr;
  h.ring_lo = P.ring_lo; h.ring_hi = P.ring_hi;
  h.test_active = P.test_active; h.rotated = P.rotated; h.telescope_kind = P.telescope_kind; h.spoke_n = P.spoke_n;
  h.n_shells = P.n_shells; h.lut_n = P.lut_n; h.radius_span = P.radius_span; h.inner_blocks = P.inner_blocks;
  return h;
}

// DevParams::shell0_miss_radius, or -1 where the bound below is not available (X-ray test source: slopes unbounded; cones).
double shell0_miss_radius_of(const sart_setup_t& s, const DevParams& P, int n_radii) {
  if (P.test_active || n_radii < 1) return -1.0;   // the bound on |slope| holds for rays from the Sun only
  const double R = P.radius_cb;
  const double r_sun_max = (0.0015 + (n_radii - 1) * 0.0005) * P.sun_radius;
  const double s_max = (r_sun_max + R) / (P.sun_distance + P.length_b - r_sun_max) * (1.0 + 1e-6);
  // Innermost shell of a Wolter telescope (XMM, Abrixas: findPosParabolic, :1985-1988): its first mirror is the surface
  // rho(z)^2 = r3^2 + e (l - z), 0 < z < l cos(beta), never closer to the axis than r3 (:668-675).  In the telescope's frame
  // (where phase A measures the radial distance rho0 at the entrance plane z = 0, :1897-1905) the ray is a straight line whose
  // angle to the telescope axis is at most atan(s_max) + the tilt of the telescope (a rotation is rigid), so over that range
  // it stays within rho0 + s_tel l cos(beta) of the axis: below r3 it cannot cross the surface there, neither root is accepted
  // (:677-682), `s = 0` returns the input point, lineHitsNickel is false for the lowest shell (:1719) and the no-hit test ends
  // the ray (:2055).  Offsets of the entrance do not enter (rho0 is measured behind them).  1e-3 mm of margin against rounding
  // in either evaluation.
  if (P.telescope_wolter && s.n_shells > 0) {
    const double tilt = P.rotated ? std::acos(std::min(1.0, P.rx_c * P.ry_c)) : 0.0;   // angle between magnet and telescope axis
    const double s_tel = std::tan(std::atan(s_max) + tilt) * (1.0 + 1e-6);
    const double beta = s.all_angles_deg[0] * (kPi / 180.0), l = s.l_mirror, r1 = s.all_r1[0];
    const double tl = std::tan(beta) * l;
    const double r3 = -tl + std::sqrt(tl * tl + r1 * r1);
    return std::max(-1.0, r3 - s_tel * l * std::cos(beta) - 1e-3);
  }
  return -1.0;
}

// A bound on the angle of the telescope's rotation (rotateInY o rotateInX about (0, 0, lT/2), raytracer.nim:1888-1894) for
// telescope_turned_y = angle_y_deg: the angle of a product of two rotations is at most the sum of their angles.  Exactly 0 for a
// telescope that is not turned (then build_zones adds no margin: the zones of rounds 1-5).  With SART_NO_TILT_ZONES set in the
// environment (experiments: A/B of the zones for turned telescopes) a turned telescope gets none of kind (b), as before round 6.
double tilt_bound_of(const sart_setup_t& s, double angle_y_deg) {
  const double t = std::fabs(deg2rad(s.telescope_turned_x_deg)) + std::fabs(deg2rad(angle_y_deg));
  if (t == 0.0) return 0.0;
  static const bool no_tilt_zones = std::getenv("SART_NO_TILT_ZONES") != nullptr;
  return no_tilt_zones ? -1.0 : t * (1.0 + 1e-9);
}

// Stage A0 zones (see HotA).  With r = R sqrt(u3) the distance of the point on the bore exit from the axis and
// |slope| <= s_max for every ray from the Sun, the ray's distance from the axis at a plane dz further on lies in
// [r - dz s_max, r + dz s_max].  From that: (a) r - dz_k s_max >= R_k for one of the three cuts behind the
// magnetic field (cold-bore exit, two pipe cuts; raytracer.nim:1846-1868) => dead, not "reached";
// (b) telescopes on the magnet axis (entrance offset 0): r certainly inside bore + pipes and certainly inside the inner
// disc / the XMM ring / beyond the outermost shell (:1653, :1674-1692, :1934) => dead, "reached".  Every bound carries a
// safety margin far above f64 rounding, so the verdict equals the reference's.
//
// (b) for a TURNED telescope (round 6; `tilt_max` = a bound on the angle of the rotation :1888-1894, |turnedX| + |turnedY|, in
// radians; a fused angular scan passes the largest of its launch).  The frame change maps the ray's two points A = (x1, y1, -Lp),
// B = (x3, y3, 0) with q - c = Rot (p - c), c = (0, 0, lT/2), and measures the radial distance where the new line meets z = 0
// (:1897-1905).  A rotation by an angle <= tilt_max moves B by delta <= 2 sin(tilt_max / 2) |B - c| <= 2 sin(tilt_max / 2)
// sqrt(R_pipe^2 + (lT/2)^2) (B passed the last pipe cut: zones of kind (b) hold below K_in only), in particular |b_z| <= delta;
// the turned line makes an angle gamma <= atan(s_max) + tilt_max with the z axis, so from b to the plane z = 0 it moves
// sideways by at most delta tan(gamma).  Hence | (X0, Y0) - (x3, y3) | <= delta (1 + tan(gamma)): the radial uncertainty at the
// entrance grows by that much and the three zones shrink accordingly (0.3 deg on BabyIAXO / XMM: 2.5 mm on 65 / 350 mm).  No
// zone of kind (b) beyond gamma = 0.5 rad (they would be empty anyway).
void build_zones(const sart_setup_t& s, const DevParams& P, int n_radii, HotA& h, double tilt_max) {
  h.n_zones = 0;
  h.zone_reached = 0;
  for (int z = 0; z < kMaxZones; ++z) { h.zone_lo[z] = 1u; h.zone_hi[z] = 0u; }   // empty
  if (P.test_active || n_radii < 1) return;   // the bound on |slope| holds for rays from the Sun only
  const double R = P.radius_cb;
  const double r_sun_max = (0.0015 + (n_radii - 1) * 0.0005) * P.sun_radius;
  const double s_max = (r_sun_max + R) / (P.sun_distance + P.length_b - r_sun_max) * (1.0 + 1e-6);
  const double eps = 1e-6;  // mm
  const double dz[3] = {h.dz1, h.dz2, h.dz3};
  const double Rk[3] = {R, std::sqrt(P.pipe1_radius_sq), std::sqrt(P.pipe1_radius_sq)};
  // (a) r >= K_dead => dead
  double K_dead = 1e300;
  for (int k = 0; k < 3; ++k) K_dead = std::min(K_dead, Rk[k] + dz[k] * s_max + eps);
  // r <= K_in => certainly through entrance plane, cold-bore exit and both pipes
  double K_in = R - P.length_b * s_max - eps;
  for (int k = 0; k < 3; ++k) K_in = std::min(K_in, Rk[k] - dz[k] * s_max - eps);
  double spread = dz[2] * s_max + eps;   // radial uncertainty at the telescope entrance
  const double gamma = std::atan(s_max) + tilt_max;
  const bool tilt_ok = tilt_max >= 0.0 && gamma < 0.5;
  if (tilt_max > 0.0 && tilt_ok) {
    const double h_tel = P.half_length_telescope;
    const double delta = 2.0 * std::sin(0.5 * tilt_max) * std::sqrt(Rk[2] * Rk[2] + h_tel * h_tel);
    spread += delta * (1.0 + std::tan(gamma)) * (1.0 + 1e-6) + eps;
  }
  struct Z { double lo, hi; bool reached; };
  std::vector<Z> zones;
  if (K_dead < R) zones.push_back({K_dead, 1e300, false});
  // zones in terms of the radial distance at the telescope entrance need the telescope on the magnet axis
  const bool on_axis = (P.entrance_x == 0.0 && P.entrance_y == 0.0) && tilt_ok;
  if (on_axis && K_in > 0) {
    auto add_reached = [&](double lo, double hi) {   // radial in [lo, hi] certainly => blocked; clip to r <= K_in
      hi = std::min(hi, K_in);
      if (hi > lo) zones.push_back({lo, hi, true});
    };
    if (s.telescope_kind == SART_TK_XMM && P.inner_blocks > 0) add_reached(0.0, P.inner_radius - spread);       // r <= 64.7
    if (s.telescope_kind == SART_TK_ABRIXAS) add_reached(0.0, P.inner_radius - spread);                          // r < 37.5
    if (s.telescope_kind == SART_TK_XMM) add_reached(P.ring_lo + spread, P.ring_hi - spread);                    // ring
    add_reached(P.r1_last + spread, std::min(K_in, K_dead));                                                      // beyond the last shell
  }
  for (const Z& z : zones) {
    if (h.n_zones >= kMaxZones) break;
    // u3 = (r/R)^2; hi word w covers u3 in [w, w+1) / 2^32: keep only words entirely inside the zone
    const double ulo = std::min(1.0, (z.lo / R) * (z.lo / R)), uhi = std::min(1.0, (z.hi / R) * (z.hi / R));
    const double wlo = std::ceil(ulo * 4294967296.0) + 1.0, whi = std::floor(uhi * 4294967296.0) - 2.0;
    if (!(whi >= wlo)) continue;
    h.zone_lo[h.n_zones] = static_cast<uint32_t>(std::min(wlo, 4294967295.0));
    h.zone_hi[h.n_zones] = static_cast<uint32_t>(std::min(whi, 4294967295.0));
    if (z.reached) h.zone_reached |= (1u << h.n_zones);
    h.n_zones++;
  }
}

// True if every ray from the Sun that survives the three cuts behind the magnetic field (cold-bore exit, two pipe cuts;
// raytracer.nim:1846-1868) crossed the entrance plane z = 0 inside the bore, i.e. none of them entered through the bore
// wall (lineIntersectsCylinderOnce, :1825-1843): with |slope| <= s_max a ray that is within R_k of the axis at the plane
// dz_k behind the field exit was within R_k + (lengthB + dz_k) s_max of it at z = 0.  The path in the magnetic field is
// then lengthB (times the slope factor) for every ray that reaches the mirrors, and the kernel variant 5 does not carry it.
bool path_is_constant(const DevParams& P, const HotA& h, int n_radii) {
  if (P.test_active || n_radii < 1) return false;
  const double R = P.radius_cb;
  const double r_sun_max = (0.0015 + (n_radii - 1) * 0.0005) * P.sun_radius;
  const double s_max = (r_sun_max + R) / (P.sun_distance + P.length_b - r_sun_max) * (1.0 + 1e-6);
  const double dz[3] = {h.dz1, h.dz2, h.dz3};
  const double Rk[3] = {R, std::sqrt(P.pipe1_radius_sq), std::sqrt(P.pipe1_radius_sq)};
  double reach = 1e300;   // largest distance from the axis at z = 0 of a ray that passes all three cuts
  for (int k = 0; k < 3; ++k) reach = std::min(reach, Rk[k] + (P.length_b + dz[k]) * s_max);
  return reach + 1e-6 < R;
}

DevTables tables_of(sart_context* c);

// (Re)uploads the parameter blob if the host mirror changed.  Ordered after all work already queued on
// the stream: the previous launches read the old blob.
int sync_blob(sart_context* c) {
  if (!c->blob_dirty) return 0;
  SART_HIP(hipStreamSynchronize(c->stream));
  c->params.shell0_miss_radius =
      (c->knobs.no_sure_miss || c->knobs.no_early_reject) ? -1.0 : shell0_miss_radius_of(c->setup, c->params, c->n_radii);
  DevBlob b;
  std::memset(&b, 0, sizeof b);
  b.P = c->params;
  b.T = tables_of(c);
  if (int rc = c->d_blob.upload(&b, 1)) return rc;
  c->hot = hot_of(c->params);
  c->hotb.diff_flux_cdfs = c->d_ecdf.p;
  c->hotb.cdf_hi32 = c->d_ecdf_hi32.p;
  c->hotb.energy_guide = c->d_eguide.p;
  c->hotb.energy_tab = c->d_etab.p;
  c->hotb.refl = c->d_refl.p;
  c->hotb.n_energies = c->n_energies;
  c->hotb.refl_n_angles = c->refl_na;
  c->hotb.cdf_stride = c->n_energies + kEnergyCdfPad;
  c->hotb._pad = 0;
  if (!c->knobs.no_early_reject) build_zones(c->setup, c->params, c->n_radii, c->hot, tilt_bound_of(c->setup, c->setup.telescope_turned_y_deg));
  c->path_const = path_is_constant(c->params, c->hot, c->n_radii);
  c->blob_dirty = false;
  if (c->spot_may_have_moved) c->tile.valid = false;   // a new axion mass alone (weights only) keeps the tile where it is
  c->spot_may_have_moved = false;
  return 0;
}

DevTables tables_of(sart_context* c) {
  DevTables t;
  t.sincos_tab = c->d_sincos.p;
  t.shells = c->d_shells.p;
  t.shell_lut = c->d_lut.p;
  t.flux_radius_cdf = c->d_rcdf.p;
  t.radius_guide = c->d_rguide.p;
  t.diff_flux_cdfs = c->d_ecdf.p;
  t.energy_guide = c->d_eguide.p;
  t.energy_tab = c->d_etab.p;
  t.refl = c->d_refl.p;
  return t;
}

// Upper bound of one ray's weight for the current setup, tables and flags (phase_b of sart_kernels.hip: reflectivity^2 x
// cos(yaw) x conversion probability x absorption x window x gas x exposure): the SART_ACCUM_FIXED64 weight quantum is derived
// from it.  It is the scale of the largest weights, not a guarantee: the factor 1 + slope^2 of the path length (< 1.00003 for
// rays from the Sun) and reflectivities extrapolated beyond the edge of their grid are left out, which is harmless - the
// conversion of a weight to quanta stays exact up to 2^(51 - 63 + headroom) >= 2^4 times the bound, it only uses up headroom.
// Gas stage: bound of the conversion probability (axionMassforMagnet.nim:75-98) for |m_gamma^2 - m_a^2| = dm2_abs,
//   P / (g B / 2)^2 = |integral_0^L exp((i q - Gamma / 2) z) dz|^2 <= min(L^2, 4 / (q^2 + Gamma^2 / 4)),   q = dm2_abs / (2 E),
// maximised over the energies a ray can have (the table rows of the solar source, or the test source's one row).  Off resonance
// P falls like 4 / (q L)^2: a bound that ignored the mass (L^2 alone) would leave the weights of a far-off-resonance scan point
// 1e-6 .. 1e-7 of it, at or below what the integer quanta resolve.
double gas_prob_bound(const sart_context* c, double dm2_abs) {
  const DevParams& P = c->params;
  const double l_nat = P.length_b * P.gas_inv_hbarc_m;
  const size_t nE = static_cast<size_t>(c->n_energies);
  const size_t lo = P.test_active ? nE : 0, hi = P.test_active ? nE + 1 : nE;
  double best = 0.0;
  for (size_t i = lo; i < hi && i < c->etab_host.size(); ++i) {
    const EnergyDev& e = c->etab_host[i];
    const double q = dm2_abs * e.inv_two_e_ev;
    best = std::max(best, std::min(l_nat * l_nat, 4.0 / (q * q + 0.25 * e.gamma * e.gamma)));
  }
  return P.gas_term1 * best;
}

double weight_bound_of(const sart_context* c, uint32_t flags, double dm2_abs) {
  const DevParams& P = c->params;
  const double* m = c->etab_max.data() + (P.test_active ? 4 : 0);   // the test source has ONE energy: its row, not the maxima
  double b = 1.0;
  if (!(flags & SART_CF_IGNORE_REFLECTION)) b *= m[3] * m[3];
  if (!(flags & SART_CF_IGNORE_CONV_PROB)) {
    // vacuum: conv_k pathCB^2, pathCB ~ lengthB (:363-365); gas: see gas_prob_bound
    b *= P.stage_gas ? gas_prob_bound(c, dm2_abs) : P.conv_k * P.length_b * P.length_b;
  }
  if (!(flags & SART_CF_IGNORE_DET_WINDOW)) b *= std::max(m[0], m[1]);
  if (!(flags & SART_CF_IGNORE_GAS_ABS)) b *= m[2];
  if (!(flags & SART_CF_XRAY_TEST)) b *= P.exposure;
  return b;
}

// The exponents of the FIXED64 quanta for a weight bound b < 2^e: weights (pixels, SUM_WEIGHTS, weight spectra) in units of
// 2^(e - 63 + headroom); squared weights in units of 2^(2 e - 39) - SUM_WEIGHTS_SQ has a high limb, so its quantum only has to
// leave room for the rays of one workgroup of one launch (< 2^23) in an int64.
struct QuantaExp { int w, w2; };
int quanta_exp_of(const sart_context* c, double b, QuantaExp& q, int* bound_exp = nullptr) {
  if (!std::isfinite(b) || b < 0.0) return fail(SART_ERR_INVALID_ARGUMENT, "FIXED64: the weight bound of this setup is not finite");
  int e = 0;
  if (b > 0.0) (void)std::frexp(b, &e);   // b < 2^e
  q.w = e - (63 - c->headroom_bits);
  q.w2 = 2 * e - 39;
  if (bound_exp) *bound_exp = e;
  return 0;
}

// Freezes the FIXED64 quanta from the bound `b` (or checks that `b` still fits the frozen ones).
int freeze_quanta(sart_context* c, double b, bool refreeze) {
  QuantaExp q;
  int e = 0;
  if (int rc = quanta_exp_of(c, b, q, &e)) return rc;
  if (c->quanta_frozen && !refreeze) {
    if (e > c->weight_exp + (63 - c->headroom_bits))
      return fail(SART_ERR_INVALID_ARGUMENT, "FIXED64: this launch's weights do not fit the quantum frozen for the accumulator "
                                             "(flags or setup changed between accumulate == 1 launches)");
    return 0;
  }
  c->weight_exp = q.w;
  c->weight_sq_exp = q.w2;
  c->weight_bound = b;
  c->quanta_frozen = true;
  return 0;
}

// Kernel variant of the accumulating kernels for the context's current setup (sart_kernels.hip: SART_HIST_VARIANTS).
// 0 / 3 / 4: compile-time specialisations for the solar source without the hole loop (vacuum; gas stage; rotated telescope =
// the angular scan); 1 / 2: everything else with the switches read at run time (1 not rotated, 2 rotated); 5 / 6: variants
// 0 / 3 when no surviving ray entered through the bore wall (constant path in the magnetic field: ring 1 does not carry it
// and the LDS image tile - or the mass scan's accumulators - use its space beside stage A0).
int hist_variant_of(const sart_context* c) {
  const DevParams& P = c->params;
  const bool fast = !P.test_active && !(P.telescope_kind == SART_TK_XMM && P.inner_blocks < 0) && !c->knobs.force_generic;
  int variant = P.rotated ? 2 : 1;
  if (fast && !P.rotated) variant = P.stage_gas ? 3 : 0;
  if (fast && P.rotated && !P.stage_gas) variant = 4;
  if ((variant == 0 || variant == 3) && c->path_const && !c->knobs.no_path_const) variant = variant == 0 ? 5 : 6;
  return variant;
}

int grid_for(uint64_t n_rays, int n_cu, int blocks_per_cu, int block) {
  const uint64_t bs = static_cast<uint64_t>(block);
  const uint64_t need = (n_rays + bs - 1) / bs;
  const uint64_t cap = static_cast<uint64_t>(n_cu) * static_cast<uint64_t>(std::max(1, blocks_per_cu));
  return static_cast<int>(std::max<uint64_t>(1, std::min(need, cap)));
}

struct TimedLaunch {
  sart_context* c;
  hipEvent_t stop = nullptr;
  explicit TimedLaunch(sart_context* ctx) : c(ctx) {
    if (!c->timing) return;
    if (c->events_used == c->events.size()) {
      hipEvent_t a, b;
      if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
      c->events.emplace_back(a, b);
    }
    auto& ev = c->events[c->events_used++];
    (void)hipEventRecord(ev.first, c->stream);
    stop = ev.second;
  }
  ~TimedLaunch() { if (stop) (void)hipEventRecord(stop, c->stream); }
};

}  // namespace

// FIXED64 status word (sart_kernels.hip: kFixedStatus*).  ensure: allocates it; enqueue_copy: behind a finalize kernel, the
// cumulative device word -> pinned host memory on the same stream; take: after a stream synchronisation, turns what the
// finalize kernels found into an error (once) and clears it.
int status_ensure(sart_context* c) {
  if (!c->d_status.p) {
    if (int rc = c->d_status.resize(fixed_check_bytes())) return rc;
    // (on the context's stream: a memset on the null stream is not ordered with launches on a non-blocking stream)
    SART_HIP(hipMemsetAsync(c->d_status.p, 0, fixed_check_bytes(), c->stream));
  }
  if (!c->h_status) {
    SART_HIP(hipHostMalloc(reinterpret_cast<void**>(&c->h_status), 4 * sizeof(uint32_t), hipHostMallocDefault));
    std::memset(c->h_status, 0, 4 * sizeof(uint32_t));
  }
  return 0;
}
int status_enqueue_copy(sart_context* c) {
  SART_HIP(hipMemcpyAsync(c->h_status, c->d_status.p, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
  c->status_pending = true;
  return 0;
}
int status_take(sart_context* c) {   // the stream has been synchronised
  if (!c->status_pending) return 0;
  c->status_pending = false;
  const uint32_t st = c->h_status[0];
  if (st == 0) return 0;
  c->h_status[0] = 0;
  SART_HIP(hipMemsetAsync(c->d_status.p, 0, sizeof(uint32_t), c->stream));
  std::string msg = "FIXED64:";
  if (st & 1u) msg += " a slot of the accumulator is negative or >= 2^62 - it wrapped, or is about to (more bound-weight rays on one pixel "
                      "/ bin than 2^headroom_bits: use a larger headroom, or finalize and start a new accumulator earlier);";
  if (st & 4u) {
    msg += " the pixels / radial / energy bins of the accumulator do not add up to its SUM_WEIGHTS: a slot wrapped since the accumulator "
           "was zeroed (use a larger headroom, or finalize and start a new accumulator earlier)";
    std::vector<uint8_t> copy(fixed_check_bytes());   // the sums of the finalize that failed (only read on this path)
    if (hipMemcpy(copy.data(), c->d_status.p, copy.size(), hipMemcpyDeviceToHost) == hipSuccess) msg += fixed_check_describe(copy.data());
    msg += ";";
  }
  if (st & 2u) msg += " the accumulated weights average below 2^12 quanta per passed ray (bound " + std::to_string(c->weight_bound) +
                      ": an outlier in a table inflated it) - choose a smaller headroom or SART_ACCUM_F64;";
  return fail(SART_ERR_ACCUMULATOR, msg);
}

// Used by the other translation units of libsart.so (sart_emission.hip); not part of the C-ABI.
namespace sart {
__attribute__((visibility("hidden"))) int context_device(sart_context* c) { return c->device; }
__attribute__((visibility("hidden"))) hipStream_t context_stream(sart_context* c) { return c->stream; }
__attribute__((visibility("hidden"))) int set_error(int code, const std::string& msg) { return fail(code, msg); }
}  // namespace sart

extern "C" {

int sart_abi_version(void) { return SART_ABI_VERSION; }
const char* sart_last_error(void) { return g_err.c_str(); }

int sart_create(int device_ordinal, sart_context** out) {
  if (!out) return fail(SART_ERR_INVALID_ARGUMENT, "out is NULL");
  *out = nullptr;
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
    return fail(SART_ERR_NO_DEVICE, "no HIP device available (libsart has no CPU fallback)");
  if (device_ordinal < 0 || device_ordinal >= n_dev) return fail(SART_ERR_INVALID_ARGUMENT, "device ordinal out of range");
  SART_HIP(hipSetDevice(device_ordinal));
  auto* c = new sart_context();
  c->device = device_ordinal;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_ordinal) != hipSuccess) {
    delete c;
    return fail(SART_ERR_NO_DEVICE, "hipGetDeviceProperties failed");
  }
  c->n_cu = prop.multiProcessorCount;
  c->device_name = prop.name;
  if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) {
    delete c;
    return fail(SART_ERR_NO_DEVICE, "hipStreamCreate failed");
  }
  c->stream = c->own_stream;
  {
    const std::vector<double> sc = sincos_table();
    if (int rc = c->d_sincos.upload(sc.data(), sc.size())) {
      (void)hipStreamDestroy(c->own_stream);
      delete c;
      return rc;
    }
  }
  {
    auto flag = [](const char* name) { return std::getenv(name) != nullptr; };
    auto number = [](const char* name) { const char* e = std::getenv(name); return e ? std::max(0, std::atoi(e)) : 0; };
#ifdef SART_DEBUG_KNOBS
    c->knobs.no_image_atomics = flag("SART_DEBUG_NO_IMAGE_ATOMICS");
    if (const char* e = std::getenv("SART_DEBUG_FLAGS")) c->knobs.debug_flags = static_cast<uint32_t>(std::strtoul(e, nullptr, 16));
#endif
    c->knobs.no_early_reject = flag("SART_NO_EARLY_REJECT");
    c->knobs.no_sure_miss = flag("SART_NO_SURE_MISS");
    c->knobs.no_image_tile = flag("SART_NO_IMAGE_TILE");
    c->knobs.no_path_const = flag("SART_NO_PATH_CONST");
    c->knobs.no_host_prefault = flag("SART_NO_HOST_PREFAULT");
    c->knobs.records_chunk = number("SART_RECORDS_CHUNK");
    c->knobs.records_fail_chunk = number("SART_RECORDS_FAIL_CHUNK");
    c->knobs.prefault_threads = number("SART_PREFAULT_THREADS");
    c->knobs.force_generic = flag("SART_FORCE_GENERIC");
    c->knobs.image_replicas = number("SART_IMAGE_REPLICAS");
    c->knobs.hist_blocks_per_cu = number("SART_HIST_BLOCKS_PER_CU");
  }
  *out = c;
  return 0;
}

int sart_destroy(sart_context* c) {
  if (!c) return 0;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  for (auto& ev : c->events) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
  for (int k = 0; k < 2; ++k) {
    if (c->rec_traced[k]) (void)hipEventDestroy(c->rec_traced[k]);
    if (c->rec_copied[k]) (void)hipEventDestroy(c->rec_copied[k]);
  }
  if (c->copy_stream) { (void)hipStreamSynchronize(c->copy_stream); (void)hipStreamDestroy(c->copy_stream); }
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  if (c->h_status) (void)hipHostFree(c->h_status);
  if (c->h_cmp_totals) (void)hipHostFree(c->h_cmp_totals);
  delete c;
  return 0;
}

int sart_set_stream(sart_context* c, void* hip_stream) {
  if (!c) return fail(SART_ERR_INVALID_ARGUMENT, "ctx is NULL");
  hipStream_t next = hip_stream ? static_cast<hipStream_t>(hip_stream) : c->own_stream;
  if (next != c->stream) {
    // the scratch buffers of the context (image replicas, per-workgroup partials, parameter blob) are shared by all
    // launches: work queued on the old stream must have finished before launches on the new one reuse them
    SART_HIP(hipSetDevice(c->device));
    SART_HIP(hipStreamSynchronize(c->stream));
  }
  c->stream = next;
  return 0;
}

int sart_synchronize(sart_context* c) {
  if (!c) return fail(SART_ERR_INVALID_ARGUMENT, "ctx is NULL");
  SART_HIP(hipSetDevice(c->device));
  SART_HIP(hipStreamSynchronize(c->stream));
  return status_take(c);   // what the FIXED64 finalize kernels queued before this call found (include/sart.h "accumulation mode")
}

int sart_set_setup(sart_context* c, const sart_setup_t* s) {
  if (!c || !s) return fail(SART_ERR_INVALID_ARGUMENT, "NULL argument");
  if (s->telescope_kind != SART_TK_LLNL && s->telescope_kind != SART_TK_XMM && s->telescope_kind != SART_TK_ABRIXAS)
    return fail(SART_ERR_UNSUPPORTED,
                "telescope kind has no opaque structures / reflectivities implemented (reference asserts, raytracer.nim:1233, 1703)");
  if (s->n_shells < 9 || s->n_shells > SART_MAX_SHELLS)
    return fail(SART_ERR_INVALID_ARGUMENT, "n_shells must be in [9, 64] (allXsep[8] is read, raytracer.nim:2070)");
  for (int j = 1; j < s->n_shells; ++j)
    if (!(s->all_r1[j] > s->all_r1[j - 1])) return fail(SART_ERR_INVALID_ARGUMENT, "all_r1 must be strictly ascending");
  for (int j = 0; j + 1 < s->n_shells; ++j)
    if (!(s->all_thickness[j] >= 0.0 && s->all_r1[j] + s->all_thickness[j] < s->all_r1[j + 1]))
      return fail(SART_ERR_INVALID_ARGUMENT, "shell glass must be thinner than the spacing to the next shell");
  if (s->reflectivity_kind != SART_RK_SINGLE_COATING && s->reflectivity_kind != SART_RK_MULTI_COATING)
    return fail(SART_ERR_UNSUPPORTED, "rkEffectiveArea is unreachable in the reference (raytracer.nim:1347) and not supported");
  if (s->n_coatings < 1 || s->n_coatings > SART_MAX_COATINGS) return fail(SART_ERR_INVALID_ARGUMENT, "n_coatings out of range");
  if (s->experiment != SART_ES_CAST && s->experiment != SART_ES_BABYIAXO) return fail(SART_ERR_INVALID_ARGUMENT, "experiment");
  if (s->stage != SART_SK_VACUUM && s->stage != SART_SK_GAS) return fail(SART_ERR_INVALID_ARGUMENT, "stage");
  if (!(s->magnet_radiusCB > 0) || !(s->magnet_lengthB > 0) || !(s->magnet_lengthColdbore >= s->magnet_lengthB))
    return fail(SART_ERR_INVALID_ARGUMENT, "magnet geometry");
  if (s->number_of_strips < 0 || s->number_of_strips > 2 * kMaxStrips) return fail(SART_ERR_INVALID_ARGUMENT, "number_of_strips");
  // (the hole loop of lineIntersectsOpaqueTelescopeStructures runs number_of_holes times per ray on the device: a count from
  // uninitialised memory must not become a kernel that never ends; the reference builds 1 and 5, :1256-1346)
  if (s->hole_type < SART_HT_NONE || s->hole_type > SART_HT_DIAMOND) return fail(SART_ERR_INVALID_ARGUMENT, "hole_type");
  if (s->number_of_holes < 0 || s->number_of_holes > 64) return fail(SART_ERR_INVALID_ARGUMENT, "number_of_holes must be in [0, 64]");
  {
    // a NaN or an infinity in the geometry gives rays that fail every comparison: zero flux and no error.  Refused here by name.
    struct Field { const char* name; const double* p; int n; };
    const Field fields[] = {
        {"magnet_B", &s->magnet_B, 1}, {"magnet_lengthB", &s->magnet_lengthB, 1}, {"magnet_lengthColdbore", &s->magnet_lengthColdbore, 1},
        {"magnet_radiusCB", &s->magnet_radiusCB, 1}, {"magnet_pGasRoom", &s->magnet_pGasRoom, 1}, {"magnet_tGas", &s->magnet_tGas, 1},
        {"pipe_cb_vt3_length", &s->pipe_cb_vt3_length, 1}, {"pipe_cb_vt3_radius", &s->pipe_cb_vt3_radius, 1},
        {"pipe_vt3_xrt_length", &s->pipe_vt3_xrt_length, 1}, {"pipe_vt3_xrt_radius", &s->pipe_vt3_xrt_radius, 1},
        {"pipes_turned_deg", &s->pipes_turned_deg, 1}, {"distance_cb_axis_xrt_axis", &s->distance_cb_axis_xrt_axis, 1},
        {"optics_entrance", s->optics_entrance, 3}, {"optics_exit", s->optics_exit, 3},
        {"telescope_turned_x_deg", &s->telescope_turned_x_deg, 1}, {"telescope_turned_y_deg", &s->telescope_turned_y_deg, 1},
        {"all_r1", s->all_r1, s->n_shells}, {"all_thickness", s->all_thickness, s->n_shells}, {"all_xsep", s->all_xsep, s->n_shells},
        {"all_angles_deg", s->all_angles_deg, s->n_shells}, {"l_mirror", &s->l_mirror, 1}, {"hole_in_optics", &s->hole_in_optics, 1},
        {"distance_detector_xrt", &s->distance_detector_xrt, 1}, {"distance_window_focal_plane", &s->distance_window_focal_plane, 1},
        {"lateral_shift", &s->lateral_shift, 1}, {"transversal_shift", &s->transversal_shift, 1}, {"radius_window", &s->radius_window, 1},
        {"open_aperture_ratio", &s->open_aperture_ratio, 1}, {"strip_dist_window", &s->strip_dist_window, 1},
        {"strip_width_window", &s->strip_width_window, 1}, {"theta_rad", &s->theta_rad, 1}, {"depth_det", &s->depth_det, 1},
        {"test_energy", &s->test_energy, 1}, {"test_distance", &s->test_distance, 1}, {"test_radius", &s->test_radius, 1},
        {"test_off_axis_up", &s->test_off_axis_up, 1}, {"test_off_axis_left", &s->test_off_axis_left, 1},
        {"test_activity", &s->test_activity, 1}, {"test_length_col", &s->test_length_col, 1},
        {"distance_sun_earth", &s->distance_sun_earth, 1}, {"radius_sun", &s->radius_sun, 1}, {"room_temp", &s->room_temp, 1},
        {"m_axion", &s->m_axion, 1}, {"g_agamma", &s->g_agamma, 1}, {"chip_x_max", &s->chip_x_max, 1}, {"chip_y_max", &s->chip_y_max, 1}};
    for (const Field& f : fields)
      for (int i = 0; i < f.n; ++i)
        if (!std::isfinite(f.p[i])) return fail(SART_ERR_INVALID_ARGUMENT, std::string("setup field is not finite: ") + f.name);
  }
  c->setup = *s;
  c->have_setup = true;
  c->derived_dirty = true;
  return 0;
}

int sart_get_setup(sart_context* c, sart_setup_t* out) {
  if (!c || !out) return fail(SART_ERR_INVALID_ARGUMENT, "NULL argument");
  if (!c->have_setup) return fail(SART_ERR_NOT_READY, "no setup");
  *out = c->setup;
  return 0;
}

int sart_set_telescope_angles(sart_context* c, double tx, double ty) {
  if (!c) return fail(SART_ERR_INVALID_ARGUMENT, "ctx is NULL");
  if (!c->have_setup) return fail(SART_ERR_NOT_READY, "no setup");
  if (std::isinf(tx) || std::isinf(ty)) return fail(SART_ERR_INVALID_ARGUMENT, "telescope angle is infinite (NaN = keep)");
  if (!std::isnan(tx)) c->setup.telescope_turned_x_deg = tx;
  if (!std::isnan(ty)) c->setup.telescope_turned_y_deg = ty;
  if (c->derived_dirty) return 0;
  // cheap path: only the parameter blob changes (the shell table does not depend on the angles); sync_blob() waits
  // for the launches that still read the old blob before it uploads the new one
  c->blob_dirty = true;
  c->spot_may_have_moved = true;
  return hoist_setup(c);
}

int sart_set_axion_mass(sart_context* c, double m) {
  if (!c) return fail(SART_ERR_INVALID_ARGUMENT, "ctx is NULL");
  if (!c->have_setup) return fail(SART_ERR_NOT_READY, "no setup");
  if (!std::isfinite(m)) return fail(SART_ERR_INVALID_ARGUMENT, "axion mass is not finite");
  c->setup.m_axion = m;
  // (FIXED64, gas stage: the weight bound follows the mass - gas_prob_bound - and with it the quanta of the next launch that
  // STARTS an accumulator, accumulate == 0.  Frozen quanta are not released here: an accumulate == 1 launch adds into integers
  // that were counted in them; it is refused if the new mass's bound does not fit, and a bound that shrank far below them shows
  // up in the finalize kernel's resolution check.)
  if (c->derived_dirty) return 0;
  c->params.gas_dm2_abs = std::fabs(c->params.gas_m_gamma_sq - m * m);
  c->blob_dirty = true;
  return 0;
}

int sart_set_solar_tables(sart_context* c, const double* rcdf, const double* ecdf, const double* energies, int32_t nR,
                          int32_t nE) {
  if (!c || !rcdf || !ecdf || !energies) return fail(SART_ERR_INVALID_ARGUMENT, "NULL argument");
  if (nR < 1 || nR > 2048) return fail(SART_ERR_INVALID_ARGUMENT, "n_radii must be in [1, 2048] (LDS-resident radius CDF)");
  if (nE < 2 || nE > 65535) return fail(SART_ERR_INVALID_ARGUMENT, "n_energies must be in [2, 65535]");
  // both CDFs must be non-decreasing and end at exactly 1.0 (toCdf, raytracer.nim:2675-2677)
  for (int i = 1; i < nR; ++i) if (rcdf[i] < rcdf[i - 1]) return fail(SART_ERR_INVALID_ARGUMENT, "flux_radius_cdf not monotone");
  if (rcdf[nR - 1] != 1.0) return fail(SART_ERR_INVALID_ARGUMENT, "flux_radius_cdf must end at 1.0");
  SART_HIP(hipSetDevice(c->device));
  SART_HIP(hipStreamSynchronize(c->stream));
  // guide tables: g[k] = lowerBound(cdf, k / K); lowerBound(cdf, u) for u in [k/K, (k+1)/K) lies in [g[k], g[k+1]]
  // (radius guide: 2048 buckets, and 1024 finer ones for u >= 31/32 behind them - sart_device.h)
  std::vector<uint16_t> rg(kRadiusGuideEntries);
  for (int k = 0; k < kRadiusGuideEntries; ++k) {
    const double edge = k <= kRadiusGuide ? static_cast<double>(k) / kRadiusGuide
                                          : 0.96875 + static_cast<double>(k - (kRadiusGuide + 1)) / 32768.0;   // exact
    rg[k] = static_cast<uint16_t>(std::min<size_t>(lower_bound_idx(rcdf, nR, edge), nR - 1));
  }
  int span = 0;   // the widest bracket a draw can meet: buckets of u < 31/32, and the fine ones above
  for (int k = 0; k < kRadiusGuideEntries - 1; ++k)
    if (k < kRadiusGuide * 31 / 32 || k > kRadiusGuide) span = std::max(span, static_cast<int>(rg[k + 1]) - static_cast<int>(rg[k]));
  c->radius_span = span;
  // Energy guide (sart_device.h: kEnergyGuide*): entry k of a row brackets bucket k from below, entry k + 1 from above.
  //   k <= Uniform:   lowerBound(row, k / Div)                        buckets [k / Div, (k + 1) / Div), Uniform = Div * 31/32
  //   k = 1984 + j:   upperBound(row, 1 - decode(code0 - j + 1))      buckets (1 - v_hi, 1 - v_lo] of the codes of v = 1 - u
  // (a bucket open from below needs the upper bound of its lower edge: lowerBound(u) >= upperBound(a) for every u > a.)
  // lowerBound(row, u) of any u in bucket k then lies in [entry k, entry k + 1].  The last entry is n_energies - 1.
  auto decode = [](uint32_t code) {   // the double whose high word is code << 14
    const uint64_t bits = static_cast<uint64_t>(code) << (14 + 32);
    double d;
    std::memcpy(&d, &bits, sizeof d);
    return d;
  };
  const size_t stride = static_cast<size_t>(nE) + kEnergyCdfPad;
  std::vector<uint16_t> eg(static_cast<size_t>(nR) * kEnergyGuideEntries);
  std::vector<double> padded(static_cast<size_t>(nR) * stride, 1.0);
  for (int r = 0; r < nR; ++r) {
    const double* row = ecdf + static_cast<size_t>(r) * nE;
    for (int i = 1; i < nE; ++i)
      if (row[i] < row[i - 1]) return fail(SART_ERR_INVALID_ARGUMENT, "diff_flux_cdfs row not monotone");
    if (row[nE - 1] != 1.0) return fail(SART_ERR_INVALID_ARGUMENT, "every diff_flux_cdfs row must end at 1.0");
    uint16_t* g = eg.data() + static_cast<size_t>(r) * kEnergyGuideEntries;
    const size_t last = static_cast<size_t>(nE) - 1;
    for (int k = 0; k <= kEnergyGuideUniform; ++k)
      g[k] = static_cast<uint16_t>(std::min(lower_bound_idx(row, nE, static_cast<double>(k) / static_cast<double>(kEnergyGuideDiv)), last));
    for (int j = 1; j <= kEnergyGuideLogMax; ++j) {
      const double edge = 1.0 - decode(kEnergyGuideCode0 - static_cast<uint32_t>(j) + 1u);   // exact: v <= 1/32
      const size_t ub = static_cast<size_t>(std::upper_bound(row, row + nE, edge) - row);
      g[kEnergyGuideUniform + j] = static_cast<uint16_t>(std::min(ub, last));
    }
    g[kEnergyGuideBuckets] = static_cast<uint16_t>(last);
    std::memcpy(padded.data() + static_cast<size_t>(r) * stride, row, static_cast<size_t>(nE) * sizeof(double));
  }
  if (int rc = c->d_rcdf.upload(rcdf, nR)) return rc;
  if (int rc = c->d_ecdf.upload(padded.data(), padded.size())) return rc;
  if (int rc = c->d_ecdf_hi32.resize(padded.size())) return rc;
  launch_cdf_hi32(c->d_ecdf.p, c->d_ecdf_hi32.p, padded.size(), c->stream);   // one definition of the 32-bit table: the device kernel
  SART_HIP(hipGetLastError());
  SART_HIP(hipStreamSynchronize(c->stream));
  if (int rc = c->d_rguide.upload(rg.data(), rg.size())) return rc;
  if (int rc = c->d_eguide.upload(eg.data(), eg.size())) return rc;
  c->energies.assign(energies, energies + nE);
  c->n_radii = nR;
  c->n_energies = nE;
  c->have_solar = true;
  c->derived_dirty = true;
  return 0;
}

// The same tables as sart_set_solar_tables, built on the device from a device-resident emission-rate table
// (csrc/sart_tables.hip: bit-identical to sart_host_build_cdfs + the guide construction above).
int sart_set_solar_tables_device(sart_context* c, const double* em_rates_dev, const double* radii, const double* energies,
                                 int32_t nR, int32_t nE) {
  if (!c || !em_rates_dev || !radii || !energies) return fail(SART_ERR_INVALID_ARGUMENT, "NULL argument");
  if (nR < 1 || nR > 2048) return fail(SART_ERR_INVALID_ARGUMENT, "n_radii must be in [1, 2048] (LDS-resident radius CDF)");
  if (nE < 2 || nE > 65535) return fail(SART_ERR_INVALID_ARGUMENT, "n_energies must be in [2, 65535]");
  SART_HIP(hipSetDevice(c->device));
  SART_HIP(hipStreamSynchronize(c->stream));   // launches still in flight read the tables that are replaced below
  DevBuf<double> d_radii, d_energies, d_row_sum;
  DevBuf<uint32_t> d_status;
  if (int rc = d_radii.upload(radii, nR)) return rc;
  if (int rc = d_energies.upload(energies, nE)) return rc;
  if (int rc = d_row_sum.resize(nR)) return rc;
  if (int rc = d_status.resize(2)) return rc;
  if (int rc = c->d_rcdf.resize(nR)) return rc;
  if (int rc = c->d_ecdf.resize(static_cast<size_t>(nR) * (static_cast<size_t>(nE) + kEnergyCdfPad))) return rc;
  if (int rc = c->d_rguide.resize(kRadiusGuideEntries)) return rc;
  if (int rc = c->d_eguide.resize(static_cast<size_t>(nR) * kEnergyGuideEntries)) return rc;
  if (int rc = c->d_ecdf_hi32.resize(static_cast<size_t>(nR) * (static_cast<size_t>(nE) + kEnergyCdfPad))) return rc;
  c->have_solar = false;                       // until the new tables are known to be CDFs
  launch_build_solar_tables(em_rates_dev, d_radii.p, d_energies.p, nR, nE, c->d_ecdf.p, d_row_sum.p, c->d_rcdf.p, c->d_rguide.p,
                            c->d_eguide.p, d_status.p, c->stream);
  launch_cdf_hi32(c->d_ecdf.p, c->d_ecdf_hi32.p, c->d_ecdf_hi32.n, c->stream);
  SART_HIP(hipGetLastError());
  uint32_t status[2] = {0, 0};
  SART_HIP(hipMemcpyAsync(status, d_status.p, sizeof status, hipMemcpyDeviceToHost, c->stream));
  SART_HIP(hipStreamSynchronize(c->stream));   // also keeps the temporaries alive until the kernels are done
  if (status[0] & 1u) return fail(SART_ERR_INVALID_ARGUMENT, "emission table: a radius row does not give a CDF (negative / non-finite rates, or a row that sums to zero)");
  if (status[0] & 2u) return fail(SART_ERR_INVALID_ARGUMENT, "emission table: fluxRadiusCDF is not monotone or does not end at 1.0");
  c->radius_span = static_cast<int>(status[1]);
  c->energies.assign(energies, energies + nE);
  c->n_radii = nR;
  c->n_energies = nE;
  c->have_solar = true;
  c->derived_dirty = true;
  return 0;
}

static_assert(kEnergyGuideEntries == 2594 && kRadiusGuideEntries == 3074, "include/sart.h (sart_get_solar_tables) and _lib.py (ENERGY_GUIDE_ENTRIES, RADIUS_GUIDE_ENTRIES) state these numbers");
// Host copies of the sampling tables the context holds (whichever entry point set them).
int sart_get_solar_tables(sart_context* c, double* rcdf_out, double* ecdf_out, uint16_t* radius_guide_out, uint16_t* energy_guide_out) {
  if (!c) return fail(SART_ERR_INVALID_ARGUMENT, "ctx is NULL");
  if (!c->have_solar) return fail(SART_ERR_NOT_READY, "no solar tables");
  SART_HIP(hipSetDevice(c->device));
  SART_HIP(hipStreamSynchronize(c->stream));
  const size_t nR = static_cast<size_t>(c->n_radii), nE = static_cast<size_t>(c->n_energies);
  if (rcdf_out) SART_HIP(hipMemcpy(rcdf_out, c->d_rcdf.p, nR * sizeof(double), hipMemcpyDeviceToHost));
  if (ecdf_out)   // drop the pad behind every row
    SART_HIP(hipMemcpy2D(ecdf_out, nE * sizeof(double), c->d_ecdf.p, (nE + kEnergyCdfPad) * sizeof(double), nE * sizeof(double), nR,
                         hipMemcpyDeviceToHost));
  if (radius_guide_out) SART_HIP(hipMemcpy(radius_guide_out, c->d_rguide.p, kRadiusGuideEntries * sizeof(uint16_t), hipMemcpyDeviceToHost));
  if (energy_guide_out) SART_HIP(hipMemcpy(energy_guide_out, c->d_eguide.p, nR * kEnergyGuideEntries * sizeof(uint16_t), hipMemcpyDeviceToHost));
  return 0;
}

int sart_set_reflectivity(sart_context* c, int32_t nC, int32_t nA, int32_t nE, double amin, double amax, double emin,
                          double emax, const double* data) {
  if (!c || !data) return fail(SART_ERR_INVALID_ARGUMENT, "NULL argument");
  if (nC < 1 || nC > SART_MAX_COATINGS || nA < 2 || nE < 2 || !(amax > amin) || !(emax > emin))
    return fail(SART_ERR_INVALID_ARGUMENT, "invalid reflectivity grid");
  c->refl_data.assign(data, data + static_cast<size_t>(nC) * nA * nE);
  c->refl_nc = nC; c->refl_na = nA; c->refl_ne = nE;
  c->refl_amin = amin; c->refl_amax = amax; c->refl_emin = emin; c->refl_emax = emax;
  c->have_refl = true;
  c->derived_dirty = true;
  return 0;
}

int sart_set_detector_tables(sart_context* c, const double* sbx, const double* sby, int32_t nsb, const double* wx,
                             const double* wy, int32_t nw, const double* gx, const double* gy, int32_t ng) {
  if (!c || !sbx || !sby || !wx || !wy || !gx || !gy) return fail(SART_ERR_INVALID_ARGUMENT, "NULL argument");
  if (nsb < 2 || nw < 2 || ng < 2) return fail(SART_ERR_INVALID_ARGUMENT, "tables need at least two points");
  auto ascending = [](const double* x, int n) { for (int i = 1; i < n; ++i) if (!(x[i] > x[i - 1])) return false; return true; };
  if (!ascending(sbx, nsb) || !ascending(wx, nw) || !ascending(gx, ng))
    return fail(SART_ERR_INVALID_ARGUMENT, "table abscissae must be strictly ascending");
  c->sb_x.assign(sbx, sbx + nsb); c->sb_y.assign(sby, sby + nsb);
  c->win_x.assign(wx, wx + nw); c->win_y.assign(wy, wy + nw);
  c->gas_x.assign(gx, gx + ng); c->gas_y.assign(gy, gy + ng);
  c->have_det = true;
  c->derived_dirty = true;
  return 0;
}

static int trace_records_impl(sart_context* c, const sart_trace_params_t* p, sart_axion_t* out_dev, const double* uniforms_dev) {
  if (!c || !out_dev) return fail(SART_ERR_INVALID_ARGUMENT, "NULL argument");
  SART_HIP(hipSetDevice(c->device));
  if (int rc = refresh_derived(c)) return rc;
  if (int rc = sync_blob(c)) return rc;
  TraceArgs a;
  if (int rc = make_args(c, p, a)) return rc;
  if (a.n_rays == 0) return 0;
  if (c->blocks_per_cu_rec == 0) c->blocks_per_cu_rec = 4;
  {
    TimedLaunch tl(c);
    launch_trace_records(c->hot, c->hotb, c->d_blob.p, a, out_dev,
                         grid_for(a.n_rays, c->n_cu, c->blocks_per_cu_rec, records_block()), c->stream, uniforms_dev);
  }
  SART_HIP(hipGetLastError());
  return 0;
}

int sart_trace_records_device(sart_context* c, const sart_trace_params_t* p, sart_axion_t* out_dev) {
  return trace_records_impl(c, p, out_dev, nullptr);
}

// Test entry (not in sart.h): the records of params->n_rays rays whose six uniforms are GIVEN (uniforms_host[n][6], draw order
// of SURVEY App. B) instead of drawn from the Philox stream: tests/golden/uniform_keyed_*.npz pin the physics independently
// of the seed -> uniform mapping.  Host buffers, blocking.
__attribute__((visibility("default"))) int sart_internal_trace_records_uniforms(sart_context* c, const sart_trace_params_t* p,
                                                                               const double* uniforms_host, sart_axion_t* out_host) {
  if (!c || !p || !uniforms_host || !out_host) return fail(SART_ERR_INVALID_ARGUMENT, "NULL argument");
  if (p->n_rays == 0) return 0;
  SART_HIP(hipSetDevice(c->device));
  DevBuf<double> d_u;
  if (int rc = d_u.upload(uniforms_host, 6 * p->n_rays)) return rc;
  if (int rc = c->d_rec.reserve(p->n_rays)) return rc;
  if (int rc = trace_records_impl(c, p, c->d_rec.p, d_u.p)) return rc;
  SART_HIP(hipMemcpyAsync(out_host, c->d_rec.p, p->n_rays * sizeof(sart_axion_t), hipMemcpyDeviceToHost, c->stream));
  SART_HIP(hipStreamSynchronize(c->stream));
  return 0;
}

// Test entry (not in sart.h): the stage-A0 zone table a single launch of this context would use (words of the disc-radius stream, sart_kernels.hip).
__attribute__((visibility("default"))) int sart_internal_zones(sart_context* c, uint32_t* lo, uint32_t* hi, uint32_t* reached_bits) {
  if (!c || !lo || !hi || !reached_bits) return -1;
  if (hipSetDevice(c->device) != hipSuccess || refresh_derived(c) != 0 || sync_blob(c) != 0) return -1;
  for (int z = 0; z < c->hot.n_zones; ++z) { lo[z] = c->hot.zone_lo[z]; hi[z] = c->hot.zone_hi[z]; }
  *reached_bits = c->hot.zone_reached;
  return c->hot.n_zones;
}

// Test entry (not in sart.h): DevParams::shell0_miss_radius as the next launch would use it (-1: the shortcut is off for this setup).
__attribute__((visibility("default"))) int sart_internal_shell0_miss_radius(sart_context* c, double* out) {
  if (!c || !out) return fail(SART_ERR_INVALID_ARGUMENT, "NULL argument");
  SART_HIP(hipSetDevice(c->device));
  if (int rc = refresh_derived(c)) return rc;
  if (int rc = sync_blob(c)) return rc;
  *out = c->params.shell0_miss_radius;
  return 0;
}

// ---- the literal drop-in: records into CALLER memory ------------------------------------------------------------------------
// The kernel writes records at ~1.3 TB/s; what bounds this call is the way into the caller's buffer (raytracer.nim:2760:
// newSeq[Axion] - zero pages that nobody has touched yet).  Measured on the MI355X boxes (tools/microbench/d2h_rates*.hip,
// profiles/r03_microbench_d2h_rates.txt): PCIe D2H 57 GB/s into pinned memory, 51-55 GB/s through the runtime's staging path
// into pageable memory that is already mapped - but 10-20 GB/s into a fresh mapping, where every 4 KiB page faults on its
// first write.  Hence: (1) the interior of the caller's buffer is advised MADV_HUGEPAGE (a hint; the boxes run transparent
// huge pages in `madvise` mode) and faulted in by a few host threads ahead of the copy, one chunk at a time (2 MiB pages
// fault at 100-220 GB/s: the pre-fault of chunk k + 1 hides behind the copy of chunk k); every byte touched is a byte this
// call overwrites with records; (2) the rays are traced in chunks into two device buffers, chunk k + 1 on the context's stream
// while chunk k crosses PCIe on a second stream.  SART_NO_HOST_PREFAULT switches (1) off.
namespace {

class HostPrefault {   // faults [base, base + bytes) in, chunk by chunk, on a background thread; wait(k) blocks until chunk k is mapped
 public:
  // keep: the buffer's contents survive (every page's first byte is rewritten with itself); otherwise a zero is written - every
  // byte touched is one the caller's records overwrite
  HostPrefault(char* base, size_t bytes, size_t chunk_bytes, bool enabled, unsigned threads, bool keep = false)
      : base_(base), bytes_(bytes), chunk_(chunk_bytes), threads_(threads ? threads : 8u), keep_(keep) {
    n_chunks_ = (bytes + chunk_bytes - 1) / chunk_bytes;
    if (!enabled || bytes < (size_t(8) << 20)) { done_ = n_chunks_; return; }   // small buffers: not worth a thread
    const long page = sysconf(_SC_PAGESIZE);
    page_ = page > 0 ? static_cast<size_t>(page) : 4096;
    const uintptr_t lo = (reinterpret_cast<uintptr_t>(base) + page_ - 1) / page_ * page_;
    const uintptr_t hi = (reinterpret_cast<uintptr_t>(base) + bytes) / page_ * page_;
    if (hi > lo) (void)madvise(reinterpret_cast<void*>(lo), hi - lo, MADV_HUGEPAGE);   // a hint: failure changes nothing
    worker_ = std::thread([this] { run(); });
  }
  ~HostPrefault() {
    allow(0, true);
    if (worker_.joinable()) worker_.join();
  }
  void wait(size_t k) {
    allow(k + 1);
    std::unique_lock<std::mutex> lock(m_);
    cv_.wait(lock, [&] { return done_ > k; });
  }
  // lazy mode (sart_trace_records_passed: how much of the buffer will be written is known chunk by chunk): the worker maps
  // chunk k only once allow(> k) has been called
  void set_lazy() { std::lock_guard<std::mutex> lock(m_); allowed_ = 0; }
  void allow(size_t chunks, bool stop = false) {
    {
      std::lock_guard<std::mutex> lock(m_);
      allowed_ = std::max(allowed_, std::min(chunks, n_chunks_));
      stop_ = stop_ || stop;
    }
    cv_.notify_all();
  }

 private:
  void run() {
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const unsigned n_threads = std::min(threads_, hw);
    for (size_t k = 0; k < n_chunks_; ++k) {
      {
        std::unique_lock<std::mutex> lock(m_);
        cv_.wait(lock, [&] { return allowed_ > k || stop_; });
        if (allowed_ <= k) break;   // stopped: the chunks nobody asked for stay untouched
      }
      // whole pages inside this chunk (the partial pages at the ends of the buffer are mapped by the copy itself)
      const uintptr_t b = reinterpret_cast<uintptr_t>(base_) + k * chunk_;
      const uintptr_t e = reinterpret_cast<uintptr_t>(base_) + std::min(bytes_, (k + 1) * chunk_);
      const uintptr_t lo = (b + page_ - 1) / page_ * page_, hi = e / page_ * page_;
      if (hi > lo) {
        const size_t pages = (hi - lo) / page_, per = (pages + n_threads - 1) / n_threads;
        std::vector<std::thread> th;
        for (unsigned t = 0; t < n_threads; ++t) {
          const size_t p0 = std::min(pages, per * t), p1 = std::min(pages, per * (t + 1));
          if (p1 > p0)
            th.emplace_back([=] {
              volatile char* q = reinterpret_cast<volatile char*>(lo);
              if (keep_) for (size_t pg = p0; pg < p1; ++pg) q[pg * page_] = q[pg * page_];
              else for (size_t pg = p0; pg < p1; ++pg) q[pg * page_] = 0;   // one write per page: overwritten by the records
            });
        }
        for (auto& t : th) t.join();
      }
      {
        std::lock_guard<std::mutex> lock(m_);
        done_ = k + 1;
      }
      cv_.notify_all();
    }
  }
  char* base_;
  size_t bytes_, chunk_, n_chunks_ = 0, page_ = 4096, done_ = 0, allowed_ = ~size_t(0);
  bool stop_ = false;
  unsigned threads_;
  bool keep_ = false;
  std::mutex m_;
  std::condition_variable cv_;
  std::thread worker_;
};

}  // namespace

int sart_trace_records(sart_context* c, const sart_trace_params_t* p, sart_axion_t* out) {
  if (!c || !p || (!out && p->n_rays)) return fail(SART_ERR_INVALID_ARGUMENT, "NULL argument");
  if (p->n_rays == 0) return 0;
  SART_HIP(hipSetDevice(c->device));
  const uint64_t n = p->n_rays;
  const uint64_t chunk = c->knobs.records_chunk > 0 ? static_cast<uint64_t>(c->knobs.records_chunk) : (uint64_t(1) << 20);   // 208 MiB of records
  if (n <= chunk) {   // one launch, one copy
    if (int rc = c->d_rec.reserve(n)) return rc;
    if (int rc = sart_trace_records_device(c, p, c->d_rec.p)) return rc;
    SART_HIP(hipMemcpyAsync(out, c->d_rec.p, n * sizeof(sart_axion_t), hipMemcpyDeviceToHost, c->stream));
    SART_HIP(hipStreamSynchronize(c->stream));
    return 0;
  }
  if (int rc = c->d_rec.reserve(chunk)) return rc;
  if (int rc = c->d_rec2.reserve(chunk)) return rc;
  if (!c->copy_stream) SART_HIP(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
  for (int k = 0; k < 2; ++k) {
    if (!c->rec_traced[k]) SART_HIP(hipEventCreateWithFlags(&c->rec_traced[k], hipEventDisableTiming));
    if (!c->rec_copied[k]) SART_HIP(hipEventCreateWithFlags(&c->rec_copied[k], hipEventDisableTiming));
  }
  sart_axion_t* const bufs[2] = {c->d_rec.p, c->d_rec2.p};
  const uint64_t n_chunks = (n + chunk - 1) / chunk;
  HostPrefault prefault(reinterpret_cast<char*>(out), n * sizeof(sart_axion_t), chunk * sizeof(sart_axion_t), !c->knobs.no_host_prefault,
                        static_cast<unsigned>(c->knobs.prefault_threads));
  sart_trace_params_t q = *p;
  int rc = 0;
  // A failure inside the pipeline must not leave through an early return: kernels and copies into the CALLER's buffer are queued
  // on two streams, and the caller may free or reuse that buffer the moment this call returns.  Every step records its error
  // and breaks; both streams are synchronised on every way out (the HostPrefault destructor joins the pre-fault threads).
  hipError_t he = hipSuccess;
  const char* what = "";
  auto step = [&](hipError_t e, const char* name) {
    if (e != hipSuccess && he == hipSuccess) { he = e; what = name; }
    return e == hipSuccess;
  };
  for (uint64_t k = 0; k <= n_chunks && rc == 0 && he == hipSuccess; ++k) {
    if (k < n_chunks) {   // trace chunk k into buffer k & 1 once the copy of chunk k - 2 has left it
      if (k >= 2 && !step(hipStreamWaitEvent(c->stream, c->rec_copied[k & 1], 0), "hipStreamWaitEvent(stream)")) break;
      q.n_rays = std::min(chunk, n - k * chunk);
      q.ray_id_offset = p->ray_id_offset + k * chunk;
      rc = (c->knobs.records_fail_chunk > 0 && k + 1 == static_cast<uint64_t>(c->knobs.records_fail_chunk))
               ? fail(SART_ERR_INTERNAL, "sart_trace_records: failure injected by SART_RECORDS_FAIL_CHUNK (test hook)")
               : sart_trace_records_device(c, &q, bufs[k & 1]);
      if (rc) break;
      if (!step(hipEventRecord(c->rec_traced[k & 1], c->stream), "hipEventRecord(traced)")) break;
    }
    if (k > 0) {          // chunk k - 1 crosses PCIe while chunk k is traced (the call may block the host: pageable destination)
      const uint64_t j = k - 1, cnt = std::min(chunk, n - j * chunk);
      prefault.wait(j);
      if (!step(hipStreamWaitEvent(c->copy_stream, c->rec_traced[j & 1], 0), "hipStreamWaitEvent(copy_stream)")) break;
      if (!step(hipMemcpyAsync(out + j * chunk, bufs[j & 1], cnt * sizeof(sart_axion_t), hipMemcpyDeviceToHost, c->copy_stream), "hipMemcpyAsync(records)")) break;
      if (!step(hipEventRecord(c->rec_copied[j & 1], c->copy_stream), "hipEventRecord(copied)")) break;
    }
  }
  const hipError_t e1 = hipStreamSynchronize(c->copy_stream), e2 = hipStreamSynchronize(c->stream);
  if (rc) return rc;
  if (he != hipSuccess) return fail(SART_ERR_NO_DEVICE, std::string("sart_trace_records: ") + what + ": " + hipGetErrorString(he));
  if (e1 != hipSuccess || e2 != hipSuccess) return fail(SART_ERR_NO_DEVICE, std::string("sart_trace_records: ") + hipGetErrorString(e1 != hipSuccess ? e1 : e2));
  return 0;
}

// ---- passed rays only ----------------------------------------------------------------------------------------------------------
// The consumers of the record buffer (generateResultPlots raytracer.nim:2252-2283, the scan sum :2800) filter on `passed` and
// count two flags; BabyIAXO passes 21 % of its rays, so 79 % of what sart_trace_records moves across PCIe is never read.  Here
// the records of a chunk are traced into device scratch as before, compacted in ray order on the device (sart_kernels.hip:
// records_count / scan / scatter), and only the passed ones travel.
namespace {
int compact_scratch(sart_context* c, uint64_t chunk) {
  if (int rc = c->d_rec.reserve(chunk)) return rc;
  if (int rc = c->d_cmp_counts.resize(1024)) return rc;
  if (int rc = c->d_cmp_first.resize(1024)) return rc;
  return 0;
}
uint64_t compact_chunk_of(const sart_context* c) {
  const uint64_t cap = static_cast<uint64_t>(compact_chunk_max());
  return c->knobs.records_chunk > 0 ? std::min<uint64_t>(static_cast<uint64_t>(c->knobs.records_chunk), cap) : cap;
}
}  // namespace

int sart_trace_records_passed_device(sart_context* c, const sart_trace_params_t* p, sart_axion_t* out_dev, uint64_t capacity, uint64_t* counts_dev) {
  if (!c || !p || !counts_dev || (!out_dev && capacity)) return fail(SART_ERR_INVALID_ARGUMENT, "NULL argument");
  SART_HIP(hipSetDevice(c->device));
  if (!p->accumulate) SART_HIP(hipMemsetAsync(counts_dev, 0, 4 * sizeof(uint64_t), c->stream));
  if (p->n_rays == 0) return 0;
  const uint64_t chunk = compact_chunk_of(c);
  if (int rc = compact_scratch(c, std::min<uint64_t>(chunk, p->n_rays))) return rc;
  sart_trace_params_t q = *p;
  for (uint64_t done = 0; done < p->n_rays; done += chunk) {
    q.n_rays = std::min(chunk, p->n_rays - done);
    q.ray_id_offset = p->ray_id_offset + done;
    if (int rc = sart_trace_records_device(c, &q, c->d_rec.p)) return rc;
    launch_compact_records(c->d_rec.p, static_cast<uint32_t>(q.n_rays), out_dev, capacity, c->d_cmp_counts.p, c->d_cmp_first.p,
                           reinterpret_cast<unsigned long long*>(counts_dev), c->stream);
    SART_HIP(hipGetLastError());
  }
  return 0;
}

int sart_trace_records_passed(sart_context* c, const sart_trace_params_t* p, sart_axion_t* out, uint64_t capacity, sart_record_counts_t* counts) {
  if (!c || !p || !counts || (!out && capacity)) return fail(SART_ERR_INVALID_ARGUMENT, "NULL argument");
  std::memset(counts, 0, sizeof *counts);
  if (p->n_rays == 0) return 0;
  SART_HIP(hipSetDevice(c->device));
  const uint64_t n = p->n_rays, chunk = compact_chunk_of(c), n_chunks = (n + chunk - 1) / chunk;
  if (int rc = compact_scratch(c, std::min(chunk, n))) return rc;
  // two compacted half-buffers (one when a single chunk does it): the record path's second trace buffer and one of this path's
  // own, both grow-only and kept until sart_release_scratch / sart_destroy (include/sart.h: "device scratch")
  DevBuf<sart_axion_t>* const cmp[2] = {&c->d_rec2, &c->d_cmp};
  for (int k = 0; k < (n_chunks > 1 ? 2 : 1); ++k)
    if (int rc = cmp[k]->reserve(std::min(chunk, n))) return rc;
  if (int rc = c->d_cmp_totals.resize(8)) return rc;
  if (!c->h_cmp_totals) SART_HIP(hipHostMalloc(reinterpret_cast<void**>(&c->h_cmp_totals), 8 * sizeof(unsigned long long), hipHostMallocDefault));
  if (!c->copy_stream) SART_HIP(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
  for (int k = 0; k < 2; ++k) {
    if (!c->rec_traced[k]) SART_HIP(hipEventCreateWithFlags(&c->rec_traced[k], hipEventDisableTiming));
    if (!c->rec_copied[k]) SART_HIP(hipEventCreateWithFlags(&c->rec_copied[k], hipEventDisableTiming));
  }
  // (pages of `out` beyond the records written keep what they held: the pre-fault rewrites every page's first byte with itself)
  HostPrefault prefault(reinterpret_cast<char*>(out), std::min(capacity, n) * sizeof(sart_axion_t), chunk * sizeof(sart_axion_t),
                        !c->knobs.no_host_prefault, static_cast<unsigned>(c->knobs.prefault_threads), /*keep=*/true);
  prefault.set_lazy();
  sart_trace_params_t q = *p;
  int rc = 0;
  hipError_t he = hipSuccess;   // as in sart_trace_records: no early return while work on the caller's buffer is queued
  const char* what = "";
  auto step = [&](hipError_t e, const char* name) {
    if (e != hipSuccess && he == hipSuccess) { he = e; what = name; }
    return e == hipSuccess;
  };
  uint64_t written = 0;
  for (uint64_t k = 0; k <= n_chunks && rc == 0 && he == hipSuccess; ++k) {
    if (k < n_chunks) {   // trace + compact chunk k into half-buffer k & 1 once the copy of chunk k - 2 has left it
      const int b = static_cast<int>(k & 1);
      if (k >= 2 && !step(hipStreamWaitEvent(c->stream, c->rec_copied[b], 0), "hipStreamWaitEvent(stream)")) break;
      q.n_rays = std::min(chunk, n - k * chunk);
      q.ray_id_offset = p->ray_id_offset + k * chunk;
      unsigned long long* totals = c->d_cmp_totals.p + 4 * b;
      if (!step(hipMemsetAsync(totals, 0, 4 * sizeof(unsigned long long), c->stream), "hipMemsetAsync(counts)")) break;
      rc = (c->knobs.records_fail_chunk > 0 && k + 1 == static_cast<uint64_t>(c->knobs.records_fail_chunk))
               ? fail(SART_ERR_INTERNAL, "sart_trace_records_passed: failure injected by SART_RECORDS_FAIL_CHUNK (test hook)")
               : sart_trace_records_device(c, &q, c->d_rec.p);
      if (rc) break;
      launch_compact_records(c->d_rec.p, static_cast<uint32_t>(q.n_rays), cmp[b]->p, q.n_rays, c->d_cmp_counts.p, c->d_cmp_first.p, totals, c->stream);
      if (!step(hipGetLastError(), "compaction kernels")) break;
      if (!step(hipMemcpyAsync(c->h_cmp_totals + 4 * b, totals, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream), "hipMemcpyAsync(counts)")) break;
      if (!step(hipEventRecord(c->rec_traced[b], c->stream), "hipEventRecord(traced)")) break;
    }
    if (k > 0) {          // the passed records of chunk k - 1 cross PCIe while chunk k is traced
      const int b = static_cast<int>((k - 1) & 1);
      if (!step(hipEventSynchronize(c->rec_traced[b]), "hipEventSynchronize(traced)")) break;
      const unsigned long long* t = c->h_cmp_totals + 4 * b;
      counts->n_rays += t[0]; counts->n_passed += t[1]; counts->n_passed_till_window += t[2]; counts->n_hit_nickel += t[3];
      const uint64_t room = capacity - written, cnt = std::min<uint64_t>(t[1], room);
      if (cnt) {
        prefault.allow((written + 3 * cnt) / chunk + 1);   // this copy and, at this pass rate, the next two
        for (uint64_t j = written / chunk; j <= (written + cnt - 1) / chunk; ++j) prefault.wait(j);   // (its chunks: `chunk` records each)
        if (!step(hipMemcpyAsync(out + written, cmp[b]->p, cnt * sizeof(sart_axion_t), hipMemcpyDeviceToHost, c->copy_stream), "hipMemcpyAsync(records)")) break;
        written += cnt;
      }
      if (!step(hipEventRecord(c->rec_copied[b], c->copy_stream), "hipEventRecord(copied)")) break;
    }
  }
  const hipError_t e1 = hipStreamSynchronize(c->copy_stream), e2 = hipStreamSynchronize(c->stream);
  if (rc) return rc;
  if (he != hipSuccess) return fail(SART_ERR_NO_DEVICE, std::string("sart_trace_records_passed: ") + what + ": " + hipGetErrorString(he));
  if (e1 != hipSuccess || e2 != hipSuccess) return fail(SART_ERR_NO_DEVICE, std::string("sart_trace_records_passed: ") + hipGetErrorString(e1 != hipSuccess ? e1 : e2));
  return 0;
}

int sart_release_scratch(sart_context* c) {
  if (!c) return fail(SART_ERR_INVALID_ARGUMENT, "ctx is NULL");
  SART_HIP(hipSetDevice(c->device));
  SART_HIP(hipStreamSynchronize(c->stream));
  if (c->copy_stream) SART_HIP(hipStreamSynchronize(c->copy_stream));
  c->d_rec.release();
  c->d_rec2.release();
  c->d_cmp.release();
  return 0;
}

int sart_trace_histogram_device(sart_context* c, const sart_trace_params_t* p, double* acc_dev) {
  if (!c || !acc_dev) return fail(SART_ERR_INVALID_ARGUMENT, "NULL argument");
  SART_HIP(hipSetDevice(c->device));
  if (int rc = refresh_derived(c)) return rc;
  if (int rc = sync_blob(c)) return rc;
  TraceArgs a;
  if (int rc = make_args(c, p, a)) return rc;
  // SART_ACCUM_FIXED64 (the pilot launch that places the LDS tile reads f64 sums and always runs in f64)
  const bool fixed = c->accum_mode == SART_ACCUM_FIXED64 && !c->tile.in_pilot;
  a.fx_scale_w = a.fx_scale_w2 = 0.0;
  if (fixed) {
    if (int rc = freeze_quanta(c, weight_bound_of(c, p->flags, c->params.gas_dm2_abs), !p->accumulate)) return rc;
    a.fx_scale_w = std::ldexp(1.0, -c->weight_exp);
    a.fx_scale_w2 = std::ldexp(1.0, -c->weight_sq_exp);
  }
  if (!p->accumulate) SART_HIP(hipMemsetAsync(acc_dev, 0, acc_len_of(c, p) * sizeof(double), c->stream));
  if (a.n_rays == 0) return 0;
  if (a.n_rays > (1ull << 31)) {   // ray indices inside one launch are 32-bit (stage A0 ring): split
    sart_trace_params_t q = *p;
    q.accumulate = 1;
    for (uint64_t done = 0; done < p->n_rays;) {
      const uint64_t n = std::min<uint64_t>(p->n_rays - done, 1ull << 31);
      q.n_rays = n;
      q.ray_id_offset = p->ray_id_offset + done;
      if (int rc = sart_trace_histogram_device(c, &q, acc_dev)) return rc;
      done += n;
    }
    return 0;
  }
  const int variant = hist_variant_of(c);
  {
    // Replication factor from the expected size of the solar image in pixels: plate scale (distance XRT -> detector)
    // times the angular radius of the emitting core (~0.25 R_sun).  Scattered f64 atomics execute at the memory side
    // (TCC_EA0_ATOMIC = every atomic) at ~2.4e10 lane-operations / s for the whole chip whatever the type or scope
    // (tools/microbench/atomic_rates.hip), and the hot pixels of one image serialise there: wide images (BabyIAXO / XMM:
    // f = 7.5 m) take 8 replicas (measured 1 / 2 / 4 / 8 / 16 / 32 / 128 replicas: 1.87 / 1.78 / 1.77 / 1.74-1.79 / 1.78 /
    // 1.78 / 1.79 ms per 1e8 rays; 1.67 ms with the atomics switched off), small focal spots 64.
    const sart_setup_t& s = c->setup;
    const double spot_px = s.distance_detector_xrt * (0.25 * s.radius_sun / s.distance_sun_earth) *
                           a.image_inv_step_x;
    // Measured on CAST / LLNL (88 % of all rays land within ~30 pixels): 1 replica 12.4 ms per 1e8 rays, 16: 5.5, 64: 4.4,
    // 128: 4.4, 512: 4.6 (3.99 ms with the atomics switched off).
    int R = spot_px > 96.0 ? 8 : 64;
    if (static_cast<size_t>(p->image_nx) * static_cast<size_t>(p->image_ny) > (1u << 20)) R = std::min(R, 1);   // heat maps of millions of pixels: no scratch copies
    if (s.test_active) R = 64;
    if (c->knobs.image_replicas > 0) R = std::min(kMaxImageReplicas, c->knobs.image_replicas);
    if (c->tile.in_pilot) R = 64;     // one-pixel image
    if (p->image_nx == 0) R = 1;      // flux-only launch: no image, so no scratch copies, no LDS tile, no pilot launch - whatever the source or the knobs say
    while (R & (R - 1)) R &= R - 1;   // power of two
    if (R > 1) {
      const size_t n_img = static_cast<size_t>(p->image_nx) * static_cast<size_t>(p->image_ny);
      const size_t pad = 32;   // replicas do not start on the same power-of-two boundary (measured: within noise)
      if (n_img + pad > 0xFFFFFFFFull) return fail(SART_ERR_INVALID_ARGUMENT, "image too large for replicated accumulation");
      a.replica_stride = static_cast<uint32_t>(n_img + pad);
      const size_t need = static_cast<size_t>(R) * (n_img + pad);
      // the pilot launch (one pixel) has scratch images of its own, so the caller's buffer keeps its layout
      DevBuf<double>& reps = c->tile.in_pilot ? c->d_pilot_replicas : c->d_replicas;
      if (reps.n != need || !reps.p) {
        SART_HIP(hipStreamSynchronize(c->stream));
        if (int rc = reps.resize(need)) return rc;
        // (on the context's stream, in front of the launch that adds into them: a memset on the null stream is not ordered with
        // launches on a non-blocking stream)
        SART_HIP(hipMemsetAsync(reps.p, 0, need * sizeof(double), c->stream));
      }
      a.replicas = reps.p;
      a.replica_mask = static_cast<uint32_t>(R - 1);
      // No stage A0 (its ring space in LDS is free) or the constant-path variant (the path column of ring 1 is free):
      // accumulate the centre of the spot in a per-workgroup LDS tile (56 x 56: CAST / LLNL 82 % of the hits, BabyIAXO / XMM 34 %;
      // 33 x 33 behind the tables alone for the variants whose rings are all in use: stage A0 on and the path carried).
      // The tile is centred on the spot's centroid, measured once per setup and image binning by a pilot launch of 2e5 rays
      // into a one-pixel image (only SUM_X / SUM_Y / N_PASSED are read).
      const bool ring_cells_free = c->hot.n_zones == 0 || variant == 5 || variant == 6;
      if (!c->knobs.no_image_tile && !c->tile.in_pilot) {
        sart_context::TileCache& t = c->tile;
        const bool same = t.valid && t.nx == p->image_nx && t.ny == p->image_ny && t.x_min == p->image_x_min &&
                          t.x_max == p->image_x_max && t.y_min == p->image_y_min && t.y_max == p->image_y_max &&
                          t.ring_cells_free == ring_cells_free;
        if (!same) {
          t.valid = false;
          t.n = 0;
          sart_trace_params_t q = *p;
          q.n_rays = 200000;   // always the same sample, whatever the size of the call that triggers it (its first ray ids)
          q.image_nx = q.image_ny = 1;
          q.accumulate = 0;
          q.spectra = 0;
          if (int rc = c->d_pilot.resize(sart_accumulator_len(1, 1))) return rc;
          t.in_pilot = true;
          const bool was_timing = c->timing;
          c->timing = false;   // the pilot is not one of the caller's launches
          const int rc = sart_trace_histogram_device(c, &q, c->d_pilot.p);
          c->timing = was_timing;
          t.in_pilot = false;
          if (rc) return rc;
          double sc[SART_ACC_COUNT];
          SART_HIP(hipMemcpyAsync(sc, c->d_pilot.p + 1, sizeof sc, hipMemcpyDeviceToHost, c->stream));
          SART_HIP(hipStreamSynchronize(c->stream));
          t.nx = p->image_nx; t.ny = p->image_ny;
          t.x_min = p->image_x_min; t.x_max = p->image_x_max; t.y_min = p->image_y_min; t.y_max = p->image_y_max;
          t.ring_cells_free = ring_cells_free;
          if (sc[SART_ACC_N_PASSED] >= 100.0) {
            const double cx = (sc[SART_ACC_SUM_X] / sc[SART_ACC_N_PASSED] - p->image_x_min) * a.image_inv_step_x;
            const double cy = (sc[SART_ACC_SUM_Y] / sc[SART_ACC_N_PASSED] - p->image_y_min) * a.image_inv_step_y;
            const int n = std::min({static_cast<int>(ring_cells_free ? kImageTileMax : kImageTileExtraMax), p->image_nx, p->image_ny});
            const int x0 = std::clamp(static_cast<int>(std::floor(cx)) - n / 2, 0, p->image_nx - n);
            const int y0 = std::clamp(static_cast<int>(std::floor(cy)) - n / 2, 0, p->image_ny - n);
            t.x0 = x0; t.y0 = y0; t.n = n;
          }
          t.valid = true;
        }
        a.tile_x0 = t.x0; a.tile_y0 = t.y0; a.tile_n = t.n;
        a.tile_base = ring_cells_free ? 0 : kTileRingCells;
      }
    } else {
      a.replicas = acc_dev;
      a.replica_mask = 0u;
    }
  }
  if (c->blocks_per_cu_hist[variant] == 0) {
    c->blocks_per_cu_hist[variant] = std::max(1, histogram_blocks_per_cu(variant));
    if (c->knobs.hist_blocks_per_cu > 0) c->blocks_per_cu_hist[variant] = c->knobs.hist_blocks_per_cu;
  }
  const int n_blocks = grid_for(a.n_rays, c->n_cu, c->blocks_per_cu_hist[variant], histogram_block_of(variant));
  if (c->d_partials.n < static_cast<size_t>(n_blocks) * SART_ACC_COUNT) {   // one row of scalars per workgroup
    SART_HIP(hipStreamSynchronize(c->stream));
    const size_t rows = std::max<size_t>(static_cast<size_t>(n_blocks), static_cast<size_t>(c->n_cu) * 4);
    if (int rc = c->d_partials.resize(rows * SART_ACC_COUNT)) return rc;
  }
  a.partials = c->d_partials.p;
  {
    TimedLaunch tl(c);
    launch_trace_histogram(c->hot, c->hotb, c->d_blob.p, a, acc_dev, n_blocks, c->stream, variant, fixed);
  }
  SART_HIP(hipGetLastError());
  return 0;
}

int sart_trace_histogram(sart_context* c, const sart_trace_params_t* p, double* image_out, sart_summary_t* summary) {
  return sart_trace_histogram_spectra(c, p, image_out, summary, nullptr);
}

int sart_trace_histogram_spectra(sart_context* c, const sart_trace_params_t* p, double* image_out, sart_summary_t* summary,
                                 double* spectra_out) {
  if (!c || !p) return fail(SART_ERR_INVALID_ARGUMENT, "NULL argument");
  SART_HIP(hipSetDevice(c->device));
  if ((p->image_nx < 1 || p->image_ny < 1) && !(p->image_nx == 0 && p->image_ny == 0)) return fail(SART_ERR_INVALID_ARGUMENT, "invalid image specification");
  if (p->spectra && (p->n_radial_bins < 1 || !c->have_solar)) return fail(SART_ERR_INVALID_ARGUMENT, "invalid spectra specification");
  const size_t len = acc_len_of(c, p);
  const bool fresh = (c->d_acc.n != len) || !c->d_acc.p || c->d_acc_stale;
  if (int rc = c->d_acc.resize(len)) return rc;
  sart_trace_params_t q = *p;
  if (fresh) q.accumulate = 0;
  if (int rc = sart_trace_histogram_device(c, &q, c->d_acc.p)) return rc;
  c->d_acc_stale = false;
  const size_t nimg = static_cast<size_t>(p->image_nx) * p->image_ny;
  const double* src = c->d_acc.p;
  if (c->accum_mode == SART_ACCUM_FIXED64) {   // the scratch accumulator holds integers: hand out its f64 image
    if (int rc = c->d_fin.resize(len)) return rc;
    if (int rc = sart_finalize_accumulator_device(c, p, c->d_acc.p, c->d_fin.p)) return rc;
    src = c->d_fin.p;
  }
  if (image_out && nimg) SART_HIP(hipMemcpyAsync(image_out, src, nimg * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  if (summary)
    SART_HIP(hipMemcpyAsync(summary->v, src + nimg, SART_ACC_COUNT * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  if (spectra_out && p->spectra)
    SART_HIP(hipMemcpyAsync(spectra_out, src + nimg + SART_ACC_COUNT, (len - nimg - SART_ACC_COUNT) * sizeof(double),
                            hipMemcpyDeviceToHost, c->stream));
  SART_HIP(hipStreamSynchronize(c->stream));
  // FIXED64: the quantum is derived from a BOUND of the weights.  If what was accumulated averages below 2^12 quanta per ray
  // (tables or flags the bound does not see through), or a slot wrapped, the finalize kernel has said so: an error, not noise.
  return status_take(c);
}

int sart_set_accumulation_mode(sart_context* c, int mode, int headroom_bits) {
  if (!c) return fail(SART_ERR_INVALID_ARGUMENT, "ctx is NULL");
  if (mode != SART_ACCUM_F64 && mode != SART_ACCUM_FIXED64) return fail(SART_ERR_INVALID_ARGUMENT, "unknown accumulation mode");
  if (headroom_bits != 0 && (headroom_bits < 16 || headroom_bits > 44))
    return fail(SART_ERR_INVALID_ARGUMENT, "headroom_bits must be 0 (default 27) or in [16, 44]");
  SART_HIP(hipSetDevice(c->device));
  SART_HIP(hipStreamSynchronize(c->stream));
  const int headroom = headroom_bits ? headroom_bits : 27;
  if (mode == c->accum_mode && headroom == c->headroom_bits) return 0;   // nothing changes: the frozen quanta stay (an accumulator may hold data in them)
  c->d_acc_stale = true;   // the scratch accumulator of the blocking calls holds data of the old mode / quanta
  c->accum_mode = mode;
  c->headroom_bits = headroom;
  c->quanta_frozen = false;
  return 0;
}

int sart_get_accumulation_mode(sart_context* c, int* mode_out) {
  if (!c || !mode_out) return fail(SART_ERR_INVALID_ARGUMENT, "NULL argument");
  *mode_out = c->accum_mode;
  return 0;
}

int sart_get_fixed_quanta(sart_context* c, sart_fixed_quanta_t* out) {
  if (!c || !out) return fail(SART_ERR_INVALID_ARGUMENT, "NULL argument");
  if (!c->quanta_frozen) return fail(SART_ERR_NOT_READY, "no FIXED64 launch has fixed the quanta yet");
  out->weight = std::ldexp(1.0, c->weight_exp);
  out->weight_sq = std::ldexp(1.0, c->weight_sq_exp);
  out->position = 1.0 / kFixedPositionScale;
  out->reflect = 1.0 / kFixedReflectScale;
  return 0;
}

int sart_finalize_accumulator_device(sart_context* c, const sart_trace_params_t* p, const void* acc_fixed_dev, double* out_dev) {
  return sart_finalize_accumulator_limbs_device(c, p, acc_fixed_dev, nullptr, out_dev);
}

int sart_rollover_accumulator_device(sart_context* c, const sart_trace_params_t* p, void* acc_fixed_dev, void* hi_limbs_dev) {
  if (!c || !p || !acc_fixed_dev || !hi_limbs_dev) return fail(SART_ERR_INVALID_ARGUMENT, "NULL argument");
  if (((p->image_nx < 1 || p->image_ny < 1) && !(p->image_nx == 0 && p->image_ny == 0)) || (p->spectra && p->n_radial_bins < 1))
    return fail(SART_ERR_INVALID_ARGUMENT, "invalid image specification");
  if (acc_fixed_dev == hi_limbs_dev) return fail(SART_ERR_INVALID_ARGUMENT, "the limbs need an array of their own");
  if (c->accum_mode != SART_ACCUM_FIXED64)
    return fail(SART_ERR_INVALID_ARGUMENT, "sart_rollover_accumulator_device folds the int64 slots of a raw SART_ACCUM_FIXED64 accumulator; the "
                                           "context is in SART_ACCUM_F64 mode (its accumulators hold doubles)");
  SART_HIP(hipSetDevice(c->device));
  if (int rc = status_ensure(c)) return rc;
  launch_rollover_fixed(acc_fixed_dev, hi_limbs_dev, acc_len_of(c, p), c->d_status.p, c->stream);
  SART_HIP(hipGetLastError());
  return status_enqueue_copy(c);
}

int sart_finalize_accumulator_limbs_device(sart_context* c, const sart_trace_params_t* p, const void* acc_fixed_dev, const void* hi_limbs_dev,
                                           double* out_dev) {
  if (!c || !p || !acc_fixed_dev || !out_dev) return fail(SART_ERR_INVALID_ARGUMENT, "NULL argument");
  if (hi_limbs_dev && (hi_limbs_dev == acc_fixed_dev || hi_limbs_dev == static_cast<const void*>(out_dev)))
    return fail(SART_ERR_INVALID_ARGUMENT, "the limbs need an array of their own");
  if (((p->image_nx < 1 || p->image_ny < 1) && !(p->image_nx == 0 && p->image_ny == 0)) || (p->spectra && p->n_radial_bins < 1))
    return fail(SART_ERR_INVALID_ARGUMENT, "invalid image specification");
  if (!c->quanta_frozen) return fail(SART_ERR_NOT_READY, "no FIXED64 launch has fixed the quanta yet");
  SART_HIP(hipSetDevice(c->device));
  if (int rc = status_ensure(c)) return rc;
  launch_finalize_fixed(acc_fixed_dev, hi_limbs_dev, out_dev, static_cast<size_t>(p->image_nx) * static_cast<size_t>(p->image_ny), p->spectra ? 1 : 0,
                        p->spectra ? p->n_radial_bins : 0, c->n_energies + 1, std::ldexp(1.0, c->weight_exp),
                        std::ldexp(1.0, c->weight_sq_exp), 1.0 / kFixedPositionScale, 1.0 / kFixedReflectScale, c->d_status.p, c->stream);
  SART_HIP(hipGetLastError());
  return status_enqueue_copy(c);
}

// ---- fused axion-mass scan (include/sart.h) -----------------------------------------------------------------------------------
namespace {

int scan_check(sart_context* c, const sart_trace_params_t* p, const double* masses, int32_t n) {
  if (!c || !p || !masses) return fail(SART_ERR_INVALID_ARGUMENT, "NULL argument");
  if (n < 1 || n > 65536) return fail(SART_ERR_INVALID_ARGUMENT, "n_masses must be in [1, 65536]");
  for (int32_t k = 0; k < n; ++k)
    if (!std::isfinite(masses[k]) || masses[k] < 0.0) return fail(SART_ERR_INVALID_ARGUMENT, "axion masses must be finite and >= 0");
  return 0;
}

// |m_gamma^2 - m_a^2| exactly as hoist_setup / sart_set_axion_mass compute it for the single-mass kernels
double dm2_abs_of(const sart_context* c, double m_axion) { return std::fabs(c->params.gas_m_gamma_sq - m_axion * m_axion); }

}  // namespace

int sart_trace_mass_scan_device(sart_context* c, const sart_trace_params_t* p, const double* masses, int32_t n_masses, double* scan_dev) {
  if (int rc = scan_check(c, p, masses, n_masses)) return rc;
  if (!scan_dev) return fail(SART_ERR_INVALID_ARGUMENT, "scan accumulator is NULL");
  SART_HIP(hipSetDevice(c->device));
  if (int rc = refresh_derived(c)) return rc;
  if (int rc = sync_blob(c)) return rc;
  if (!c->params.stage_gas)
    return fail(SART_ERR_INVALID_ARGUMENT, "sart_trace_mass_scan: the setup's stage is vacuum, whose conversion probability "
                                           "(raytracer.nim:363-365) does not depend on the axion mass");
  sart_trace_params_t q = *p;   // a scan accumulates no image: whatever the caller left in the image fields is not read
  q.image_nx = q.image_ny = 1;
  q.image_x_min = q.image_y_min = 0.0;
  q.image_x_max = q.image_y_max = 1.0;
  q.spectra = 0;
  TraceArgs a;
  if (int rc = make_args(c, &q, a)) return rc;
  a.fx_scale_w = a.fx_scale_w2 = 0.0;
  const bool fixed = c->accum_mode == SART_ACCUM_FIXED64;
  if (!p->accumulate) SART_HIP(hipMemsetAsync(scan_dev, 0, sart_mass_scan_len(n_masses) * sizeof(double), c->stream));
  if (p->n_rays == 0) return 0;
  const int variant = hist_variant_of(c);
  if (variant != 1 && variant != 2 && variant != 3 && variant != 6) return fail(SART_ERR_INTERNAL, "no scan kernel for this variant");
  if (c->blocks_per_cu_hist[variant] == 0) {
    c->blocks_per_cu_hist[variant] = std::max(1, histogram_blocks_per_cu(variant));
    if (c->knobs.hist_blocks_per_cu > 0) c->blocks_per_cu_hist[variant] = c->knobs.hist_blocks_per_cu;
  }
  // ray indices inside one launch are 32-bit: pieces of at most 2^31 rays; every piece runs once per group of masses
  for (uint64_t done = 0; done < p->n_rays;) {
    const uint64_t n = std::min<uint64_t>(p->n_rays - done, 1ull << 31);
    a.n_rays = n;
    a.ray_id_offset = p->ray_id_offset + done;
    const int n_blocks = grid_for(n, c->n_cu, c->blocks_per_cu_hist[variant], histogram_block_of(variant));
    if (c->d_partials.n < static_cast<size_t>(n_blocks) * SART_ACC_COUNT) {
      SART_HIP(hipStreamSynchronize(c->stream));
      const size_t rows = std::max<size_t>(static_cast<size_t>(n_blocks), static_cast<size_t>(c->n_cu) * 4);
      if (int rc = c->d_partials.resize(rows * SART_ACC_COUNT)) return rc;
    }
    const size_t scan_partials = static_cast<size_t>(n_blocks) * kScanMaxMasses * kScanPartialSlots;
    if (c->d_scan_partials.n < scan_partials) {
      SART_HIP(hipStreamSynchronize(c->stream));
      const size_t rows = std::max<size_t>(static_cast<size_t>(n_blocks), static_cast<size_t>(c->n_cu) * 4);
      if (int rc = c->d_scan_partials.resize(rows * kScanMaxMasses * kScanPartialSlots)) return rc;
    }
    a.partials = c->d_partials.p;
    for (int32_t k0 = 0; k0 < n_masses; k0 += kScanMaxMasses) {
      ScanArgs sc;
      std::memset(&sc, 0, sizeof sc);
      sc.n_masses = std::min<int32_t>(kScanMaxMasses, n_masses - k0);
      sc.partials = c->d_scan_partials.p;
      for (int k = 0; k < sc.n_masses; ++k) {
        ScanMass& m = sc.m[k];
        m.dm2_abs = dm2_abs_of(c, masses[k0 + k]);
        m.fx_scale_w = m.fx_scale_w2 = 0.0;
        if (fixed) {
          QuantaExp qe;
          if (int rc = quanta_exp_of(c, weight_bound_of(c, p->flags, m.dm2_abs), qe)) return rc;
          m.fx_scale_w = std::ldexp(1.0, -qe.w);
          m.fx_scale_w2 = std::ldexp(1.0, -qe.w2);
        }
      }
      double* const rows = scan_dev + static_cast<size_t>(k0) * SART_SCAN_ROW;
      double* const shared = (k0 == 0) ? scan_dev + static_cast<size_t>(n_masses) * SART_SCAN_ROW : nullptr;   // counters: once per piece
      {
        TimedLaunch tl(c);
        if (!launch_trace_mass_scan(c->hot, c->hotb, c->d_blob.p, a, sc, rows, shared, n_blocks, c->stream, variant, fixed))
          return fail(SART_ERR_INTERNAL, "no scan kernel for this variant");
      }
      SART_HIP(hipGetLastError());
    }
    done += n;
  }
  return 0;
}

int sart_finalize_mass_scan_device(sart_context* c, const sart_trace_params_t* p, const double* masses, int32_t n_masses,
                                   const void* raw_dev, double* out_dev) {
  if (int rc = scan_check(c, p, masses, n_masses)) return rc;
  if (!raw_dev || !out_dev) return fail(SART_ERR_INVALID_ARGUMENT, "NULL argument");
  SART_HIP(hipSetDevice(c->device));
  if (int rc = refresh_derived(c)) return rc;
  if (int rc = sync_blob(c)) return rc;
  if (int rc = status_ensure(c)) return rc;
  for (int32_t k0 = 0; k0 < n_masses; k0 += kScanMaxMasses) {
    const int n = std::min<int32_t>(kScanMaxMasses, n_masses - k0);
    double q_w[kScanMaxMasses], q_w2[kScanMaxMasses];
    for (int k = 0; k < n; ++k) {
      QuantaExp qe;
      if (int rc = quanta_exp_of(c, weight_bound_of(c, p->flags, dm2_abs_of(c, masses[k0 + k])), qe)) return rc;
      q_w[k] = std::ldexp(1.0, qe.w);
      q_w2[k] = std::ldexp(1.0, qe.w2);
    }
    const size_t off = static_cast<size_t>(k0) * SART_SCAN_ROW;
    launch_finalize_scan(static_cast<const long long*>(raw_dev) + off, out_dev + off, n, q_w, q_w2,
                         (k0 + n == n_masses) ? n : -1, c->d_status.p, c->stream, 1u << SART_SCAN_N_PASSED);   // the counter row sits behind the last group
    SART_HIP(hipGetLastError());
  }
  return status_enqueue_copy(c);
}

int sart_trace_mass_scan(sart_context* c, const sart_trace_params_t* p, const double* masses, int32_t n_masses, double* out_host) {
  if (int rc = scan_check(c, p, masses, n_masses)) return rc;
  if (!out_host) return fail(SART_ERR_INVALID_ARGUMENT, "output is NULL");
  SART_HIP(hipSetDevice(c->device));
  const size_t len = sart_mass_scan_len(n_masses);
  if (int rc = c->d_scan.resize(len)) return rc;
  sart_trace_params_t q = *p;
  q.accumulate = 0;   // the blocking form has no accumulator the caller could add into
  if (int rc = sart_trace_mass_scan_device(c, &q, masses, n_masses, c->d_scan.p)) return rc;
  const double* src = c->d_scan.p;
  if (c->accum_mode == SART_ACCUM_FIXED64) {
    if (int rc = c->d_scan_fin.resize(len)) return rc;
    if (int rc = sart_finalize_mass_scan_device(c, p, masses, n_masses, c->d_scan.p, c->d_scan_fin.p)) return rc;
    src = c->d_scan_fin.p;
  }
  SART_HIP(hipMemcpyAsync(out_host, src, len * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  SART_HIP(hipStreamSynchronize(c->stream));
  return status_take(c);
}

// ---- fused angular scan (include/sart.h) ---------------------------------------------------------------------------------------
namespace {

int ascan_check(sart_context* c, const sart_trace_params_t* p, int32_t n) {
  if (!c || !p) return fail(SART_ERR_INVALID_ARGUMENT, "NULL argument");
  if (n < 1 || n > 65536) return fail(SART_ERR_INVALID_ARGUMENT, "n_angles must be in [1, 65536]");
  return 0;
}

// The rotation and the bounds that follow it for telescope_turned_y = angle_y_deg, with the expressions of hoist_setup() /
// sync_blob() (a single-angle launch after sart_set_telescope_angles computes the same bits).
AScanAngle ascan_angle_of(const sart_context* c, double angle_y_deg) {
  const sart_setup_t& s = c->setup;
  AScanAngle a;
  std::memset(&a, 0, sizeof a);
  const double turnedX = deg2rad(s.telescope_turned_x_deg), turnedY = deg2rad(angle_y_deg);
  a.rx_c = std::cos(turnedX); a.rx_s = std::sin(turnedX);
  a.ry_c = std::cos(turnedY); a.ry_s = std::sin(turnedY);
  a.half_length_telescope = c->params.half_length_telescope;
  DevParams P = c->params;   // what shell0_miss_radius_of reads of the rotation
  P.rotated = (s.telescope_turned_x_deg != 0.0 || angle_y_deg != 0.0) ? 1 : 0;
  P.rx_c = a.rx_c; P.rx_s = a.rx_s; P.ry_c = a.ry_c; P.ry_s = a.ry_s;
  a.shell0_miss_radius = (c->knobs.no_sure_miss || c->knobs.no_early_reject) ? -1.0 : shell0_miss_radius_of(s, P, c->n_radii);
  a.mx = a.rx_s;
  a.my = -a.rx_c * a.ry_s;
  a.mz = a.rx_c * a.ry_c;
  return a;
}

}  // namespace

int sart_trace_angular_scan_device(sart_context* c, const sart_trace_params_t* p, const double* turned_y_deg, int32_t n_angles,
                                   double* scan_dev) {
  if (int rc = ascan_check(c, p, n_angles)) return rc;
  if (!turned_y_deg || !scan_dev) return fail(SART_ERR_INVALID_ARGUMENT, "NULL argument");
  for (int32_t k = 0; k < n_angles; ++k)
    if (!std::isfinite(turned_y_deg[k]) || !(std::fabs(turned_y_deg[k]) < 90.0))
      return fail(SART_ERR_INVALID_ARGUMENT, "telescope angles must be finite and inside (-90, 90) degrees");
  SART_HIP(hipSetDevice(c->device));
  if (int rc = refresh_derived(c)) return rc;
  if (int rc = sync_blob(c)) return rc;
  sart_trace_params_t q = *p;   // a scan accumulates no image: whatever the caller left in the image fields is not read
  q.image_nx = q.image_ny = 1;
  q.image_x_min = q.image_y_min = 0.0;
  q.image_x_max = q.image_y_max = 1.0;
  q.spectra = 0;
  TraceArgs a;
  if (int rc = make_args(c, &q, a)) return rc;
  a.fx_scale_w = a.fx_scale_w2 = 0.0;
  const bool fixed = c->accum_mode == SART_ACCUM_FIXED64;
  if (fixed) {   // a pure function of setup, tables, flags and headroom (the bound does not depend on the angle)
    QuantaExp qe;
    if (int rc = quanta_exp_of(c, weight_bound_of(c, p->flags, c->params.gas_dm2_abs), qe)) return rc;
    a.fx_scale_w = std::ldexp(1.0, -qe.w);
    a.fx_scale_w2 = std::ldexp(1.0, -qe.w2);
  }
  if (!p->accumulate) SART_HIP(hipMemsetAsync(scan_dev, 0, sart_angular_scan_len(n_angles) * sizeof(double), c->stream));
  if (p->n_rays == 0) return 0;
  const DevParams& P = c->params;
  const bool fast = !P.test_active && !(P.telescope_kind == SART_TK_XMM && P.inner_blocks < 0) && !c->knobs.force_generic && !P.stage_gas;
  // Stage A0 for a telescope that is turned through several angles: the cuts behind the field do not depend on the angle; the
  // zones in terms of the radial distance at the entrance (inner disc, ring, beyond the last shell) are built per launch group
  // with the margin of its largest tilt (build_zones) - a ray inside one of them is dead at EVERY angle of the group and is never
  // sampled, stashed or turned.
  HotA hot = c->hot;
  hot.rotated = 1;
  int& bpc = c->blocks_per_cu_ascan[fast ? 1 : 0];
  if (bpc == 0) {
    bpc = std::max(1, angular_scan_blocks_per_cu(fast));
    if (c->knobs.hist_blocks_per_cu > 0) bpc = c->knobs.hist_blocks_per_cu;
  }
  const int32_t n_groups = (n_angles + kAScanMaxAngles - 1) / kAScanMaxAngles;
  for (uint64_t done = 0; done < p->n_rays;) {   // ray indices inside one launch are 32-bit: pieces of at most 2^31 rays
    const uint64_t n = std::min<uint64_t>(p->n_rays - done, 1ull << 31);
    a.n_rays = n;
    a.ray_id_offset = p->ray_id_offset + done;
    const int n_blocks = grid_for(n, c->n_cu, bpc, 1024);
    const size_t need = static_cast<size_t>(n_blocks) * kAScanMaxAngles * kAScanPartialSlots;
    if (c->d_ascan_partials.n < need) {
      SART_HIP(hipStreamSynchronize(c->stream));
      const size_t rows = std::max<size_t>(static_cast<size_t>(n_blocks), static_cast<size_t>(c->n_cu) * 4);
      if (int rc = c->d_ascan_partials.resize(rows * kAScanMaxAngles * kAScanPartialSlots)) return rc;
    }
    a.partials = nullptr;
    for (int32_t g = 0, k0 = 0; g < n_groups; ++g) {   // balanced groups: sizes differ by at most one
      AScanArgs an;
      std::memset(&an, 0, sizeof an);
      an.n_angles = n_angles / n_groups + (g < n_angles % n_groups ? 1 : 0);
      an.partials = c->d_ascan_partials.p;
      double tilt_max = 0.0;
      for (int k = 0; k < an.n_angles; ++k) {
        an.a[k] = ascan_angle_of(c, turned_y_deg[k0 + k]);
        const double t = tilt_bound_of(c->setup, turned_y_deg[k0 + k]);
        tilt_max = (t < 0.0 || tilt_max < 0.0) ? -1.0 : std::max(tilt_max, t);
      }
      if (!c->knobs.no_early_reject) build_zones(c->setup, P, c->n_radii, hot, tilt_max);
      double* const rows = scan_dev + static_cast<size_t>(k0) * SART_ASCAN_ROW;
      double* const shared = (k0 == 0) ? scan_dev + static_cast<size_t>(n_angles) * SART_ASCAN_ROW : nullptr;   // counters: once per piece
      {
        TimedLaunch tl(c);
        launch_trace_angular_scan(hot, c->hotb, c->d_blob.p, a, an, rows, shared, n_blocks, c->stream, fast, fixed);
      }
      SART_HIP(hipGetLastError());
      k0 += an.n_angles;
    }
    done += n;
  }
  return 0;
}

int sart_finalize_angular_scan_device(sart_context* c, const sart_trace_params_t* p, int32_t n_angles, const void* raw_dev, double* out_dev) {
  if (int rc = ascan_check(c, p, n_angles)) return rc;
  if (!raw_dev || !out_dev) return fail(SART_ERR_INVALID_ARGUMENT, "NULL argument");
  SART_HIP(hipSetDevice(c->device));
  if (int rc = refresh_derived(c)) return rc;
  if (int rc = sync_blob(c)) return rc;
  if (int rc = status_ensure(c)) return rc;
  QuantaExp qe;
  if (int rc = quanta_exp_of(c, weight_bound_of(c, p->flags, c->params.gas_dm2_abs), qe)) return rc;
  double q_w[kScanMaxMasses], q_w2[kScanMaxMasses];
  for (int k = 0; k < kScanMaxMasses; ++k) { q_w[k] = std::ldexp(1.0, qe.w); q_w2[k] = std::ldexp(1.0, qe.w2); }
  // bit 31: an angle whose few, faint rays the common quantum does not resolve reads NaN in its own row (finalize_scan_kernel)
  constexpr uint32_t kCounters = (1u << SART_ASCAN_N_PASSED) | (1u << SART_ASCAN_N_SHELL_SELECTED) | (1u << SART_ASCAN_N_HIT_NICKEL) |
                                 (1u << SART_ASCAN_N_PASSED_TILL_WINDOW) | (1u << 31);
  static_assert(SART_ASCAN_ROW == SART_SCAN_ROW && SART_ASCAN_SUM_WEIGHTS == SART_SCAN_SUM_WEIGHTS && SART_ASCAN_SUM_WEIGHTS_SQ == SART_SCAN_SUM_WEIGHTS_SQ &&
                    SART_ASCAN_SUM_WEIGHTS_HI == SART_SCAN_SUM_WEIGHTS_HI && SART_ASCAN_SUM_WEIGHTS_SQ_HI == SART_SCAN_SUM_WEIGHTS_SQ_HI &&
                    SART_ASCAN_N_PASSED == SART_SCAN_N_PASSED, "the two scans share the row layout of their sums and the finalize kernel");
  for (int32_t k0 = 0; k0 < n_angles; k0 += kScanMaxMasses) {
    const int n = std::min<int32_t>(kScanMaxMasses, n_angles - k0);
    const size_t off = static_cast<size_t>(k0) * SART_ASCAN_ROW;
    launch_finalize_scan(static_cast<const long long*>(raw_dev) + off, out_dev + off, n, q_w, q_w2, (k0 + n == n_angles) ? n : -1,
                         c->d_status.p, c->stream, kCounters);
    SART_HIP(hipGetLastError());
  }
  return status_enqueue_copy(c);
}

int sart_trace_angular_scan(sart_context* c, const sart_trace_params_t* p, const double* turned_y_deg, int32_t n_angles, double* out_host) {
  if (int rc = ascan_check(c, p, n_angles)) return rc;
  if (!turned_y_deg || !out_host) return fail(SART_ERR_INVALID_ARGUMENT, "NULL argument");
  SART_HIP(hipSetDevice(c->device));
  const size_t len = sart_angular_scan_len(n_angles);
  if (int rc = c->d_scan.resize(len)) return rc;
  sart_trace_params_t q = *p;
  q.accumulate = 0;   // the blocking form has no accumulator the caller could add into
  if (int rc = sart_trace_angular_scan_device(c, &q, turned_y_deg, n_angles, c->d_scan.p)) return rc;
  const double* src = c->d_scan.p;
  if (c->accum_mode == SART_ACCUM_FIXED64) {
    if (int rc = c->d_scan_fin.resize(len)) return rc;
    if (int rc = sart_finalize_angular_scan_device(c, p, n_angles, c->d_scan.p, c->d_scan_fin.p)) return rc;
    src = c->d_scan_fin.p;
  }
  SART_HIP(hipMemcpyAsync(out_host, src, len * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  SART_HIP(hipStreamSynchronize(c->stream));
  return status_take(c);
}

// RCCL through dlopen (the library is only needed by hosts that drive several GPUs from one process).  Types, enum values and the
// entry points' signatures are the installed header's (decltype of its declarations: nothing typed by hand); the symbols are
// looked up at run time, so libsart.so carries no link-time dependency on librccl.
namespace {
struct Rccl {
  void* lib = nullptr;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclReduce) Reduce = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  bool ok = false;
  static Rccl& get() {
    static Rccl r = load();   // (thread-safe: a function-local static is initialised once)
    return r;
  }
  static Rccl load() {
    Rccl r;
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so", nullptr};
    for (int i = 0; names[i] && !r.lib; ++i) r.lib = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
    if (!r.lib) return r;
    r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(dlsym(r.lib, "ncclCommInitAll"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.lib, "ncclCommDestroy"));
    r.Reduce = reinterpret_cast<decltype(r.Reduce)>(dlsym(r.lib, "ncclReduce"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(dlsym(r.lib, "ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(dlsym(r.lib, "ncclGroupEnd"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.lib, "ncclGetErrorString"));
    r.ok = r.CommInitAll && r.CommDestroy && r.Reduce && r.GroupStart && r.GroupEnd && r.GetErrorString;
    return r;
  }
};

// One communicator set per ordered device list, created on first use and kept for the life of the process: a scan
// reduces once per point, and ncclCommInitAll costs orders of magnitude more than the 512 KB reduce itself.
std::mutex g_comm_mutex;
std::map<std::vector<int>, std::vector<ncclComm_t>> g_comm_cache;

enum : uint32_t {
  kReduceForceRccl = 1u,       // take the RCCL route for n == 1 too (the public entry returns before it: nothing to add up)
  kReduceInjectFailure = 2u,   // hand ncclReduce a data type that does not exist: the real failure branch, communicators dropped
};

// The body of sart_reduce_across_devices.  `recv` (test entry only): out-of-place receive buffers, NULL = in place as the public
// entry does it.  info[0] = 1 if the communicators came from the cache, info[1] = communicator sets alive after the call.
int reduce_across_devices_impl(sart_context* const* ctxs, double* const* accs, double* const* recv, int32_t n, size_t n_doubles,
                               int32_t root, uint32_t flags, int32_t* info) {
  if (!ctxs || !accs || n < 1 || root < 0 || root >= n) return fail(SART_ERR_INVALID_ARGUMENT, "sart_reduce_across_devices: bad argument");
  for (int i = 0; i < n; ++i) {
    if (!ctxs[i] || !accs[i] || (recv && !recv[i])) return fail(SART_ERR_INVALID_ARGUMENT, "NULL context / accumulator");
    if (ctxs[i]->accum_mode != ctxs[0]->accum_mode) return fail(SART_ERR_INVALID_ARGUMENT, "contexts differ in their accumulation mode");
    for (int j = 0; j < i; ++j)
      if (ctxs[j]->device == ctxs[i]->device) return fail(SART_ERR_INVALID_ARGUMENT, "contexts must be on distinct devices");
  }
  ncclDataType_t dtype = ctxs[0]->accum_mode == SART_ACCUM_FIXED64 ? ncclInt64 : ncclFloat64;
  if (flags & kReduceInjectFailure) dtype = static_cast<ncclDataType_t>(ncclNumTypes + 17);
  for (int i = 0; i < n; ++i) {   // everything queued so far must be visible to the collective
    SART_HIP(hipSetDevice(ctxs[i]->device));
    SART_HIP(hipStreamSynchronize(ctxs[i]->stream));
  }
  if (n == 1 && !(flags & kReduceForceRccl)) return 0;
  Rccl& r = Rccl::get();
  if (!r.ok) return fail(SART_ERR_UNSUPPORTED, "librccl could not be loaded");
  std::vector<int> devs(n);
  for (int i = 0; i < n; ++i) devs[i] = ctxs[i]->device;
  std::lock_guard<std::mutex> lock(g_comm_mutex);
  auto it = g_comm_cache.find(devs);
  if (info) info[0] = it != g_comm_cache.end();
  if (it == g_comm_cache.end()) {
    std::vector<ncclComm_t> fresh(n, nullptr);
    const ncclResult_t rc_init = r.CommInitAll(fresh.data(), n, devs.data());
    if (rc_init != ncclSuccess) return fail(SART_ERR_INTERNAL, std::string("ncclCommInitAll: ") + r.GetErrorString(rc_init));
    it = g_comm_cache.emplace(devs, std::move(fresh)).first;
  }
  const std::vector<ncclComm_t>& comms = it->second;
  ncclResult_t rc = r.GroupStart();
  bool set_device_failed = false;
  for (int i = 0; i < n && rc == ncclSuccess; ++i) {
    if (hipSetDevice(ctxs[i]->device) != hipSuccess) { set_device_failed = true; break; }
    rc = r.Reduce(accs[i], recv ? recv[i] : accs[i], n_doubles, dtype, ncclSum, root, comms[i], ctxs[i]->stream);
  }
  const ncclResult_t rc_end = r.GroupEnd();
  if (rc == ncclSuccess) rc = rc_end;
  for (int i = 0; i < n; ++i) {
    (void)hipSetDevice(ctxs[i]->device);
    (void)hipStreamSynchronize(ctxs[i]->stream);
  }
  const bool failed = rc != ncclSuccess || set_device_failed;
  if (failed) {   // a failed collective leaves the communicators in an unknown state: drop them
    for (ncclComm_t cm : comms) r.CommDestroy(cm);
    g_comm_cache.erase(it);
  }
  if (info) info[1] = static_cast<int32_t>(g_comm_cache.size());
  if (failed) return fail(SART_ERR_INTERNAL, std::string("ncclReduce: ") + (set_device_failed ? "hipSetDevice failed" : r.GetErrorString(rc)));
  return 0;
}
}  // namespace

int sart_reduce_across_devices(sart_context* const* ctxs, double* const* accs, int32_t n, size_t n_doubles, int32_t root) {
  return reduce_across_devices_impl(ctxs, accs, nullptr, n, n_doubles, root, 0u, nullptr);
}

// Test entry (not part of include/sart.h): sart_reduce_across_devices with the switches above, so that a one-GPU box runs every
// line of the RCCL leg - dlopen + symbol lookup, ncclCommInitAll, the grouped ncclReduce in both element types, the communicator
// cache and the failure branch (tests/test_gpu_rccl_leg.py).
__attribute__((visibility("default"))) int sart_internal_reduce_across_devices(sart_context* const* ctxs, double* const* accs,
                                                                               double* const* recv, int32_t n, size_t n_doubles,
                                                                               int32_t root, uint32_t flags, int32_t* info) {
  return reduce_across_devices_impl(ctxs, accs, recv, n, n_doubles, root, flags, info);
}

int sart_enable_kernel_timing(sart_context* c, int enable) {
  if (!c) return fail(SART_ERR_INVALID_ARGUMENT, "ctx is NULL");
  c->timing = enable != 0;
  c->events_used = 0;
  return 0;
}

int sart_get_kernel_timing(sart_context* c, double* total_ms, int64_t* n_launches) {
  if (!c) return fail(SART_ERR_INVALID_ARGUMENT, "ctx is NULL");
  SART_HIP(hipSetDevice(c->device));
  SART_HIP(hipStreamSynchronize(c->stream));
  double tot = 0.0;
  for (size_t i = 0; i < c->events_used; ++i) {
    float ms = 0.f;
    SART_HIP(hipEventElapsedTime(&ms, c->events[i].first, c->events[i].second));
    tot += ms;
  }
  if (total_ms) *total_ms = tot;
  if (n_launches) *n_launches = static_cast<int64_t>(c->events_used);
  c->events_used = 0;
  return 0;
}

int sart_device_info(sart_context* c, int32_t* n_cu, int32_t* wave_size, char* name_buf, size_t name_buf_len) {
  if (!c) return fail(SART_ERR_INVALID_ARGUMENT, "ctx is NULL");
  if (n_cu) *n_cu = c->n_cu;
  if (wave_size) *wave_size = 64;
  if (name_buf && name_buf_len) {
    std::snprintf(name_buf, name_buf_len, "%s", c->device_name.c_str());
  }
  return 0;
}

}  // extern "C"
