// raytracer_host.cpp — host-side mirror (C++) of the reference's setup / driver layer that sits
// directly above the per-ray hot path.  Pure host code: geometry constants, CDF construction,
// window geometry and the two drivers that call the sart.h C-ABI.
//
// Reference (jovoy/SolarAxionRayTracing, src/raytracer.nim) lines are cited per function.
// The numeric tables are the experiment's hardware description (CAST / BabyIAXO magnets,
// LLNL / XMM / Abrixas shell radii and angles) and must be reproduced digit for digit.
#include "../../include/sart_host.h"

#include <cmath>
#include <cstring>
#include <string>
#include <vector>
#include <algorithm>
#include <cstdio>
#include <dlfcn.h>

namespace {
thread_local std::string g_err;
}
namespace sart_host {
int fail(int code, const std::string& msg) {   // also used by opcd_host.cpp
  g_err = msg;
  return code;
}
}  // namespace sart_host
using sart_host::fail;

namespace {

constexpr double kPi = 3.14159265358979323846;

// ---------------------------------------------------------------- shell tables ------------
// LLNL (CAST, nustar-like), raytracer.nim:1265-1274
const double kLlnlR1[14] = {63.006, 65.606, 68.305, 71.105, 74.011, 77.027, 80.157,
                            83.405, 86.775, 90.272, 93.902, 97.668, 101.576, 105.632};
const double kLlnlXsep[14] = {4.171, 4.140, 4.221, 4.190, 4.228, 4.245, 4.288,
                              4.284, 4.306, 4.324, 4.373, 4.387, 4.403, 4.481};
const double kLlnlAngles[14] = {0.579, 0.603, 0.628, 0.654, 0.680, 0.708, 0.737,
                                0.767, 0.798, 0.830, 0.863, 0.898, 0.933, 0.970};

// XMM-Newton, raytracer.nim:1289-1313
const double kXmmThickness[58] = {
    0.468, 0.475, 0.482, 0.490, 0.497, 0.504, 0.511, 0.519, 0.526, 0.534, 0.542, 0.549,
    0.557, 0.566, 0.574, 0.583, 0.591, 0.600, 0.609, 0.618, 0.627, 0.636, 0.646, 0.655,
    0.665, 0.675, 0.684, 0.694, 0.704, 0.714, 0.724, 0.735, 0.745, 0.756, 0.768, 0.779,
    0.790, 0.802, 0.814, 0.826, 0.838, 0.850, 0.862, 0.874, 0.887, 0.900, 0.913, 0.927,
    0.941, 0.955, 0.968, 0.983, 0.997, 1.011, 1.026, 1.041, 1.055, 1.070};
const double kXmmR1[58] = {
    153.118,  155.4105, 157.7235, 160.0565, 162.42,   164.803,  167.217,  169.651,  172.115,
    174.5995, 177.1145, 179.6495, 182.2145, 184.9615, 187.739,  190.5465, 193.3845, 196.253,
    199.1515, 202.0805, 205.0395, 208.0795, 211.1495, 214.25,   217.381,  220.542,  223.7435,
    226.9755, 230.2375, 233.54,   236.873,  240.236,  243.6395, 247.2855, 250.9715, 254.6985,
    258.4655, 262.2625, 266.1005, 269.9785, 273.897,  277.856,  281.8555, 285.9055, 289.9955,
    294.178,  298.661,  303.0945, 307.5685, 312.093,  316.658,  321.2735, 325.939,  330.6555,
    335.4225, 340.23,   345.0875, 349.996};
const double kXmmAngles[58] = {
    0.29,  0.294, 0.298, 0.303, 0.307, 0.312, 0.316, 0.321, 0.325, 0.33,  0.335, 0.34,
    0.345, 0.35,  0.355, 0.36,  0.366, 0.371, 0.377, 0.382, 0.388, 0.393, 0.399, 0.405,
    0.411, 0.417, 0.423, 0.429, 0.435, 0.441, 0.448, 0.454, 0.461, 0.467, 0.474, 0.481,
    0.489, 0.496, 0.503, 0.51,  0.518, 0.525, 0.533, 0.54,  0.548, 0.556, 0.564, 0.573,
    0.581, 0.59,  0.598, 0.607, 0.616, 0.625, 0.634, 0.643, 0.652, 0.661};

// Abrixas, raytracer.nim:1328-1340
const double kAbrixasThickness[27] = {0.2,  0.2,  0.2,  0.2,  0.2,  0.2,  0.2,  0.25, 0.25,
                                      0.25, 0.25, 0.25, 0.25, 0.25, 0.3,  0.3,  0.3,  0.3,
                                      0.3,  0.35, 0.35, 0.35, 0.35, 0.35, 0.4,  0.4,  0.4};
const double kAbrixasR1[27] = {38.125, 39.353, 40.581, 41.809, 43.036, 44.292, 45.577, 46.894, 48.295,
                               49.731, 51.201, 52.707, 54.249, 55.829, 57.447, 59.157, 60.909, 62.703,
                               64.540, 66.423, 68.403, 70.431, 72.509, 74.637, 76.817, 79.102, 81.443};
const double kAbrixasAngles[27] = {0.3335, 0.3443, 0.3550, 0.3657, 0.3765, 0.3874, 0.3987, 0.4102, 0.4225,
                                   0.4350, 0.4479, 0.4610, 0.4745, 0.4883, 0.5024, 0.5174, 0.5327, 0.5484,
                                   0.5644, 0.5809, 0.5982, 0.6159, 0.6340, 0.6526, 0.6716, 0.6916, 0.7120};

void set3(double* dst, double a, double b, double c) { dst[0] = a; dst[1] = b; dst[2] = c; }

// initMagnet, raytracer.nim:1098-1123
void initMagnet(int32_t setup, const sart_magnet_config_t* cfg, sart_setup_t& s) {
  if (cfg) {  // maybeParseMagnetConfig :1032-1051
    s.magnet_B = cfg->B; s.magnet_lengthB = cfg->lengthB; s.magnet_radiusCB = cfg->radiusCB;
    s.magnet_lengthColdbore = cfg->lengthColdbore; s.magnet_pGasRoom = cfg->pGasRoom; s.magnet_tGas = cfg->tGas;
    return;
  }
  if (setup == SART_ES_CAST) {
    s.magnet_B = 9.0; s.magnet_radiusCB = 21.5; s.magnet_lengthColdbore = 9756.0;
    s.magnet_lengthB = 9260.0; s.magnet_pGasRoom = 1.0; s.magnet_tGas = 1.7;
  } else {
    s.magnet_B = 2.0; s.magnet_radiusCB = 500.0; s.magnet_lengthColdbore = 11300.0;
    s.magnet_lengthB = 11000.0; s.magnet_pGasRoom = 1.0; s.magnet_tGas = 100.0;
  }
}

// initPipes, raytracer.nim:1125-1157
int initPipes(int32_t optics, sart_setup_t& s) {
  switch (optics) {
    case SART_TK_LLNL:
      s.pipe_cb_vt3_length = 127.66; s.pipe_cb_vt3_radius = 39.89;
      s.pipe_vt3_xrt_length = 111.7; s.pipe_vt3_xrt_radius = 23.935;
      s.pipes_turned_deg = 2.75; s.distance_cb_axis_xrt_axis = 0.0;
      return 0;
    case SART_TK_ABRIXAS:
      s.pipe_cb_vt3_length = 114.3; s.pipe_cb_vt3_radius = 66.65;
      s.pipe_vt3_xrt_length = 171.43; s.pipe_vt3_xrt_radius = 47.62;
      s.pipes_turned_deg = 0.0; s.distance_cb_axis_xrt_axis = 0.0;
      return 0;
    case SART_TK_CUSTOM_BABYIAXO:
    case SART_TK_XMM:
      s.pipe_cb_vt3_length = 225.0; s.pipe_cb_vt3_radius = 370.0;
      s.pipe_vt3_xrt_length = 250.0; s.pipe_vt3_xrt_radius = 370.0;
      s.pipes_turned_deg = 0.0; s.distance_cb_axis_xrt_axis = 0.0;
      return 0;
    default:
      return fail(SART_ERR_UNSUPPORTED, "Invalid telescope!");  // doAssert :1157
  }
}

// initTelescope + the `kind`/`layers` part of initReflectivity, raytracer.nim:1251-1348, :1160-1231
int initTelescope(int32_t optics, sart_setup_t& s) {
  s.telescope_kind = optics;
  s.telescope_turned_x_deg = 0.0;
  s.telescope_turned_y_deg = 0.0;
  std::memset(s.all_r1, 0, sizeof s.all_r1);
  std::memset(s.all_thickness, 0, sizeof s.all_thickness);
  std::memset(s.all_xsep, 0, sizeof s.all_xsep);
  std::memset(s.all_angles_deg, 0, sizeof s.all_angles_deg);
  std::memset(s.coating_layers, 0, sizeof s.coating_layers);
  switch (optics) {
    case SART_TK_LLNL:
      set3(s.optics_entrance, -83.0, 0.0, 0.0);
      set3(s.optics_exit, -83.0, 0.0, 454.0);
      s.n_shells = 14;
      for (int i = 0; i < 14; ++i) {
        s.all_thickness[i] = 0.2; s.all_r1[i] = kLlnlR1[i];
        s.all_xsep[i] = kLlnlXsep[i]; s.all_angles_deg[i] = kLlnlAngles[i];
      }
      s.l_mirror = 225.0; s.hole_in_optics = 0.0; s.number_of_holes = 5; s.hole_type = SART_HT_CROSS;
      s.reflectivity_kind = SART_RK_MULTI_COATING;  // :1165-1168
      s.n_coatings = 4;
      s.coating_layers[0] = 2; s.coating_layers[1] = 2 + 3;
      s.coating_layers[2] = 2 + 3 + 4; s.coating_layers[3] = 2 + 3 + 4 + 5;
      return 0;
    case SART_TK_XMM:
      set3(s.optics_entrance, 0.0, -0.0, 0.0);
      set3(s.optics_exit, 0.0, -0.0, 600.0);
      s.n_shells = 58;
      for (int i = 0; i < 58; ++i) {
        s.all_thickness[i] = kXmmThickness[i]; s.all_r1[i] = kXmmR1[i];
        s.all_xsep[i] = 0.0; s.all_angles_deg[i] = kXmmAngles[i];
      }
      s.l_mirror = 300.0; s.hole_in_optics = 0.2; s.number_of_holes = 1; s.hole_type = SART_HT_NONE;
      s.reflectivity_kind = SART_RK_SINGLE_COATING;  // :1189-1191
      s.n_coatings = 1; s.coating_layers[0] = 58;
      return 0;
    case SART_TK_ABRIXAS:
      set3(s.optics_entrance, 0.0, -60.0, 0.0);
      set3(s.optics_exit, 0.0, -60.0, 600.0);
      s.n_shells = 27;
      for (int i = 0; i < 27; ++i) {
        s.all_thickness[i] = kAbrixasThickness[i]; s.all_r1[i] = kAbrixasR1[i];
        s.all_xsep[i] = 0.0; s.all_angles_deg[i] = kAbrixasAngles[i];
      }
      s.l_mirror = 150.0; s.hole_in_optics = 0.2; s.number_of_holes = 1; s.hole_type = SART_HT_NONE;
      s.reflectivity_kind = SART_RK_SINGLE_COATING;  // :1211-1213
      s.n_coatings = 1; s.coating_layers[0] = 27;
      return 0;
    case SART_TK_CUSTOM_BABYIAXO:
      return fail(SART_ERR_UNSUPPORTED,
                  "Reflectivities are not yet implemented for the tkCustomBabyIAXO and tkAbrixas optics.");  // :1233
    default:
      return fail(SART_ERR_UNSUPPORTED, "The telescope for this kind has not been implemented yet!");  // :1348
  }
}

// initTestXraySource, raytracer.nim:1350-1379
void initTestXraySource(int32_t setup, uint32_t flags, const sart_test_source_config_t* cfg, sart_setup_t& s) {
  if (cfg) {  // maybeParseTestXraySource :1053-1076
    s.test_active = cfg->active; s.test_parallel = cfg->parallel; s.test_energy = cfg->energy;
    s.test_distance = cfg->distance; s.test_radius = cfg->radius; s.test_off_axis_up = cfg->offAxisUp;
    s.test_off_axis_left = cfg->offAxisLeft; s.test_activity = cfg->activity; s.test_length_col = cfg->lengthCol;
    return;
  }
  s.test_active = (flags & SART_CF_XRAY_TEST) ? 1 : 0;
  s.test_parallel = 1;
  if (setup == SART_ES_CAST) {
    s.test_distance = 100.0; s.test_radius = 10.0; s.test_off_axis_up = 200.0; s.test_off_axis_left = 0.0;
    s.test_length_col = 50.0; s.test_energy = 1.0; s.test_activity = 1.0;
  } else {
    s.test_distance = 2000.0; s.test_radius = 350.0; s.test_off_axis_up = 0.0; s.test_off_axis_left = 0.0;
    s.test_length_col = 0.0; s.test_energy = 0.021; s.test_activity = 0.125;
  }
}

// initDetectorInstallation, raytracer.nim:1381-1409
int initDetectorInstallation(int32_t optics, const sart_detector_install_config_t* cfg, sart_setup_t& s) {
  if (cfg) {  // maybeParseDetectorInstallation :1078-1096
    s.distance_detector_xrt = cfg->distanceDetectorXRT;
    s.distance_window_focal_plane = cfg->distanceWindowFocalPlane;
    s.lateral_shift = cfg->lateralShift; s.transversal_shift = cfg->transversalShift;
    return 0;
  }
  s.distance_window_focal_plane = 0.0; s.lateral_shift = 0.0;
  switch (optics) {
    case SART_TK_LLNL: s.distance_detector_xrt = 1485.0; s.transversal_shift = 0.0; return 0;
    case SART_TK_ABRIXAS: s.distance_detector_xrt = 1600.0; s.transversal_shift = 0.0; return 0;
    case SART_TK_XMM:
    case SART_TK_CUSTOM_BABYIAXO:
      s.distance_detector_xrt = 7500.0;
      s.transversal_shift = std::sin(0.0 * (kPi / 180.0)) * 7500.0;  // :1406
      return 0;
    default: return fail(SART_ERR_UNSUPPORTED, "Invalid telescope!");
  }
}

// calcWindowVals, raytracer.nim:1431-1462
void calcWindowVals(double radiusWindow, int numberOfStrips, double openApertureRatio, double& width,
                    double& dist) {
  const double totalArea = kPi * radiusWindow * radiusWindow;
  const double areaOfStrips = totalArea * (1.0 - openApertureRatio);
  const double dAndwPerStrip = radiusWindow * 2.0 / (static_cast<double>(numberOfStrips) + 1.0);
  double lengthAllStrips = 0.0;
  const int nHalf = static_cast<int>(std::round(numberOfStrips / 2.0));
  for (int i = 0; i <= nHalf - 1; ++i) {
    const double off = i * dAndwPerStrip + 0.5 * dAndwPerStrip;
    const double lengthStrip = std::sqrt(radiusWindow * radiusWindow - off * off) * 2.0;
    lengthAllStrips = lengthAllStrips + lengthStrip;
  }
  lengthAllStrips = lengthAllStrips * 2.0;
  width = areaOfStrips / lengthAllStrips;
  dist = dAndwPerStrip - width;
}

// newDetectorSetup without the TSV reads, raytracer.nim:1464-1496, :1528, toRad :322-332
int newDetectorSetup(int32_t kind, sart_setup_t& s) {
  if (kind != SART_DK_INGRID2017 && kind != SART_DK_INGRID2018 && kind != SART_DK_INGRIDIAXO)
    return fail(SART_ERR_INVALID_ARGUMENT, "invalid enum value for DetectorSetupKind");
  s.detector_kind = kind;
  s.radius_window = 7.0;
  s.number_of_strips = 4;
  s.open_aperture_ratio = 0.838;
  s.depth_det = 30.0;
  calcWindowVals(s.radius_window, s.number_of_strips, s.open_aperture_ratio, s.strip_width_window,
                 s.strip_dist_window);
  const double deg = (kind == SART_DK_INGRIDIAXO) ? 20.0 : 30.0;
  s.theta_rad = deg * (kPi / 180.0);
  return 0;
}

// ---- libhdf5 through dlopen (HDF5 1.10 C API; hid_t = int64_t) ------------------------------------------------
struct H5 {
  using hid = int64_t;
  void* lib = nullptr;
  int (*H5open)() = nullptr;
  hid (*H5Fopen)(const char*, unsigned, hid) = nullptr;
  hid (*H5Fcreate)(const char*, unsigned, hid, hid) = nullptr;
  int (*H5Fclose)(hid) = nullptr;
  hid (*H5Dopen2)(hid, const char*, hid) = nullptr;
  hid (*H5Dcreate2)(hid, const char*, hid, hid, hid, hid, hid) = nullptr;
  int (*H5Dclose)(hid) = nullptr;
  hid (*H5Dget_space)(hid) = nullptr;
  long long (*H5Sget_simple_extent_npoints)(hid) = nullptr;
  hid (*H5Screate_simple)(int, const unsigned long long*, const unsigned long long*) = nullptr;
  int (*H5Sclose)(hid) = nullptr;
  int (*H5Dread)(hid, hid, hid, hid, hid, void*) = nullptr;
  int (*H5Dwrite)(hid, hid, hid, hid, hid, const void*) = nullptr;
  int (*H5Lexists)(hid, const char*, hid) = nullptr;
  int (*H5Eset_auto2)(hid, void*, void*) = nullptr;
  hid native_double = -1;
  bool ok = false;

  static H5& get() {
    static H5 h;
    static bool tried = false;
    if (tried) return h;
    tried = true;
    const char* names[] = {"libhdf5.so", "libhdf5.so.103", "libhdf5_serial.so", "/opt/conda/lib/libhdf5.so", nullptr};
    for (int i = 0; names[i] && !h.lib; ++i) h.lib = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
    if (!h.lib) return h;
#define SART_H5SYM(n) h.n = reinterpret_cast<decltype(h.n)>(dlsym(h.lib, #n)); if (!h.n) return h;
    SART_H5SYM(H5open) SART_H5SYM(H5Fopen) SART_H5SYM(H5Fcreate) SART_H5SYM(H5Fclose) SART_H5SYM(H5Dopen2)
    SART_H5SYM(H5Dcreate2) SART_H5SYM(H5Dclose) SART_H5SYM(H5Dget_space) SART_H5SYM(H5Sget_simple_extent_npoints)
    SART_H5SYM(H5Screate_simple) SART_H5SYM(H5Sclose) SART_H5SYM(H5Dread) SART_H5SYM(H5Dwrite) SART_H5SYM(H5Lexists)
    SART_H5SYM(H5Eset_auto2)
#undef SART_H5SYM
    if (h.H5open() < 0) return h;
    hid* nd = reinterpret_cast<hid*>(dlsym(h.lib, "H5T_NATIVE_DOUBLE_g"));
    if (!nd) return h;
    h.native_double = *nd;
    h.H5Eset_auto2(0, nullptr, nullptr);   // errors are reported through return codes
    h.ok = true;
    return h;
  }

  // reads a whole f64 dataset; returns its number of elements or -1
  long long read_all(hid file, const char* name, std::vector<double>& out) {
    if (H5Lexists(file, name, 0) <= 0) return -1;
    const hid d = H5Dopen2(file, name, 0);
    if (d < 0) return -1;
    const hid sp = H5Dget_space(d);
    const long long n = H5Sget_simple_extent_npoints(sp);
    H5Sclose(sp);
    if (n <= 0) { H5Dclose(d); return -1; }
    out.resize(static_cast<size_t>(n));
    const int rc = H5Dread(d, native_double, 0, 0, 0, out.data());
    H5Dclose(d);
    return rc < 0 ? -1 : n;
  }
};

int h5_count_coatings(H5& h, H5::hid f, bool& single) {
  single = h.H5Lexists(f, "Reflectivity", 0) > 0;
  if (single) return 1;
  int n = 0;
  while (n < SART_MAX_COATINGS && h.H5Lexists(f, ("Reflectivity" + std::to_string(n)).c_str(), 0) > 0) ++n;
  return n;
}

}  // namespace

extern "C" {

const char* sart_host_last_error(void) { return g_err.c_str(); }

int sart_host_new_full_setup(int32_t experiment, int32_t detector, int32_t stage, int32_t telescope, uint32_t flags,
                             const sart_magnet_config_t* magnet_cfg, const sart_test_source_config_t* source_cfg,
                             const sart_detector_install_config_t* install_cfg, sart_setup_t* out) {
  if (!out) return fail(SART_ERR_INVALID_ARGUMENT, "out is NULL");
  if (experiment != SART_ES_CAST && experiment != SART_ES_BABYIAXO)
    return fail(SART_ERR_INVALID_ARGUMENT, "invalid enum value for ExperimentSetupKind");  // parseEnum ValueError :1027
  if (stage != SART_SK_VACUUM && stage != SART_SK_GAS)
    return fail(SART_ERR_INVALID_ARGUMENT, "invalid enum value for StageKind");
  sart_setup_t s;
  std::memset(&s, 0, sizeof s);
  s.experiment = experiment;
  s.stage = stage;
  initMagnet(experiment, magnet_cfg, s);
  if (int rc = initTelescope(telescope, s)) return rc;
  initTestXraySource(experiment, flags, source_cfg, s);
  if (int rc = initPipes(telescope, s)) return rc;
  if (int rc = initDetectorInstallation(telescope, install_cfg, s)) return rc;
  if (int rc = newDetectorSetup(detector, s)) return rc;
  // module constants, raytracer.nim:248-272
  s.distance_sun_earth = 1.5e14;
  s.radius_sun = 6.9e11;
  s.room_temp = 293.15;
  s.m_axion = 0.0853;
  s.g_agamma = 1e-12;
  s.chip_x_max = 14.0;
  s.chip_y_max = 14.0;
  *out = s;
  return 0;
}

int sart_host_calc_window_vals(double radius_window, int32_t number_of_strips, double open_aperture_ratio,
                               double* width, double* dist) {
  if (!width || !dist) return fail(SART_ERR_INVALID_ARGUMENT, "NULL output");
  calcWindowVals(radius_window, number_of_strips, open_aperture_ratio, *width, *dist);
  return 0;
}

int sart_host_build_cdfs(const double* em_rates, const double* radii, const double* energies_kev, int32_t n_radii,
                         int32_t n_energies, double* flux_radius_cdf_out, double* diff_flux_cdfs_out) {
  if (!em_rates || !radii || !energies_kev || !flux_radius_cdf_out || !diff_flux_cdfs_out || n_radii < 1 ||
      n_energies < 1)
    return fail(SART_ERR_INVALID_ARGUMENT, "sart_host_build_cdfs: bad argument");
  // raytracer.nim:2670-2705
  double diffRadiusSum = 0.0;
  for (int32_t iRad = 0; iRad < n_radii; ++iRad) {
    const double radius = radii[iRad];
    const double* emRate = em_rates + static_cast<size_t>(iRad) * n_energies;
    double* radiusCumSum = diff_flux_cdfs_out + static_cast<size_t>(iRad) * n_energies;
    double diffSum = 0.0;
    for (int32_t iE = 0; iE < n_energies; ++iE) {
      const double energy = energies_kev[iE];
      const double diffFlux = emRate[iE] * (energy * energy) * radius * radius;
      diffSum += diffFlux;
      radiusCumSum[iE] = diffSum;
    }
    diffRadiusSum += diffSum;
    flux_radius_cdf_out[iRad] = diffRadiusSum;
    const double integral = radiusCumSum[n_energies - 1];  // toCdf :2675-2677
    for (int32_t iE = 0; iE < n_energies; ++iE) radiusCumSum[iE] = radiusCumSum[iE] / integral;
  }
  const double integral = flux_radius_cdf_out[n_radii - 1];
  for (int32_t iRad = 0; iRad < n_radii; ++iRad) flux_radius_cdf_out[iRad] = flux_radius_cdf_out[iRad] / integral;
  return 0;
}

int sart_host_detector_tables(const double* energy_ev, const double* t_si3n4, const double* t_si,
                              const double* t_al, int32_t n, const double* argon_energy_ev, const double* t_argon,
                              int32_t n_argon, double* x_kev_out, double* strongback_out, double* window_out,
                              double* argon_x_kev_out, double* gas_abs_out) {
  if (!energy_ev || !t_si3n4 || !t_si || !t_al || !argon_energy_ev || !t_argon || n < 2 || n_argon < 2)
    return fail(SART_ERR_INVALID_ARGUMENT, "sart_host_detector_tables: bad argument");
  for (int32_t i = 0; i < n; ++i) {  // raytracer.nim:1509-1518
    x_kev_out[i] = energy_ev[i] / 1000.0;
    strongback_out[i] = t_si[i] * t_al[i];
    window_out[i] = t_si3n4[i] * t_al[i];
  }
  for (int32_t i = 0; i < n_argon; ++i) {  // :1519-1521
    argon_x_kev_out[i] = argon_energy_ev[i] / 1000.0;
    gas_abs_out[i] = 1.0 - t_argon[i];
  }
  return 0;
}

int sart_host_trace_axion_wrapper(sart_context* ctx, sart_axion_t* ax_buf, int64_t buf_len, uint64_t seed,
                                  uint64_t ray_id_offset, uint32_t flags) {
  if (buf_len < 0) return fail(SART_ERR_INVALID_ARGUMENT, "bufLen < 0");
  sart_trace_params_t p;
  std::memset(&p, 0, sizeof p);
  p.n_rays = static_cast<uint64_t>(buf_len);
  p.seed = seed;
  p.ray_id_offset = ray_id_offset;
  p.flags = flags;
  p.image_nx = 256; p.image_ny = 256;  // unused by the record path
  p.image_x_min = 0.0; p.image_x_max = 14.0; p.image_y_min = 0.0; p.image_y_max = 14.0;
  int rc = sart_trace_records(ctx, &p, ax_buf);
  if (rc) g_err = sart_last_error();
  return rc;
}

int sart_host_perform_angular_scan(sart_context* ctx, const double* angles_deg, int32_t n_angles,
                                   uint64_t n_rays_per_angle, uint64_t seed, uint64_t ray_id_offset, uint32_t flags,
                                   double* fluxes_out, double* rel_fluxes_out) {
  if (!ctx || !angles_deg || n_angles < 1 || !fluxes_out)
    return fail(SART_ERR_INVALID_ARGUMENT, "sart_host_perform_angular_scan: bad argument");
  sart_setup_t setup;
  if (int rc = sart_get_setup(ctx, &setup)) { g_err = sart_last_error(); return rc; }
  sart_trace_params_t p;
  std::memset(&p, 0, sizeof p);
  p.n_rays = n_rays_per_angle;
  p.seed = seed;
  p.flags = flags;
  p.image_nx = 0; p.image_ny = 0;  // flux-only launches: the scan reads the sum of the weights alone (:2800), no image, no tile, no pilot
  // The reference scans a copy of fullSetup (`var tel = fullSetup.expSetup.telescope`, :2794-2797): the caller's setup is
  // unchanged afterwards.  Here the context is the setup, so the original angle is put back on every way out.
  int rc_scan = 0;
  for (int32_t i = 0; i < n_angles && rc_scan == 0; ++i) {  // raytracer.nim:2791-2800
    rc_scan = sart_set_telescope_angles(ctx, std::nan(""), angles_deg[i]);  // tel.telescope_turned_y = angle :2796
    if (rc_scan) { g_err = sart_last_error(); break; }
    p.ray_id_offset = ray_id_offset + static_cast<uint64_t>(i) * n_rays_per_angle;
    sart_summary_t sum;
    rc_scan = sart_trace_histogram(ctx, &p, nullptr, &sum);
    if (rc_scan) { g_err = sart_last_error(); break; }
    fluxes_out[i] = sum.v[SART_ACC_SUM_WEIGHTS];
  }
  if (int rc = sart_set_telescope_angles(ctx, std::nan(""), setup.telescope_turned_y_deg)) {
    if (!rc_scan) { g_err = sart_last_error(); rc_scan = rc; }
  }
  if (rc_scan) return rc_scan;
  if (rel_fluxes_out) {  // :2801-2802
    double maxFlux = fluxes_out[0];
    for (int32_t i = 1; i < n_angles; ++i) maxFlux = std::max(maxFlux, fluxes_out[i]);
    for (int32_t i = 0; i < n_angles; ++i) rel_fluxes_out[i] = fluxes_out[i] / maxFlux;
  }
  return 0;
}

int sart_host_angular_scan(sart_context* ctx, const double* angles_deg, int32_t n_angles, uint64_t n_rays, uint64_t seed,
                           uint64_t ray_id_offset, uint32_t flags, double* fluxes_out, double* rel_fluxes_out, double* flux_sq_out,
                           double* n_passed_out) {
  if (!ctx || !angles_deg || n_angles < 1 || !fluxes_out)
    return fail(SART_ERR_INVALID_ARGUMENT, "sart_host_angular_scan: bad argument");
  sart_trace_params_t p;
  std::memset(&p, 0, sizeof p);
  p.n_rays = n_rays;
  p.seed = seed;
  p.ray_id_offset = ray_id_offset;
  p.flags = flags;
  std::vector<double> rows(sart_angular_scan_len(n_angles));
  if (int rc = sart_trace_angular_scan(ctx, &p, angles_deg, n_angles, rows.data())) { g_err = sart_last_error(); return rc; }
  double maxFlux = 0.0;
  for (int32_t i = 0; i < n_angles; ++i) {
    const double* r = rows.data() + static_cast<size_t>(i) * SART_ASCAN_ROW;
    fluxes_out[i] = r[SART_ASCAN_SUM_WEIGHTS];   // axions.filterIt(it.passed).mapIt(it.weights).sum() :2800
    if (flux_sq_out) flux_sq_out[i] = r[SART_ASCAN_SUM_WEIGHTS_SQ];
    if (n_passed_out) n_passed_out[i] = r[SART_ASCAN_N_PASSED];
    maxFlux = (i == 0) ? fluxes_out[0] : std::max(maxFlux, fluxes_out[i]);
  }
  if (rel_fluxes_out)  // :2801-2802
    for (int32_t i = 0; i < n_angles; ++i) rel_fluxes_out[i] = fluxes_out[i] / maxFlux;
  return 0;
}

int sart_host_axion_mass_scan(sart_context* ctx, const double* masses_ev, int32_t n_masses, uint64_t n_rays, uint64_t seed,
                              uint64_t ray_id_offset, uint32_t flags, double* fluxes_out, double* flux_sq_out, double* n_passed_out) {
  if (!ctx || !masses_ev || n_masses < 1 || !fluxes_out)
    return fail(SART_ERR_INVALID_ARGUMENT, "sart_host_axion_mass_scan: bad argument");
  sart_setup_t setup;
  if (int rc = sart_get_setup(ctx, &setup)) { g_err = sart_last_error(); return rc; }
  sart_trace_params_t p;
  std::memset(&p, 0, sizeof p);
  p.n_rays = n_rays;
  p.seed = seed;
  p.ray_id_offset = ray_id_offset;
  p.flags = flags;
  p.image_nx = 256; p.image_ny = 256;
  p.image_x_min = 0.0; p.image_x_max = setup.chip_x_max; p.image_y_min = 0.0; p.image_y_max = setup.chip_y_max;
  if (setup.stage != SART_SK_GAS) {   // conversionProb (raytracer.nim:363-365) does not depend on m_a: one launch serves every mass
    sart_summary_t sum;
    if (int rc = sart_trace_histogram(ctx, &p, nullptr, &sum)) { g_err = sart_last_error(); return rc; }
    for (int32_t i = 0; i < n_masses; ++i) {
      fluxes_out[i] = sum.v[SART_ACC_SUM_WEIGHTS];
      if (flux_sq_out) flux_sq_out[i] = sum.v[SART_ACC_SUM_WEIGHTS_SQ];
      if (n_passed_out) n_passed_out[i] = sum.v[SART_ACC_N_PASSED];
    }
    return 0;
  }
  std::vector<double> rows(sart_mass_scan_len(n_masses));
  if (int rc = sart_trace_mass_scan(ctx, &p, masses_ev, n_masses, rows.data())) { g_err = sart_last_error(); return rc; }
  for (int32_t i = 0; i < n_masses; ++i) {
    const double* r = rows.data() + static_cast<size_t>(i) * SART_SCAN_ROW;
    fluxes_out[i] = r[SART_SCAN_SUM_WEIGHTS];
    if (flux_sq_out) flux_sq_out[i] = r[SART_SCAN_SUM_WEIGHTS_SQ];
    if (n_passed_out) n_passed_out[i] = r[SART_SCAN_N_PASSED];
  }
  return 0;
}

int sart_host_perform_axion_mass_scan(sart_context* ctx, const double* masses_ev, int32_t n_masses, uint64_t n_rays_per_mass,
                                      uint64_t seed, uint64_t ray_id_offset, uint32_t flags, double* fluxes_out) {
  // The meaning of rounds 1-3, back under the name of rounds 1-3 (round 4 had silently routed it through the fused scan): a host
  // loop in performAngularScan's shape - per mass set m_a, trace n_rays_per_mass FRESH rays (mass i on the ray ids
  // [offset + i n, offset + (i + 1) n)), sum the weights of the passed rays.  Flux-only launches; the context's own mass is put
  // back on every way out.
  if (!ctx || !masses_ev || n_masses < 1 || !fluxes_out)
    return fail(SART_ERR_INVALID_ARGUMENT, "sart_host_perform_axion_mass_scan: bad argument");
  sart_setup_t setup;
  if (int rc = sart_get_setup(ctx, &setup)) { g_err = sart_last_error(); return rc; }
  sart_trace_params_t p;
  std::memset(&p, 0, sizeof p);
  p.n_rays = n_rays_per_mass;
  p.seed = seed;
  p.flags = flags;
  p.image_nx = 0; p.image_ny = 0;   // the scan reads the sum of the weights alone
  int rc_scan = 0;
  for (int32_t i = 0; i < n_masses && rc_scan == 0; ++i) {
    rc_scan = sart_set_axion_mass(ctx, masses_ev[i]);
    if (rc_scan) { g_err = sart_last_error(); break; }
    p.ray_id_offset = ray_id_offset + static_cast<uint64_t>(i) * n_rays_per_mass;
    sart_summary_t sum;
    rc_scan = sart_trace_histogram(ctx, &p, nullptr, &sum);
    if (rc_scan) { g_err = sart_last_error(); break; }
    fluxes_out[i] = sum.v[SART_ACC_SUM_WEIGHTS];
  }
  if (int rc = sart_set_axion_mass(ctx, setup.m_axion)) {
    if (!rc_scan) { g_err = sart_last_error(); rc_scan = rc; }
  }
  return rc_scan;
}

int sart_host_h5_reflectivity_info(const char* path, int32_t* n_coatings, int32_t* n_angles, int32_t* n_energies,
                                   double* amin, double* amax, double* emin, double* emax) {
  H5& h = H5::get();
  if (!h.ok) return fail(SART_ERR_UNSUPPORTED, "libhdf5 could not be loaded");
  if (!path) return fail(SART_ERR_INVALID_ARGUMENT, "path is NULL");
  const H5::hid f = h.H5Fopen(path, 0u /* H5F_ACC_RDONLY */, 0);
  if (f < 0) return fail(SART_ERR_INVALID_ARGUMENT, std::string("cannot open ") + path);  // IOError in the reference (:1174)
  std::vector<double> energy, angles;
  const long long nE = h.read_all(f, "Energy", energy), nA = h.read_all(f, "Angles", angles);
  bool single = false;
  const int nC = h5_count_coatings(h, f, single);
  h.H5Fclose(f);
  if (nE < 2 || nA < 2 || nC < 1) return fail(SART_ERR_INVALID_ARGUMENT, "not a reflectivity file (datasets Energy / Angles / Reflectivity*)");
  if (n_coatings) *n_coatings = nC;
  if (n_angles) *n_angles = static_cast<int32_t>(nA);
  if (n_energies) *n_energies = static_cast<int32_t>(nE);
  if (amin) *amin = *std::min_element(angles.begin(), angles.end());   // (angles.min, angles.max) :1183
  if (amax) *amax = *std::max_element(angles.begin(), angles.end());
  if (emin) *emin = *std::min_element(energy.begin(), energy.end());
  if (emax) *emax = *std::max_element(energy.begin(), energy.end());
  return 0;
}

int sart_host_h5_read_reflectivity(const char* path, double* data_out) {
  H5& h = H5::get();
  if (!h.ok) return fail(SART_ERR_UNSUPPORTED, "libhdf5 could not be loaded");
  if (!path || !data_out) return fail(SART_ERR_INVALID_ARGUMENT, "NULL argument");
  const H5::hid f = h.H5Fopen(path, 0u, 0);
  if (f < 0) return fail(SART_ERR_INVALID_ARGUMENT, std::string("cannot open ") + path);
  std::vector<double> energy, angles, buf;
  const long long nE = h.read_all(f, "Energy", energy), nA = h.read_all(f, "Angles", angles);
  bool single = false;
  const int nC = h5_count_coatings(h, f, single);
  int rc = 0;
  for (int c = 0; c < nC && !rc; ++c) {
    const std::string name = single ? "Reflectivity" : "Reflectivity" + std::to_string(c);
    if (h.read_all(f, name.c_str(), buf) != nE * nA) { rc = fail(SART_ERR_INVALID_ARGUMENT, name + ": unexpected size"); break; }
    // memory is [angle][energy] whatever the declared shape says (reshape(reflDset.shape), :1180)
    std::memcpy(data_out + static_cast<size_t>(c) * nA * nE, buf.data(), static_cast<size_t>(nA * nE) * sizeof(double));
  }
  h.H5Fclose(f);
  return rc;
}

int sart_host_h5_write_reflectivity(const char* path, int32_t nC, int32_t nA, int32_t nE, const double* angles,
                                    const double* energies, const double* data) {
  H5& h = H5::get();
  if (!h.ok) return fail(SART_ERR_UNSUPPORTED, "libhdf5 could not be loaded");
  if (!path || !angles || !energies || !data || nC < 1 || nA < 2 || nE < 2) return fail(SART_ERR_INVALID_ARGUMENT, "bad argument");
  const H5::hid f = h.H5Fcreate(path, 2u /* H5F_ACC_TRUNC */, 0, 0);
  if (f < 0) return fail(SART_ERR_INVALID_ARGUMENT, std::string("cannot create ") + path);
  auto put = [&](const std::string& name, unsigned long long d0, unsigned long long d1, const double* src) {
    const unsigned long long dims[2] = {d0, d1};
    const H5::hid sp = h.H5Screate_simple(2, dims, nullptr);
    const H5::hid d = h.H5Dcreate2(f, name.c_str(), h.native_double, sp, 0, 0, 0);
    const int rc = (d < 0) ? -1 : h.H5Dwrite(d, h.native_double, 0, 0, 0, src);
    if (d >= 0) h.H5Dclose(d);
    h.H5Sclose(sp);
    return rc;
  };
  int rc = put("Energy", nE, 1, energies) | put("Angles", nA, 1, angles);
  for (int c = 0; c < nC; ++c)   // declared (nE, nA), written from [angle][energy] memory - as the reference's tools do
    rc |= put(nC == 1 ? "Reflectivity" : "Reflectivity" + std::to_string(c), nE, nA, data + static_cast<size_t>(c) * nA * nE);
  h.H5Fclose(f);
  return rc < 0 ? fail(SART_ERR_INTERNAL, "HDF5 write failed") : 0;
}

int sart_host_containment_radii(const double* counts, const double* weights, int32_t n_bins, double radial_max,
                                double* r_sigma1, double* r_sigma2, double* r_sigma1_w, double* r_sigma2_w) {
  if (!counts || !weights || n_bins < 1 || !(radial_max > 0.0))
    return fail(SART_ERR_INVALID_ARGUMENT, "sart_host_containment_radii: bad argument");
  const double bin = radial_max / n_bins;
  double n = 0.0, sumW = 0.0;
  for (int32_t k = 0; k < n_bins; ++k) { n += counts[k]; sumW += weights[k]; }
  // unweighted: pointR[round(0.68 n) - 1] of the sorted radii (raytracer.nim:2464-2473)
  const double k1 = std::round(n * 0.68), k2 = std::round(n * 0.955);
  double r1 = 0.0, r2 = 0.0, cum = 0.0;
  bool got1 = false, got2 = false;
  for (int32_t k = 0; k < n_bins; ++k) {
    cum += counts[k];
    if (!got1 && k1 >= 1.0 && cum >= k1) { r1 = (k + 1) * bin; got1 = true; }
    if (!got2 && k2 >= 1.0 && cum >= k2) { r2 = (k + 1) * bin; got2 = true; }
  }
  // weighted: the last radius with cumulative weight < 0.68 / 0.955 of the total (:2511-2524)
  double r1w = 0.0, r2w = 0.0, cw = 0.0;
  for (int32_t k = 0; k < n_bins; ++k) {
    cw += weights[k];
    if (counts[k] > 0.0) {
      if (cw < sumW * 0.68) r1w = (k + 1) * bin;
      else if (cw < sumW * 0.955) r2w = (k + 1) * bin;
    }
  }
  if (r_sigma1) *r_sigma1 = r1;
  if (r_sigma2) *r_sigma2 = r2;
  if (r_sigma1_w) *r_sigma1_w = r1w;
  if (r_sigma2_w) *r_sigma2_w = r2w;
  return 0;
}

int sart_host_write_image_csv(const char* path, const double* image, int32_t width, double chip_max, double rSigma1,
                              double rSigma2, double* flux_out) {
  if (!path || !image || width < 1) return fail(SART_ERR_INVALID_ARGUMENT, "sart_host_write_image_csv: bad argument");
  std::FILE* f = std::fopen(path, "w");
  if (!f) return fail(SART_ERR_INVALID_ARGUMENT, std::string("cannot open ") + path);
  // raytracer.nim:887-899
  std::fprintf(f, "x,y,photon flux,yr0,yr02,x-position [mm],y-position [mm],xr,xrneg,yr,xr2,xrneg2,yr2\n");
  const long n = static_cast<long>(width) * width;
  const double offset = chip_max / 2.0;  // ChipCenterX :881
  double flux = 0.0;
  for (int32_t y = 0; y < width; ++y) {
    for (int32_t x = 0; x < width; ++x) {
      const long i = static_cast<long>(y) * width + x;
      const double z = image[i];  // objectsToDraw[y, x] :874
      flux += z;
      // linspace(-r, r, n) :883-884
      const double t = (n > 1) ? static_cast<double>(i) / static_cast<double>(n - 1) : 0.0;
      const double yr0 = -rSigma1 + 2.0 * rSigma1 * t, yr02 = -rSigma2 + 2.0 * rSigma2 * t;
      const double xr = std::sqrt(rSigma1 * rSigma1 - yr0 * yr0), xr2 = std::sqrt(rSigma2 * rSigma2 - yr02 * yr02);
      std::fprintf(f, "%d,%d,%.17g,%.17g,%.17g,%.17g,%.17g,%.17g,%.17g,%.17g,%.17g,%.17g,%.17g\n", x, y, z, yr0, yr02,
                   x * chip_max / width, y * chip_max / width, xr + offset, -xr + offset, yr0 + offset, xr2 + offset,
                   -xr2 + offset, yr02 + offset);
    }
  }
  std::fclose(f);
  if (flux_out) *flux_out = flux;
  return 0;
}

// ---- solar emission-table producer: host side (readOpacityFile.nim) -------------------------------------------------

// First loop of calculateOpacities (readOpacityFile.nim:655-705): per-radius number densities, electron density and the
// nearest OPCD grid points from the columns of the solar-model file.
int sart_host_solar_zones(const double* temp_K, const double* rho, const double* mass_fractions, int32_t n_radii,
                          sart_solar_zone_t* zones_out) {
  if (!temp_K || !rho || !mass_fractions || !zones_out || n_radii < 1)
    return fail(SART_ERR_INVALID_ARGUMENT, "sart_host_solar_zones: bad argument");
  // const atomicMass / charges :120-132, in the order of the model-file columns (H1 He4 He3 C12 C13 N14 N15 O16 O17 O18 Ne ... Ni)
  static const double atomic_mass[29] = {1.0078,  4.0026,  3.0160,  12.0000, 13.0033, 14.0030, 15.0001, 15.9949, 16.9991, 17.9991,
                                         20.1797, 22.9897, 24.3055, 26.9815, 28.085,  30.9737, 32.0675, 35.4515, 39.8775, 39.0983,
                                         40.078,  44.9559, 47.867,  50.9415, 51.9961, 54.9380, 55.845,  58.9331, 58.6934};
  const double amu = 1.6605e-24;  // :651
  int temperature = 0, n_e_int = 0;  // declared outside the loop in the reference (:622-629): a radius without a grid point
                                     // within one step keeps the previous radius' value
  for (int32_t i = 0; i < n_radii; ++i) {
    if (!(temp_K[i] > 0.0) || !(rho[i] > 0.0)) return fail(SART_ERR_INVALID_ARGUMENT, "sart_host_solar_zones: Temp and Rho must be positive");
    const double* x = mass_fractions + static_cast<size_t>(i) * 29;
    const double per_amu = rho[i] / amu;
    sart_solar_zone_t& z = zones_out[i];
    z.radius_frac = static_cast<double>(i) * 0.0005 + 0.0015;                                             // :698
    z.temp_K = temp_K[i];
    z.rho = rho[i];
    z.n_H = (x[0] / atomic_mass[0]) * per_amu;                                                           // :661
    z.n_He = (x[1] + x[2]) / ((atomic_mass[1] * x[1] + atomic_mass[2] * x[2]) / (x[1] + x[2])) * rho[i] / amu;  // :662-667
    double n_e = 0.0;
    for (int k = 0; k < 29; ++k) {
      const double charge = (k == 0) ? 1.0 : (k <= 2) ? 2.0 : (k <= 4) ? 6.0 : (k <= 6) ? 7.0 : (k <= 9) ? 8.0 : static_cast<double>(k);  // :129-132
      n_e += per_amu * charge * x[k] / atomic_mass[k];                                                   // :681-683
    }
    z.n_e = n_e;
    const double lt = std::log10(temp_K[i]) / 0.025;
    for (int it = 0; it <= 90; ++it)
      if (std::fabs(lt - static_cast<double>(140 + 2 * it)) <= 1.0) temperature = 140 + 2 * it;          // :686-689
    const double ln = std::log10(n_e) / 0.25;
    for (int in = 0; in <= 17; ++in)
      if (std::fabs(ln - static_cast<double>(74 + in * 2)) <= 1.0) n_e_int = 74 + in * 2;                // :692-695
    z.temp_index = temperature;
    z.ne_index = n_e_int;
  }
  return 0;
}

// getFluxFractionR (readOpacityFile.nim:535-584): flux spectrum at Earth in 1/(keV y m^2), summed over the radial zones.
int sart_host_flux_spectrum(const double* em_rates, int32_t n_radii, const double* energies_kev, int32_t n_energies,
                            double* diff_flux_out) {
  if (!em_rates || !energies_kev || !diff_flux_out || n_radii < 1 || n_energies < 1)
    return fail(SART_ERR_INVALID_ARGUMENT, "sart_host_flux_spectrum: bad argument");
  const double pi = 3.14159265358979323846;
  const double r_sun = 6.957e11, r_sunearth = 1.5e14, hbar = 6.582119514e-25, keV2cm = 1.97327e-8;   // :540-545
  const double factor = std::pow(r_sun * 0.1 / keV2cm, 3.0) / (std::pow(0.1 * r_sunearth, 2.0) * (1.0e6 * hbar)) /
                        (3.1709791983765E-8 * 1.0e-4);                                                // :546-548
  for (int32_t e = 0; e < n_energies; ++e) {
    const double e_keV = energies_kev[e];
    double sum = 0.0, r_last = 0.0;
    for (int32_t r = 0; r < n_radii; ++r) {
      const double r_perc = static_cast<double>(r) * 0.0005 + 0.0015;
      // both branches of the `if e_keV > 0.4` are the same expression (:569-575)
      sum += em_rates[static_cast<size_t>(r) * n_energies + e] * (r_perc - r_last) * r_perc * r_perc * e_keV * e_keV * 0.5 / (pi * pi);
      r_last = r_perc;
    }
    diff_flux_out[e] = sum * factor;
  }
  return 0;
}

}  // extern "C"
