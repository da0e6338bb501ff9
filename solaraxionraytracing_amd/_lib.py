"""ctypes bindings of the C-ABI (include/sart.h, include/sart_host.h).

The product path is the HIP library ``libsart.so``; there is no Python or CPU fallback.  If the
library has not been built (``python -c 'import __graft_entry__ as g; g.build()'`` or
``make -C solaraxionraytracing_amd/csrc``) loading fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIBSART_PATH = os.environ.get("SART_LIBSART", os.path.join(_PKG_DIR, "libsart.so"))   # override: A/B timing of kernel builds
LIBSART_HOST_PATH = os.path.join(_PKG_DIR, "libsart_host.so")

SART_MAX_SHELLS = 64
SART_MAX_COATINGS = 8
SART_ACC_COUNT = 24
SART_ABI_VERSION = 5

# enums (values of include/sart.h)
ES_CAST, ES_BABYIAXO = 0, 1
SK_VACUUM, SK_GAS = 0, 1
TK_LLNL, TK_XMM, TK_CUSTOM_BABYIAXO, TK_ABRIXAS, TK_OTHER = 0, 1, 2, 3, 4
DK_INGRID2017, DK_INGRID2018, DK_INGRIDIAXO = 0, 1, 2
HT_NONE, HT_CROSS, HT_STAR, HT_CIRCLE, HT_SQUARE, HT_DIAMOND = range(6)
RK_EFFECTIVE_AREA, RK_SINGLE_COATING, RK_MULTI_COATING = 0, 1, 2
CF_IGNORE_DET_WINDOW = 1 << 0
CF_IGNORE_GAS_ABS = 1 << 1
CF_IGNORE_REFLECTION = 1 << 2
CF_IGNORE_CONV_PROB = 1 << 3
CF_XRAY_TEST = 1 << 4
CF_READ_MAGNET_CONFIG = 1 << 5
CF_READ_DET_INSTALL_CONFIG = 1 << 6

RADIUS_GUIDE_ENTRIES = 3074   # csrc/sart_device.h: kRadiusGuideEntries (2049 entries of 1/2048 buckets + 1025 of 1/32768 buckets for u >= 31/32)
ENERGY_GUIDE_ENTRIES = 2594   # csrc/sart_device.h: kEnergyGuideEntries (992 uniform + 1600 logarithmic buckets, bracketed)
ACCUM_F64, ACCUM_FIXED64 = 0, 1
ACC_HI = dict(SUM_WEIGHTS=12, SUM_X=13, SUM_Y=14, SUM_R=15, SUM_WEIGHTS_SQ=16)   # SART_ACC_SUM_*_HI: high limbs of the raw FIXED64 accumulator
# fused mass scan (include/sart.h: SART_SCAN_*): (n_masses + 1) rows of SCAN_ROW slots; the last row holds the counters of SCAN_SHARED
SCAN_ROW = 8
SCAN = dict(SUM_WEIGHTS=0, SUM_WEIGHTS_SQ=1, N_PASSED=2)
SCAN_HI = dict(SUM_WEIGHTS=4, SUM_WEIGHTS_SQ=5)
SCAN_SHARED = dict(N_RAYS=0, N_REACHED_TELESCOPE=1, N_SHELL_SELECTED=2, N_HIT_NICKEL=3, N_ON_DETECTOR=4)
# fused angular scan (include/sart.h: SART_ASCAN_*): (n_angles + 1) rows of ASCAN_ROW slots; the last row holds the counters of ASCAN_SHARED
ASCAN_ROW = 8
ASCAN = dict(SUM_WEIGHTS=0, SUM_WEIGHTS_SQ=1, N_PASSED=2, N_SHELL_SELECTED=3, N_HIT_NICKEL=6, N_PASSED_TILL_WINDOW=7)
ASCAN_HI = dict(SUM_WEIGHTS=4, SUM_WEIGHTS_SQ=5)
ASCAN_SHARED = dict(N_RAYS=0, N_REACHED_TELESCOPE=1)
ASCAN_MAX_ANGLES = 32   # csrc/sart_device.h: kAScanMaxAngles (angles per kernel launch)
FIXED_LIMB_BITS = 40

ACC = dict(SUM_WEIGHTS=0, N_PASSED=1, N_PASSED_TILL_WINDOW=2, N_HIT_NICKEL=3, SUM_X=4, SUM_Y=5, SUM_R=6,
           SUM_WEIGHTS_SQ=7, N_RAYS=8, N_REACHED_TELESCOPE=9, N_SHELL_SELECTED=10, N_OUTSIDE_IMAGE=11)

SART_ERR_INVALID_ARGUMENT, SART_ERR_NO_DEVICE = -1, -2
SART_ERR_NOT_READY, SART_ERR_UNSUPPORTED, SART_ERR_OUT_OF_MEMORY, SART_ERR_INTERNAL, SART_ERR_ACCUMULATOR = -3, -4, -5, -6, -7

_d = C.c_double
_i = C.c_int32


class Setup(C.Structure):
    """sart_setup_t"""
    _fields_ = [
        ("experiment", _i), ("stage", _i), ("telescope_kind", _i), ("detector_kind", _i),
        ("magnet_B", _d), ("magnet_lengthB", _d), ("magnet_lengthColdbore", _d), ("magnet_radiusCB", _d),
        ("magnet_pGasRoom", _d), ("magnet_tGas", _d),
        ("pipe_cb_vt3_length", _d), ("pipe_cb_vt3_radius", _d), ("pipe_vt3_xrt_length", _d),
        ("pipe_vt3_xrt_radius", _d), ("pipes_turned_deg", _d), ("distance_cb_axis_xrt_axis", _d),
        ("optics_entrance", _d * 3), ("optics_exit", _d * 3),
        ("telescope_turned_x_deg", _d), ("telescope_turned_y_deg", _d),
        ("n_shells", _i), ("hole_type", _i), ("number_of_holes", _i), ("reflectivity_kind", _i),
        ("all_r1", _d * SART_MAX_SHELLS), ("all_thickness", _d * SART_MAX_SHELLS),
        ("all_xsep", _d * SART_MAX_SHELLS), ("all_angles_deg", _d * SART_MAX_SHELLS),
        ("l_mirror", _d), ("hole_in_optics", _d),
        ("n_coatings", _i), ("coating_layers", _i * SART_MAX_COATINGS),
        ("distance_detector_xrt", _d), ("distance_window_focal_plane", _d), ("lateral_shift", _d),
        ("transversal_shift", _d),
        ("radius_window", _d), ("number_of_strips", _i), ("_pad0", _i), ("open_aperture_ratio", _d),
        ("strip_dist_window", _d), ("strip_width_window", _d), ("theta_rad", _d), ("depth_det", _d),
        ("test_active", _i), ("test_parallel", _i), ("test_energy", _d), ("test_distance", _d),
        ("test_radius", _d), ("test_off_axis_up", _d), ("test_off_axis_left", _d), ("test_activity", _d),
        ("test_length_col", _d),
        ("distance_sun_earth", _d), ("radius_sun", _d), ("room_temp", _d), ("m_axion", _d), ("g_agamma", _d),
        ("chip_x_max", _d), ("chip_y_max", _d),
    ]

    def copy(self) -> "Setup":
        out = Setup()
        C.memmove(C.byref(out), C.byref(self), C.sizeof(Setup))
        return out


class Axion(C.Structure):
    """sart_axion_t (208 bytes, the reference's Axion object)."""
    _fields_ = [
        ("passed", C.c_uint8), ("passedTillWindow", C.c_uint8), ("hitNickel", C.c_uint8), ("_pad0", C.c_uint8 * 5),
        ("pointdataX", _d), ("pointdataY", _d), ("pointdataXBefore", _d), ("pointdataYBefore", _d),
        ("pointdataR", _d), ("weights", _d), ("weightsAll", _d), ("transmissionMagnet", _d), ("yawAngles", _d),
        ("pixvalsX", _d), ("pixvalsY", _d), ("radii", _d), ("energiesAx", _d), ("energiesAxAll", _d),
        ("energiesAxWindow", _d), ("kinds", C.c_uint8), ("kindsWindow", C.c_uint8), ("_pad1", C.c_uint8 * 6),
        ("transProbWindow", _d), ("transProbArgon", _d), ("transProbDetector", _d), ("transProbMagnet", _d),
        ("deviationDet", _d), ("shellNumber", C.c_int64), ("energiesPre", _d), ("emratesPre", _d), ("reflect", _d),
    ]


assert C.sizeof(Axion) == 208

# numpy view of an Axion buffer
import numpy as _np  # noqa: E402

AXION_DTYPE = _np.dtype({
    "names": [n for n, _ in Axion._fields_ if not n.startswith("_pad")],
    "formats": [("u1" if t is C.c_uint8 else "i8" if t is C.c_int64 else "f8")
                for n, t in Axion._fields_ if not n.startswith("_pad")],
    "offsets": [getattr(Axion, n).offset for n, _ in Axion._fields_ if not n.startswith("_pad")],
    "itemsize": 208,
})


class RecordCounts(C.Structure):
    """sart_record_counts_t"""
    _fields_ = [("n_rays", C.c_uint64), ("n_passed", C.c_uint64), ("n_passed_till_window", C.c_uint64), ("n_hit_nickel", C.c_uint64)]


class TraceParams(C.Structure):
    """sart_trace_params_t"""
    _fields_ = [
        ("n_rays", C.c_uint64), ("seed", C.c_uint64), ("ray_id_offset", C.c_uint64), ("flags", C.c_uint32),
        ("image_nx", _i), ("image_ny", _i), ("accumulate", _i),
        ("image_x_min", _d), ("image_x_max", _d), ("image_y_min", _d), ("image_y_max", _d),
        ("spectra", _i), ("n_radial_bins", _i), ("radial_max", _d),
    ]


class Summary(C.Structure):
    _fields_ = [("v", _d * SART_ACC_COUNT)]


class FixedQuanta(C.Structure):
    """sart_fixed_quanta_t"""
    _fields_ = [("weight", _d), ("weight_sq", _d), ("position", _d), ("reflect", _d)]


class MagnetConfig(C.Structure):
    _fields_ = [(n, _d) for n in ("B", "radiusCB", "lengthColdbore", "lengthB", "pGasRoom", "tGas")]


class TestSourceConfig(C.Structure):
    _fields_ = [("active", _i), ("parallel", _i)] + [(n, _d) for n in (
        "energy", "distance", "radius", "offAxisUp", "offAxisLeft", "activity", "lengthCol")]


class DetectorInstallConfig(C.Structure):
    _fields_ = [(n, _d) for n in ("distanceDetectorXRT", "distanceWindowFocalPlane", "lateralShift", "transversalShift")]


class SolarZone(C.Structure):
    """sart_solar_zone_t (include/sart_emission.h)"""
    _fields_ = [(n, _d) for n in ("radius_frac", "temp_K", "rho", "n_e", "n_H", "n_He")] + [("temp_index", _i), ("ne_index", _i)]


class EmissionParams(C.Structure):
    """sart_emission_params_t"""
    _fields_ = [("g_ae", _d), ("g_agamma", _d), ("g_anuclei", _d), ("terms", C.c_uint32), ("_pad", C.c_uint32)]


class OpacityTables(C.Structure):
    """sart_opacity_tables_t (include/sart_emission.h): host pointers into arrays the owner keeps alive"""
    _fields_ = [("u_mesh", C.POINTER(_d)), ("n_mesh", _i), ("n_slots", _i), ("slot_of_zone", C.POINTER(_i)),
                ("element_z", C.POINTER(_i)), ("n_elements", _i), ("_pad", _i), ("table_y_begin", C.POINTER(C.c_int64)),
                ("table_x_begin", C.POINTER(C.c_int64)), ("table_len", C.POINTER(_i)), ("table_x", C.POINTER(_d)),
                ("table_y", C.POINTER(_d)), ("n_table_x", C.c_int64), ("n_table_y", C.c_int64)]


# SART_EM_* term bits, in the order of the component planes
EM_TERMS = ("compton", "term1", "ee_brems", "free_free", "primakoff", "long_plasmon", "trans_plasmon", "iron57")
EM_ALL = 0xFF

_P = C.POINTER
_dp = _P(_d)

# name -> (restype, argtypes): every symbol include/sart.h declares
SART_SYMBOLS = {
    "sart_abi_version": (C.c_int, []),
    "sart_last_error": (C.c_char_p, []),
    "sart_create": (C.c_int, [C.c_int, _P(C.c_void_p)]),
    "sart_destroy": (C.c_int, [C.c_void_p]),
    "sart_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "sart_synchronize": (C.c_int, [C.c_void_p]),
    "sart_set_setup": (C.c_int, [C.c_void_p, _P(Setup)]),
    "sart_get_setup": (C.c_int, [C.c_void_p, _P(Setup)]),
    "sart_set_telescope_angles": (C.c_int, [C.c_void_p, _d, _d]),
    "sart_set_axion_mass": (C.c_int, [C.c_void_p, _d]),
    "sart_set_solar_tables": (C.c_int, [C.c_void_p, _dp, _dp, _dp, _i, _i]),
    "sart_set_solar_tables_device": (C.c_int, [C.c_void_p, C.c_void_p, _dp, _dp, _i, _i]),
    "sart_get_solar_tables": (C.c_int, [C.c_void_p, _dp, _dp, C.c_void_p, C.c_void_p]),
    "sart_set_reflectivity": (C.c_int, [C.c_void_p, _i, _i, _i, _d, _d, _d, _d, _dp]),
    "sart_set_detector_tables": (C.c_int, [C.c_void_p, _dp, _dp, _i, _dp, _dp, _i, _dp, _dp, _i]),
    "sart_trace_records": (C.c_int, [C.c_void_p, _P(TraceParams), C.c_void_p]),
    "sart_trace_records_device": (C.c_int, [C.c_void_p, _P(TraceParams), C.c_void_p]),
    "sart_trace_records_passed": (C.c_int, [C.c_void_p, _P(TraceParams), C.c_void_p, C.c_uint64, _P(RecordCounts)]),
    "sart_trace_records_passed_device": (C.c_int, [C.c_void_p, _P(TraceParams), C.c_void_p, C.c_uint64, C.c_void_p]),
    "sart_release_scratch": (C.c_int, [C.c_void_p]),
    "sart_trace_histogram_device": (C.c_int, [C.c_void_p, _P(TraceParams), C.c_void_p]),
    "sart_trace_histogram": (C.c_int, [C.c_void_p, _P(TraceParams), _dp, _P(Summary)]),
    "sart_trace_histogram_spectra": (C.c_int, [C.c_void_p, _P(TraceParams), _dp, _P(Summary), _dp]),
    "sart_set_accumulation_mode": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "sart_get_accumulation_mode": (C.c_int, [C.c_void_p, _P(C.c_int)]),
    "sart_get_fixed_quanta": (C.c_int, [C.c_void_p, _P(FixedQuanta)]),
    "sart_finalize_accumulator_device": (C.c_int, [C.c_void_p, _P(TraceParams), C.c_void_p, C.c_void_p]),
    "sart_rollover_accumulator_device": (C.c_int, [C.c_void_p, _P(TraceParams), C.c_void_p, C.c_void_p]),
    "sart_finalize_accumulator_limbs_device": (C.c_int, [C.c_void_p, _P(TraceParams), C.c_void_p, C.c_void_p, C.c_void_p]),
    "sart_trace_mass_scan_device": (C.c_int, [C.c_void_p, _P(TraceParams), _dp, _i, C.c_void_p]),
    "sart_trace_mass_scan": (C.c_int, [C.c_void_p, _P(TraceParams), _dp, _i, _dp]),
    "sart_finalize_mass_scan_device": (C.c_int, [C.c_void_p, _P(TraceParams), _dp, _i, C.c_void_p, C.c_void_p]),
    "sart_trace_angular_scan_device": (C.c_int, [C.c_void_p, _P(TraceParams), _dp, _i, C.c_void_p]),
    "sart_trace_angular_scan": (C.c_int, [C.c_void_p, _P(TraceParams), _dp, _i, _dp]),
    "sart_finalize_angular_scan_device": (C.c_int, [C.c_void_p, _P(TraceParams), _i, C.c_void_p, C.c_void_p]),
    "sart_reduce_across_devices": (C.c_int, [_P(C.c_void_p), _P(C.c_void_p), _i, C.c_size_t, _i]),
    "sart_enable_kernel_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "sart_get_kernel_timing": (C.c_int, [C.c_void_p, _dp, _P(C.c_int64)]),
    "sart_device_info": (C.c_int, [C.c_void_p, _P(_i), _P(_i), C.c_char_p, C.c_size_t]),
    "sart_build_id": (C.c_char_p, []),
    # include/sart_emission.h
    "sart_emission_default_params": (None, [_P(EmissionParams)]),
    "sart_emission_table": (C.c_int, [C.c_void_p, _P(SolarZone), _i, _dp, _i, _dp, _P(EmissionParams), _dp, _dp]),
    "sart_emission_table_device": (C.c_int, [C.c_void_p, _P(SolarZone), _i, _dp, _i, C.c_void_p, _P(EmissionParams),
                                             C.c_void_p, C.c_void_p]),
    "sart_emission_to_solar_tables": (C.c_int, [C.c_void_p, _P(SolarZone), _i, _dp, _i, C.c_void_p, _P(EmissionParams)]),
    "sart_emission_abs_coefs": (C.c_int, [C.c_void_p, _P(SolarZone), _i, _dp, _dp, _i, _P(OpacityTables), _dp]),
    "sart_emission_abs_coefs_device": (C.c_int, [C.c_void_p, _P(SolarZone), _i, _dp, _dp, _i, _P(OpacityTables), C.c_void_p]),
    "sart_emission_to_solar_tables_opcd": (C.c_int, [C.c_void_p, _P(SolarZone), _i, _dp, _dp, _i, _P(OpacityTables),
                                                     _P(EmissionParams)]),
    "sart_emission_abs_coefs_last_kernel_ms": (_d, []),
    "sart_emission_last_kernel_ms": (_d, []),
}

# every symbol include/sart_host.h declares
SART_HOST_SYMBOLS = {
    "sart_host_last_error": (C.c_char_p, []),
    "sart_host_new_full_setup": (C.c_int, [_i, _i, _i, _i, C.c_uint32, _P(MagnetConfig), _P(TestSourceConfig),
                                           _P(DetectorInstallConfig), _P(Setup)]),
    "sart_host_calc_window_vals": (C.c_int, [_d, _i, _d, _dp, _dp]),
    "sart_host_build_cdfs": (C.c_int, [_dp, _dp, _dp, _i, _i, _dp, _dp]),
    "sart_host_detector_tables": (C.c_int, [_dp, _dp, _dp, _dp, _i, _dp, _dp, _i, _dp, _dp, _dp, _dp, _dp]),
    "sart_host_perform_axion_mass_scan": (C.c_int, [C.c_void_p, _dp, _i, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, _dp]),
    "sart_host_axion_mass_scan": (C.c_int, [C.c_void_p, _dp, _i, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, _dp, _dp, _dp]),
    "sart_host_angular_scan": (C.c_int, [C.c_void_p, _dp, _i, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, _dp, _dp, _dp, _dp]),
    "sart_host_h5_reflectivity_info": (C.c_int, [C.c_char_p, _P(_i), _P(_i), _P(_i), _dp, _dp, _dp, _dp]),
    "sart_host_h5_read_reflectivity": (C.c_int, [C.c_char_p, _dp]),
    "sart_host_h5_write_reflectivity": (C.c_int, [C.c_char_p, _i, _i, _i, _dp, _dp, _dp]),
    "sart_host_containment_radii": (C.c_int, [_dp, _dp, _i, _d, _dp, _dp, _dp, _dp]),
    "sart_host_write_image_csv": (C.c_int, [C.c_char_p, _dp, _i, _d, _d, _d, _dp]),
    "sart_host_trace_axion_wrapper": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_uint64, C.c_uint64, C.c_uint32]),
    "sart_host_perform_angular_scan": (C.c_int, [C.c_void_p, _dp, _i, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32,
                                                 _dp, _dp]),
    "sart_host_solar_zones": (C.c_int, [_dp, _dp, _dp, _i, _P(SolarZone)]),
    "sart_host_flux_spectrum": (C.c_int, [_dp, _i, _dp, _i, _dp]),
    "sart_host_solar_number_densities": (C.c_int, [_dp, _dp, _i, _dp]),
    "sart_host_opcd_read_mesh": (C.c_int, [C.c_char_p, _dp, _i, _P(_i)]),
    "sart_host_opcd_file_info": (C.c_int, [C.c_char_p, _P(_i), _P(_i), _P(_i), _P(_i), _P(_i), _i]),
    "sart_host_opcd_read_table": (C.c_int, [C.c_char_p, _i, _dp, _dp, _i, _P(_i)]),
    "sart_host_opcd_load": (C.c_int, [C.c_char_p, _P(SolarZone), _i, _i, _P(C.c_void_p)]),
    "sart_host_opcd_tables": (_P(OpacityTables), [C.c_void_p]),
    "sart_host_opcd_slot": (C.c_int, [C.c_void_p, _i, _P(_i), _P(_i)]),
    "sart_host_opcd_free": (None, [C.c_void_p]),
}


def _bind(lib, table):
    for name, (res, args) in table.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    return lib


_sart = None
_host = None


def load_sart() -> C.CDLL:
    """Loads libsart.so (HIP kernels + C-ABI).  Raises if it has not been built."""
    global _sart
    if _sart is None:
        if not os.path.exists(LIBSART_PATH):
            raise RuntimeError(
                f"{LIBSART_PATH} is missing: build the HIP extension first (python -c 'import __graft_entry__ as g; "
                "g.build()').  There is no CPU fallback for the hot path.")
        _sart = _bind(C.CDLL(LIBSART_PATH, mode=C.RTLD_GLOBAL), SART_SYMBOLS)
    return _sart


def load_host() -> C.CDLL:
    """Loads libsart_host.so (C++ host mirror of the reference's setup / driver layer)."""
    global _host
    if _host is None:
        load_sart()  # libsart_host.so depends on libsart.so
        if not os.path.exists(LIBSART_HOST_PATH):
            raise RuntimeError(f"{LIBSART_HOST_PATH} is missing: run __graft_entry__.build()")
        _host = _bind(C.CDLL(LIBSART_HOST_PATH), SART_HOST_SYMBOLS)
    return _host


def build_id() -> str:
    """sart_build_id(): hash of the device sources + compile flags the loaded libsart.so was built from."""
    return load_sart().sart_build_id().decode()


class SartError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"sart error {code}: {msg}")
        self.code = code


def check(rc: int, host: bool = False):
    if rc != 0:
        lib = load_host() if host else load_sart()
        msg = (lib.sart_host_last_error() if host else lib.sart_last_error()) or b""
        raise SartError(rc, msg.decode(errors="replace"))


def as_dp(a):
    """float64 C-contiguous numpy array -> double*"""
    assert a.dtype == _np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_dp)
