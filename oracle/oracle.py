"""ctypes wrapper of the CPU oracle (oracle/libsart_oracle.so).  TEST INFRASTRUCTURE: only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from solaraxionraytracing_amd import _lib
from solaraxionraytracing_amd._lib import AXION_DTYPE, Setup, TraceParams

_DIR = os.path.dirname(os.path.abspath(__file__))
_d, _i, _dp = C.c_double, C.c_int32, C.POINTER(C.c_double)


class OracleTables(C.Structure):
    """sart_oracle_tables_t"""
    _fields_ = [
        ("flux_radius_cdf", _dp), ("diff_flux_cdfs", _dp), ("energies_kev", _dp), ("n_radii", _i), ("n_energies", _i),
        ("refl_data", _dp), ("refl_n_coatings", _i), ("refl_n_angles", _i), ("refl_n_energies", _i), ("_pad", _i),
        ("refl_angle_min", _d), ("refl_angle_max", _d), ("refl_energy_min", _d), ("refl_energy_max", _d),
        ("strongback_x", _dp), ("strongback_y", _dp), ("n_strongback", _i), ("_pad1", _i),
        ("window_x", _dp), ("window_y", _dp), ("n_window", _i), ("_pad2", _i),
        ("gas_abs_x", _dp), ("gas_abs_y", _dp), ("n_gas_abs", _i), ("_pad3", _i),
    ]


_TARGETS = {"f64": "libsart_oracle.so", "ld": "libsart_oracle_ld.so", "q": "libsart_oracle_q.so"}


def _native_target() -> str:
    """File name of the -O3 -march=native build for THIS host's CPU (bench.py's timed CPU baseline).  Built
    .so files travel with the repository snapshot to other machines, and a -march=native object must never run on a CPU
    it was not built for, so the name carries a hash of the CPU's model and flags."""
    import hashlib
    ident = ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith(("model name", "flags")):
                    ident += line
                if line.strip() == "" and ident:
                    break
    except OSError:
        ident = "unknown"
    return "libsart_oracle_native_%s.so" % hashlib.sha1(ident.encode()).hexdigest()[:10]


def build(variant: str = "f64") -> str:
    """Compiles the oracle with gcc if needed; returns the library path."""
    if variant == "native":
        target = _native_target()
        subprocess.run(["make", "-s", "-C", _DIR, "native", "NATIVE_TARGET=" + target], check=True)
        return os.path.join(_DIR, target)
    target = _TARGETS[variant]
    subprocess.run(["make", "-s", "-C", _DIR, target], check=True)
    return os.path.join(_DIR, target)


_libs = {}


def load(variant: str = "f64") -> C.CDLL:
    if variant not in _libs:
        path = os.path.join(_DIR, _native_target() if variant == "native" else _TARGETS[variant])
        if not os.path.exists(path):
            build(variant)
        lib = C.CDLL(path)
        vp = C.c_void_p
        lib.sart_oracle_uniforms.argtypes = [C.c_uint64, C.c_uint64, _dp]
        lib.sart_oracle_uniforms.restype = None
        lib.sart_oracle_trace_records.argtypes = [C.POINTER(Setup), C.POINTER(OracleTables), C.POINTER(TraceParams), vp, C.c_int]
        lib.sart_oracle_trace_records.restype = C.c_int
        lib.sart_oracle_trace_histogram.argtypes = [C.POINTER(Setup), C.POINTER(OracleTables), C.POINTER(TraceParams), _dp, C.c_int]
        lib.sart_oracle_trace_histogram.restype = C.c_int
        lib.sart_oracle_trace_records_nim_stream.argtypes = [C.POINTER(Setup), C.POINTER(OracleTables), C.POINTER(TraceParams), vp, C.c_int]
        lib.sart_oracle_trace_records_nim_stream.restype = C.c_int
        lib.sart_oracle_nim_rand_init.argtypes = [C.POINTER(C.c_uint64 * 2), C.c_int64, C.c_int]
        lib.sart_oracle_nim_rand_init.restype = None
        lib.sart_oracle_nim_rand_next.argtypes = [C.POINTER(C.c_uint64 * 2)]
        lib.sart_oracle_nim_rand_next.restype = C.c_uint64
        lib.sart_oracle_nim_rand_float.argtypes = [C.POINTER(C.c_uint64 * 2)]
        lib.sart_oracle_nim_rand_float.restype = _d
        lib.sart_oracle_conversion_prob.argtypes = [_d, _d, _d]
        lib.sart_oracle_conversion_prob.restype = _d
        lib.sart_oracle_eff_photon_mass2.argtypes = [_d] * 4
        lib.sart_oracle_eff_photon_mass2.restype = _d
        lib.sart_oracle_axion_conversion_prob2.argtypes = [_d] * 8
        lib.sart_oracle_axion_conversion_prob2.restype = _d
        lib.sart_oracle_intensity_suppression2.argtypes = [_d] * 6
        lib.sart_oracle_intensity_suppression2.restype = _d
        lib.sart_oracle_bilinear.argtypes = [_dp, _i, _i, _d, _d, _d, _d, _d, _d]
        lib.sart_oracle_bilinear.restype = _d
        lib.sart_oracle_linear1d.argtypes = [_dp, _dp, _i, _d]
        lib.sart_oracle_linear1d.restype = _d
        lib.sart_oracle_lower_bound.argtypes = [_dp, C.c_int64, _d]
        lib.sart_oracle_lower_bound.restype = C.c_int64
        lib.sart_oracle_almost_equal.argtypes = [_d, _d]
        lib.sart_oracle_almost_equal.restype = C.c_int
        lib.sart_oracle_find_pos.argtypes = [C.c_int, _dp, _dp, _d, _d, _d, _d, _d, _dp]
        lib.sart_oracle_find_pos.restype = None
        lib.sart_oracle_vector_after_mirror.argtypes = [_dp, _dp, _dp, _d, _d, _d, _d, C.c_int, _dp]
        lib.sart_oracle_vector_after_mirror.restype = None
        lib.sart_oracle_mirror_angle_deg.argtypes = [_dp, _dp, _dp, _d, _d, _d, _d, C.c_int]
        lib.sart_oracle_mirror_angle_deg.restype = _d
        lib.sart_oracle_trace_records_uniforms.argtypes = [C.POINTER(Setup), C.POINTER(OracleTables), C.c_uint32, _dp, C.c_int64, vp, C.c_int]
        lib.sart_oracle_trace_records_uniforms.restype = C.c_int
        lib.sart_oracle_mass_attenuation.argtypes = [_d]
        lib.sart_oracle_mass_attenuation.restype = _d
        lib.sart_oracle_length_telescope.argtypes = [C.POINTER(Setup)]
        lib.sart_oracle_length_telescope.restype = _d
        _libs[variant] = lib
    return _libs[variant]


class Oracle:
    """The oracle bound to one FullRaytraceSetup (same inputs the HIP path gets)."""

    def __init__(self, full, variant: str = "f64"):
        self.lib = load(variant)
        self.full = full
        t = OracleTables()
        self._keep = []

        def dp(a):
            a = np.ascontiguousarray(a, dtype=np.float64)
            self._keep.append(a)
            return a.ctypes.data_as(_dp)

        if hasattr(full, "require_solar_tables"):
            full.require_solar_tables()   # emission="agss09-device": the CDFs live on the GPU until fetch_solar_tables(tracer)
        t.flux_radius_cdf, t.diff_flux_cdfs, t.energies_kev = dp(full.fluxRadiusCDF), dp(full.diffFluxCDFs), dp(full.energies)
        t.n_radii, t.n_energies = full.diffFluxCDFs.shape
        r = full.reflectivity
        t.refl_data = dp(r.data)
        t.refl_n_coatings, t.refl_n_angles, t.refl_n_energies = r.data.shape
        t.refl_angle_min, t.refl_angle_max, t.refl_energy_min, t.refl_energy_max = r.angle_min, r.angle_max, r.energy_min, r.energy_max
        d = full.detector_tables
        t.strongback_x, t.strongback_y, t.n_strongback = dp(d.x_kev), dp(d.strongback), d.x_kev.size
        t.window_x, t.window_y, t.n_window = dp(d.x_kev), dp(d.window), d.x_kev.size
        t.gas_abs_x, t.gas_abs_y, t.n_gas_abs = dp(d.gas_x_kev), dp(d.gas_absorption), d.gas_x_kev.size
        self.tables = t

    def params(self, n_rays, seed=299792458, ray_id_offset=0, flags=None, image_n=256, accumulate=False) -> TraceParams:
        s = self.full.setup
        p = TraceParams()
        p.n_rays, p.seed, p.ray_id_offset = int(n_rays), int(seed), int(ray_id_offset)
        p.flags = self.full.flags if flags is None else flags
        p.image_nx = p.image_ny = image_n
        p.accumulate = 1 if accumulate else 0
        p.image_x_min, p.image_x_max = 0.0, s.chip_x_max
        p.image_y_min, p.image_y_max = 0.0, s.chip_y_max
        return p

    def trace_records(self, n_rays, seed=299792458, ray_id_offset=0, flags=None, n_threads=0, setup=None) -> np.ndarray:
        buf = np.zeros(n_rays, dtype=AXION_DTYPE)
        p = self.params(n_rays, seed, ray_id_offset, flags)
        s = setup if setup is not None else self.full.setup
        self.lib.sart_oracle_trace_records(C.byref(s), C.byref(self.tables), C.byref(p), buf.ctypes.data_as(C.c_void_p), n_threads)
        return buf

    def trace_records_uniforms(self, uniforms, flags=None, n_threads=0, setup=None) -> np.ndarray:
        """Records of the rays whose six uniforms are the rows of ``uniforms`` [n][6] (draw order of SURVEY App. B)."""
        u = np.ascontiguousarray(uniforms, dtype=np.float64)
        assert u.ndim == 2 and u.shape[1] == 6
        buf = np.zeros(u.shape[0], dtype=AXION_DTYPE)
        s = setup if setup is not None else self.full.setup
        fl = self.full.flags if flags is None else flags
        self.lib.sart_oracle_trace_records_uniforms(C.byref(s), C.byref(self.tables), fl, u.ctypes.data_as(_dp), u.shape[0],
                                                    buf.ctypes.data_as(C.c_void_p), n_threads)
        return buf

    def trace_records_nim_stream(self, n_rays, seed=299792458, ray_id_offset=0, flags=None, init_variant=1, setup=None) -> np.ndarray:
        """Records with the reference's own random stream (xoroshiro128+ seeded by randomize(seed), one thread, rays in order):
        what a single-threaded Nim run of the reference computes.  init_variant: 0 = Nim < 1.4, 1 = Nim >= 1.4."""
        buf = np.zeros(n_rays, dtype=AXION_DTYPE)
        p = self.params(n_rays, seed, ray_id_offset, flags)
        s = setup if setup is not None else self.full.setup
        self.lib.sart_oracle_trace_records_nim_stream(C.byref(s), C.byref(self.tables), C.byref(p), buf.ctypes.data_as(C.c_void_p),
                                                      init_variant)
        return buf

    def trace_spectra(self, n_rays, seed=299792458, ray_id_offset=0, flags=None, image_n=256, n_radial_bins=10_000,
                      radial_max=10.0, n_threads=0, setup=None):
        from solaraxionraytracing_amd.raytracer import split_spectra
        p = self.params(n_rays, seed, ray_id_offset, flags, image_n)
        p.spectra, p.n_radial_bins, p.radial_max = 1, n_radial_bins, radial_max
        n_e1 = self.full.energies.size + 1
        n_img = image_n * image_n
        acc = np.zeros(n_img + _lib.SART_ACC_COUNT + 2 * n_radial_bins + 3 * n_e1)
        s = setup if setup is not None else self.full.setup
        self.lib.sart_oracle_trace_histogram(C.byref(s), C.byref(self.tables), C.byref(p), acc.ctypes.data_as(_dp), n_threads)
        img = acc[:n_img].reshape(image_n, image_n).copy()
        summ = {k: acc[n_img + i] for k, i in _lib.ACC.items()}
        return img, summ, split_spectra(acc[n_img + _lib.SART_ACC_COUNT:], n_radial_bins, n_e1, radial_max)

    def trace_histogram(self, n_rays, seed=299792458, ray_id_offset=0, flags=None, image_n=256, n_threads=0, setup=None):
        p = self.params(n_rays, seed, ray_id_offset, flags, image_n)
        acc = np.zeros(image_n * image_n + _lib.SART_ACC_COUNT)
        s = setup if setup is not None else self.full.setup
        used = self.lib.sart_oracle_trace_histogram(C.byref(s), C.byref(self.tables), C.byref(p), acc.ctypes.data_as(_dp), n_threads)
        img = acc[:image_n * image_n].reshape(image_n, image_n).copy()
        summ = {k: acc[image_n * image_n + i] for k, i in _lib.ACC.items()}
        return img, summ, used


# ---- emission-table producer (oracle/sart_emission_oracle.c) --------------------------------------------------------
_em_lib = None


def load_emission() -> C.CDLL:
    global _em_lib
    if _em_lib is None:
        path = os.path.join(_DIR, "libsart_emission_oracle.so")
        if not os.path.exists(path):
            subprocess.run(["make", "-s", "-C", _DIR, "libsart_emission_oracle.so"], check=True)
        from solaraxionraytracing_amd._lib import EmissionParams, SolarZone
        lib = C.CDLL(path)
        lib.sart_emission_oracle_zones.argtypes = [_dp, _dp, _dp, _i, C.POINTER(SolarZone)]
        lib.sart_emission_oracle_zones.restype = None
        lib.sart_emission_oracle_fnew.argtypes = [_d, _d]
        lib.sart_emission_oracle_fnew.restype = _d
        lib.sart_emission_oracle_bfield.argtypes = [_d]
        lib.sart_emission_oracle_bfield.restype = _d
        lib.sart_emission_oracle_table.argtypes = [C.POINTER(SolarZone), _i, _dp, _i, _dp, C.POINTER(EmissionParams), _dp, _dp,
                                                   _i, _i, _i]
        lib.sart_emission_oracle_table.restype = C.c_int
        lib.sart_emission_oracle_flux_spectrum.argtypes = [_dp, _i, _dp, _i, _dp]
        lib.sart_emission_oracle_flux_spectrum.restype = None
        lib.sart_emission_oracle_number_densities.argtypes = [_dp, _dp, _i, _dp]
        lib.sart_emission_oracle_number_densities.restype = None
        lib.sart_emission_oracle_abs_coefs.argtypes = [C.POINTER(SolarZone), _i, _dp, _dp, _i, C.c_void_p, _dp]
        lib.sart_emission_oracle_abs_coefs.restype = C.c_int64
        _em_lib = lib
    return _em_lib


def emission_number_densities(rho, mass_fractions):
    """n_Z[n][29] by proton number (readOpacityFile.nim:655-679)."""
    lib = load_emission()
    rho = np.ascontiguousarray(rho, dtype=np.float64)
    frac = np.ascontiguousarray(mass_fractions, dtype=np.float64)
    out = np.empty((rho.size, 29))
    lib.sart_emission_oracle_number_densities(rho.ctypes.data_as(_dp), frac.ctypes.data_as(_dp), rho.size, out.ctypes.data_as(_dp))
    return out


def emission_abs_coefs(zones, n_z, energies, tables):
    """absCoefs[R][E] (readOpacityFile.nim:790-823); ``tables`` = ctypes pointer to a sart_opacity_tables_t.
    Returns (table, number of cells whose evaluation leaves a table = where the reference raises)."""
    lib = load_emission()
    energies = np.ascontiguousarray(energies, dtype=np.float64)
    n_z = np.ascontiguousarray(n_z, dtype=np.float64)
    out = np.empty((len(zones), energies.size))
    n_out = lib.sart_emission_oracle_abs_coefs(zones, len(zones), n_z.ctypes.data_as(_dp), energies.ctypes.data_as(_dp), energies.size,
                                               C.cast(tables, C.c_void_p), out.ctypes.data_as(_dp))
    return out, int(n_out)


def emission_zones(temp_K, rho, mass_fractions):
    from solaraxionraytracing_amd._lib import SolarZone
    lib = load_emission()
    temp_K = np.ascontiguousarray(temp_K, dtype=np.float64)
    rho = np.ascontiguousarray(rho, dtype=np.float64)
    frac = np.ascontiguousarray(mass_fractions, dtype=np.float64)
    zones = (SolarZone * temp_K.size)()
    lib.sart_emission_oracle_zones(temp_K.ctypes.data_as(_dp), rho.ctypes.data_as(_dp), frac.ctypes.data_as(_dp), temp_K.size, zones)
    return zones


def emission_table(zones, energies, params, abs_coefs=None, components=False, r_stride=1, e_stride=1, n_threads=0):
    """Oracle emission table; only every r_stride-th radius / e_stride-th energy is evaluated (others are NaN)."""
    lib = load_emission()
    energies = np.ascontiguousarray(energies, dtype=np.float64)
    n_r, n_e = len(zones), energies.size
    out = np.full((n_r, n_e), np.nan)
    comp = np.full((8, n_r, n_e), np.nan) if components else None
    if abs_coefs is not None:
        abs_coefs = np.ascontiguousarray(abs_coefs, dtype=np.float64)
    if n_threads <= 0:
        n_threads = len(os.sched_getaffinity(0))
    rc = lib.sart_emission_oracle_table(zones, n_r, energies.ctypes.data_as(_dp), n_e,
                                        abs_coefs.ctypes.data_as(_dp) if abs_coefs is not None else None, C.byref(params),
                                        out.ctypes.data_as(_dp), comp.ctypes.data_as(_dp) if components else None,
                                        r_stride, e_stride, n_threads)
    if rc != 0:
        raise RuntimeError("sart_emission_oracle_table failed")
    return (out, comp) if components else out


def emission_flux_spectrum(em_rates, energies):
    lib = load_emission()
    em = np.ascontiguousarray(em_rates, dtype=np.float64)
    energies = np.ascontiguousarray(energies, dtype=np.float64)
    out = np.empty(energies.size)
    lib.sart_emission_oracle_flux_spectrum(em.ctypes.data_as(_dp), em.shape[0], energies.ctypes.data_as(_dp), energies.size,
                                           out.ctypes.data_as(_dp))
    return out
