"""Host-side mirror of the reference's driver layer for the per-ray hot path, in the reference's
vocabulary (src/raytracer.nim): ``initFullSetup`` -> ``FullRaytraceSetup``; ``traceAxionWrapper``;
``calculateFluxFractions``; ``performAngularScan``.

This module is plumbing: it builds the inputs (through the C++ host library) and calls the C-ABI of
libsart.so.  All per-ray work happens in the HIP kernels; there is no CPU path here.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import _lib, tables
from ._lib import (AXION_DTYPE, Setup, Summary, TraceParams)  # noqa: F401


def newFullSetup(experiment: int, detector: int, stage: int, telescope: int, flags: int = 0,
                 magnet_cfg=None, source_cfg=None, install_cfg=None) -> Setup:
    """newExperimentSetup + newDetectorSetup + module constants (raytracer.nim:1411-1423, :1464-1496, :248-272)."""
    host = _lib.load_host()
    s = Setup()
    _lib.check(host.sart_host_new_full_setup(
        experiment, detector, stage, telescope, flags,
        C.byref(magnet_cfg) if magnet_cfg is not None else None,
        C.byref(source_cfg) if source_cfg is not None else None,
        C.byref(install_cfg) if install_cfg is not None else None, C.byref(s)), host=True)
    return s


@dataclass
class FullRaytraceSetup:
    """FullRaytraceSetup, raytracer.nim:232-242 (centerVecs are derived from ``setup`` by the callee)."""
    setup: Setup
    energies: np.ndarray            # [nE] keV
    fluxRadiusCDF: np.ndarray       # [nR]
    diffFluxCDFs: np.ndarray        # [nR][nE]
    reflectivity: tables.ReflectivityGrid
    detector_tables: tables.DetectorTables
    flags: int = 0
    outpath: str = "out"
    meta: dict = field(default_factory=dict)
    # emission="agss09-device": the sampling tables are made on the GPU by the context that uploads this setup (emission
    # kernel -> CDFs -> guides, nothing crosses PCIe); fluxRadiusCDF / diffFluxCDFs stay None until fetch_solar_tables()
    device_emission: dict | None = None

    def fetch_solar_tables(self, tracer: "RayTracer"):
        """Host copies of the CDFs a context built on the device (for the CPU oracle, plots, files)."""
        self.fluxRadiusCDF, self.diffFluxCDFs = tracer.solar_tables()
        return self

    def require_solar_tables(self):
        """Consumers that need the CDFs on the host (the CPU oracle, tools/make_golden*.py) call this: with
        emission="agss09-device" they exist on the GPU only until fetch_solar_tables(tracer) has copied them."""
        if self.fluxRadiusCDF is None or self.diffFluxCDFs is None:
            raise RuntimeError("this FullRaytraceSetup keeps its sampling tables on the device (emission=%r): call "
                               "full.fetch_solar_tables(tracer) with a RayTracer built from it first" % self.meta.get("emission"))
        return self


def initFullSetup(experiment: int = _lib.ES_BABYIAXO, detector: int = _lib.DK_INGRIDIAXO,
                  stage: int = _lib.SK_VACUUM, telescope: int = _lib.TK_XMM, flags: int = 0, *,
                  emission: str | np.ndarray = "primakoff", n_radii: int = tables.N_RADII,
                  n_energies: int = tables.N_ENERGIES, reflectivity: str | tables.ReflectivityGrid = "henke",
                  refl_n_angles: int = 1000, refl_n_energies: int = 1000, solar_model_csv: str | None = None,
                  magnet_cfg=None, source_cfg=None, install_cfg=None, opcd_path: str | None = None,
                  opcd_optional: bool = False) -> FullRaytraceSetup:
    """initFullSetup, raytracer.nim:2637-2753.  Defaults = config/config_default.toml:19-22
    (BabyIAXO / InGridIAXO / vacuum / XMM).  ``emission`` / ``reflectivity`` choose the synthetic stand-ins of
    tables.py when the reference's own input files are not available."""
    setup = newFullSetup(experiment, detector, stage, telescope, flags, magnet_cfg, source_cfg, install_cfg)
    dev_em = None
    if solar_model_csv is not None:
        radii, energies, em = tables.read_solar_model_csv(solar_model_csv)
        meta_em = "csv:" + solar_model_csv
    elif isinstance(emission, str) and emission == "legacy":   # E2: the reference's own emission_rates_Hz.txt / energies.txt (397 x 233)
        radii, energies, em = tables.legacy_emission_table()
        meta_em = "E2-legacy-emission_rates_Hz"
    else:
        radii, energies = tables.solar_grid(n_radii, n_energies)
        if isinstance(emission, np.ndarray):
            em, meta_em = emission, "user"
        elif emission == "primakoff":
            em, meta_em = tables.primakoff_emission_table(n_radii, n_energies), "E1-primakoff-agss09"
        elif emission == "agss09":   # all analytic terms of readOpacityFile.nim on the AGSS09 model, made on the GPU (BASELINE configs[4])
            from . import emission as _emission
            em, meta_em = _emission.agss09_emission_table(n_radii, n_energies)[2], "E0-agss09-all-terms-gpu"
        elif emission == "agss09-device":   # the same table, but it never leaves the GPU: RayTracer builds CDFs + guides on the device
            from . import emission as _emission
            zones = _emission.solar_zones(n_radii)
            dev_em = {"zones": zones, "params": _emission.default_params()}
            em, meta_em = None, "E0-agss09-all-terms-gpu-device-cdfs"
            if opcd_path is not None:   # with the absorption coefficients of the OPCD files (readOpacityFile.nim:731-745, :790-823)
                from . import opacity as _opacity
                try:
                    dev_em["opcd"] = _opacity.OpcdSet(opcd_path, zones)
                    dev_em["n_z"] = _opacity.number_densities(n_radii=n_radii)
                    dev_em["opcd_optional"] = bool(opcd_optional)
                    meta_em = "E0-agss09-all-terms-opcd-gpu-device-cdfs"
                except (_lib.SartError, OSError, ValueError) as e:
                    # opcd_optional (a config.toml that merely HAS an OPCD directory): an incomplete or unreadable set of files
                    # must not take the whole setup down - the table is made without the OPCD absorption term and says so
                    if not opcd_optional:
                        raise
                    dev_em.pop("opcd", None)
                    dev_em["notes"] = ["OPCD files in %s could not be loaded (%s): emission table without the OPCD absorption term" % (opcd_path, e)]
        elif emission == "flat":
            em, meta_em = tables.flat_emission_table(n_radii, n_energies), "E3-flat"
        else:
            raise ValueError("unknown emission table %r" % (emission,))
    rcdf, ecdf = tables.build_cdfs(em, radii, energies) if em is not None else (None, None)
    if isinstance(reflectivity, tables.ReflectivityGrid):
        refl, meta_r = reflectivity, "user"
    else:
        multi = setup.reflectivity_kind == _lib.RK_MULTI_COATING
        if reflectivity == "henke":
            refl = (tables.llnl_reflectivity_grids if multi else tables.gold_reflectivity_grid)(refl_n_angles, refl_n_energies)
            meta_r = "L1-henke-x4" if multi else "G1-henke-gold"
        elif reflectivity == "gold":   # single gold table on every shell (BASELINE config 2)
            refl, meta_r = tables.gold_reflectivity_grid(refl_n_angles, refl_n_energies), "G1-henke-gold"
            if multi:
                setup.reflectivity_kind = _lib.RK_SINGLE_COATING
                setup.n_coatings = 1
                setup.coating_layers[0] = setup.n_shells
        elif reflectivity == "analytic":
            refl = tables.analytic_reflectivity_grid(4 if multi else 1, refl_n_angles, refl_n_energies)
            meta_r = "G2-analytic"
        else:
            raise ValueError("unknown reflectivity %r" % (reflectivity,))
    det = tables.detector_tables()
    meta = {"emission": meta_em, "reflectivity": meta_r}
    if dev_em is not None and dev_em.get("notes"):
        meta["notes"] = list(dev_em["notes"])
    return FullRaytraceSetup(setup, np.ascontiguousarray(energies), rcdf, ecdf, refl, det, flags, meta=meta, device_emission=dev_em)


class RayTracer:
    """One libsart context on one GPU holding the captures of ``traceAxionWrapper``
    (raytracer.nim:2223-2232)."""

    def __init__(self, full: FullRaytraceSetup, device: int = 0):
        self.lib = _lib.load_sart()
        self.full = full
        h = C.c_void_p()
        _lib.check(self.lib.sart_create(device, C.byref(h)))
        self.handle = h
        try:
            self._upload(full)
        except Exception:
            self.close()
            raise

    def _upload(self, full: FullRaytraceSetup):
        lib, h = self.lib, self.handle
        _lib.check(lib.sart_set_setup(h, C.byref(full.setup)))
        if full.device_emission is not None:
            # BASELINE configs[4]'s front end without a host round trip: emission kernel -> CDFs -> guide tables on the device
            de = full.device_emission
            done = False
            if de.get("opcd") is not None:
                try:
                    _lib.check(lib.sart_emission_to_solar_tables_opcd(h, de["zones"], len(de["zones"]), _lib.as_dp(de["n_z"]),
                                                                      _lib.as_dp(full.energies), full.energies.size, de["opcd"].tables,
                                                                      C.byref(de["params"])))
                    done = True
                except _lib.SartError as e:
                    # e.g. a (radius, energy) cell outside a table's abscissae (sart_opacity.hip): fatal unless the OPCD term
                    # was only an opportunistic upgrade (config.py), in which case the table is made without it - and says so
                    if not de.get("opcd_optional"):
                        raise
                    full.meta.setdefault("notes", []).append("OPCD absorption coefficients failed on the device (%s): emission table "
                                                             "without the OPCD absorption term" % e)
                    full.meta["emission"] = "E0-agss09-all-terms-gpu-device-cdfs"
            if not done:
                _lib.check(lib.sart_emission_to_solar_tables(h, de["zones"], len(de["zones"]), _lib.as_dp(full.energies),
                                                             full.energies.size, None, C.byref(de["params"])))
        else:
            n_r, n_e = full.diffFluxCDFs.shape
            _lib.check(lib.sart_set_solar_tables(h, _lib.as_dp(full.fluxRadiusCDF), _lib.as_dp(full.diffFluxCDFs),
                                                 _lib.as_dp(full.energies), n_r, n_e))
        r = full.reflectivity
        n_c, n_a, n_er = r.data.shape
        _lib.check(lib.sart_set_reflectivity(h, n_c, n_a, n_er, r.angle_min, r.angle_max, r.energy_min, r.energy_max,
                                             _lib.as_dp(r.data)))
        d = full.detector_tables
        _lib.check(lib.sart_set_detector_tables(h, _lib.as_dp(d.x_kev), _lib.as_dp(d.strongback), d.x_kev.size,
                                                _lib.as_dp(d.x_kev), _lib.as_dp(d.window), d.x_kev.size,
                                                _lib.as_dp(d.gas_x_kev), _lib.as_dp(d.gas_absorption), d.gas_x_kev.size))

    # -- lifecycle ----------------------------------------------------------------------------
    def close(self):
        if getattr(self, "handle", None):
            self.lib.sart_destroy(self.handle)
            self.handle = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- parameters ---------------------------------------------------------------------------
    def trace_params(self, n_rays: int, seed: int = 299792458, ray_id_offset: int = 0, flags: int | None = None,
                     image_n: int = 256, accumulate: bool = False) -> TraceParams:
        s = self.full.setup
        p = TraceParams()
        p.n_rays, p.seed, p.ray_id_offset = int(n_rays), int(seed), int(ray_id_offset)
        p.flags = self.full.flags if flags is None else flags
        p.image_nx = p.image_ny = image_n
        p.accumulate = 1 if accumulate else 0
        p.image_x_min, p.image_x_max = 0.0, s.chip_x_max      # beginX/endX (raytracer.nim:2622-2625)
        p.image_y_min, p.image_y_max = 0.0, s.chip_y_max
        return p

    def set_telescope_angles(self, turned_x_deg: float = float("nan"), turned_y_deg: float = float("nan")):
        _lib.check(self.lib.sart_set_telescope_angles(self.handle, turned_x_deg, turned_y_deg))

    def set_axion_mass(self, m_axion_ev: float):
        _lib.check(self.lib.sart_set_axion_mass(self.handle, m_axion_ev))

    def set_accumulation_mode(self, mode: int | str = _lib.ACCUM_F64, headroom_bits: int = 0):
        """SART_ACCUM_F64 (default: f64 atomics, as prepareHeatmap adds on the CPU) or SART_ACCUM_FIXED64 ("fixed64":
        integer accumulation, results bitwise independent of GPU count, replica placement and launch splitting;
        include/sart.h "accumulation mode").  The blocking calls keep returning doubles; trace_histogram_device then fills
        raw int64 accumulators (reduce them as int64, convert with finalize_accumulator_device)."""
        if isinstance(mode, str):
            mode = {"f64": _lib.ACCUM_F64, "fixed64": _lib.ACCUM_FIXED64}[mode]
        _lib.check(self.lib.sart_set_accumulation_mode(self.handle, int(mode), int(headroom_bits)))

    def accumulation_mode(self) -> int:
        m = C.c_int()
        _lib.check(self.lib.sart_get_accumulation_mode(self.handle, C.byref(m)))
        return m.value

    def fixed_quanta(self) -> dict:
        """The quanta the raw FIXED64 accumulators of this context count in (frozen by the first launch)."""
        q = _lib.FixedQuanta()
        _lib.check(self.lib.sart_get_fixed_quanta(self.handle, C.byref(q)))
        return {"weight": q.weight, "weight_sq": q.weight_sq, "position": q.position, "reflect": q.reflect}

    def finalize_accumulator_device(self, params: TraceParams, acc_fixed_ptr: int, out_f64_ptr: int | None = None):
        """Raw FIXED64 accumulator (device) -> f64 accumulator layout (device; in place by default).  Asynchronous."""
        _lib.check(self.lib.sart_finalize_accumulator_device(self.handle, C.byref(params), C.c_void_p(acc_fixed_ptr),
                                                             C.c_void_p(out_f64_ptr if out_f64_ptr is not None else acc_fixed_ptr)))

    def rollover_accumulator_device(self, params: TraceParams, acc_fixed_ptr: int, hi_limbs_ptr: int):
        """Long FIXED64 accumulations: moves the bits of every slot above 2^40 into the slot's limb in ``hi_limbs`` (device int64
        array of the accumulator's length, zeroed with it).  Asynchronous."""
        _lib.check(self.lib.sart_rollover_accumulator_device(self.handle, C.byref(params), C.c_void_p(acc_fixed_ptr), C.c_void_p(hi_limbs_ptr)))

    def finalize_accumulator_limbs_device(self, params: TraceParams, acc_fixed_ptr: int, hi_limbs_ptr: int | None, out_f64_ptr: int | None = None):
        """finalize_accumulator_device for an accumulator with roll-over limbs."""
        _lib.check(self.lib.sart_finalize_accumulator_limbs_device(self.handle, C.byref(params), C.c_void_p(acc_fixed_ptr),
                                                                   C.c_void_p(hi_limbs_ptr) if hi_limbs_ptr else None,
                                                                   C.c_void_p(out_f64_ptr if out_f64_ptr is not None else acc_fixed_ptr)))

    def set_solar_tables_device(self, em_rates_device_ptr: int, radii: np.ndarray, energies: np.ndarray):
        """sart_set_solar_tables_device: CDFs + guide tables built on the device from a device-resident emission table
        [n_radii][n_energies] (``torch_tensor.data_ptr()``)."""
        radii = np.ascontiguousarray(radii, dtype=np.float64)
        energies = np.ascontiguousarray(energies, dtype=np.float64)
        _lib.check(self.lib.sart_set_solar_tables_device(self.handle, C.c_void_p(em_rates_device_ptr), _lib.as_dp(radii),
                                                         _lib.as_dp(energies), radii.size, energies.size))
        self.full.energies = energies
        self._n_radii_set = radii.size

    def solar_tables(self, guides: bool = False):
        """Host copies of (fluxRadiusCDF, diffFluxCDFs) as the context holds them; with ``guides`` also the library's
        guide tables (radius guide [_lib.RADIUS_GUIDE_ENTRIES], energy guide [n_radii][_lib.ENERGY_GUIDE_ENTRIES], u16)."""
        n_e = self.full.energies.size
        n_r = self._n_radii()
        rcdf, ecdf = np.empty(n_r), np.empty((n_r, n_e))
        rg = np.empty(_lib.RADIUS_GUIDE_ENTRIES, dtype=np.uint16) if guides else None
        eg = np.empty((n_r, _lib.ENERGY_GUIDE_ENTRIES), dtype=np.uint16) if guides else None
        _lib.check(self.lib.sart_get_solar_tables(self.handle, _lib.as_dp(rcdf), _lib.as_dp(ecdf),
                                                  rg.ctypes.data_as(C.c_void_p) if guides else None,
                                                  eg.ctypes.data_as(C.c_void_p) if guides else None))
        return (rcdf, ecdf, rg, eg) if guides else (rcdf, ecdf)

    def _n_radii(self) -> int:
        if getattr(self, "_n_radii_set", None):
            return self._n_radii_set
        if self.full.device_emission is not None:
            return len(self.full.device_emission["zones"])
        if self.full.diffFluxCDFs is not None:
            return self.full.diffFluxCDFs.shape[0]
        raise RuntimeError("number of radii unknown")

    def set_stream(self, hip_stream: int | None):
        _lib.check(self.lib.sart_set_stream(self.handle, C.c_void_p(hip_stream) if hip_stream else None))

    def synchronize(self):
        _lib.check(self.lib.sart_synchronize(self.handle))

    # -- the hot path -------------------------------------------------------------------------
    def traceAxionWrapper(self, bufLen: int, seed: int = 299792458, ray_id_offset: int = 0,
                          flags: int | None = None) -> np.ndarray:
        """traceAxionWrapper (raytracer.nim:2223-2244): ``bufLen`` Axion records (numpy structured array with
        the reference's field names)."""
        buf = np.zeros(bufLen, dtype=AXION_DTYPE)
        p = self.trace_params(bufLen, seed, ray_id_offset, flags)
        _lib.check(self.lib.sart_trace_records(self.handle, C.byref(p), buf.ctypes.data_as(C.c_void_p)))
        return buf

    def traceAxionWrapperPassed(self, bufLen: int, seed: int = 299792458, ray_id_offset: int = 0, flags: int | None = None,
                                capacity: int | None = None, out: np.ndarray | None = None):
        """The rays of traceAxionWrapper(bufLen, ...) whose ``passed`` is set, in ray order, and the counts generateResultPlots
        echoes (raytracer.nim:2252-2257): (records, {"n_rays", "n_passed", "n_passed_till_window", "n_hit_nickel"}).  Only the
        passed records cross PCIe (sart_trace_records_passed).  ``capacity`` = room of the buffer in records (default bufLen:
        always enough); ``out`` = a buffer of the caller's.  len(records) = min(n_passed, capacity)."""
        if out is None:
            out = np.empty(bufLen if capacity is None else capacity, dtype=AXION_DTYPE)
        assert out.dtype == AXION_DTYPE and out.flags.c_contiguous
        capacity = len(out) if capacity is None else min(int(capacity), len(out))
        p = self.trace_params(bufLen, seed, ray_id_offset, flags)
        cnt = _lib.RecordCounts()
        _lib.check(self.lib.sart_trace_records_passed(self.handle, C.byref(p), out.ctypes.data_as(C.c_void_p), capacity, C.byref(cnt)))
        counts = {k: int(getattr(cnt, k)) for k, _ in _lib.RecordCounts._fields_}
        return out[:min(counts["n_passed"], capacity)], counts

    def trace_records_passed_device(self, params: TraceParams, out_ptr: int, capacity: int, counts_ptr: int):
        """sart_trace_records_passed_device: device pointers (records, four uint64 counts); enqueues on the stream."""
        _lib.check(self.lib.sart_trace_records_passed_device(self.handle, C.byref(params), C.c_void_p(out_ptr), int(capacity), C.c_void_p(counts_ptr)))

    def trace_records_uniforms(self, uniforms: np.ndarray, flags: int | None = None) -> np.ndarray:
        """Test entry (sart_internal_trace_records_uniforms, not part of include/sart.h): the records of the rays whose six
        uniforms are the rows of ``uniforms`` [n][6] (draw order of SURVEY App. B) instead of draws from the Philox stream."""
        u = np.ascontiguousarray(uniforms, dtype=np.float64)
        assert u.ndim == 2 and u.shape[1] == 6
        buf = np.zeros(u.shape[0], dtype=AXION_DTYPE)
        p = self.trace_params(u.shape[0], flags=flags)
        fn = self.lib.sart_internal_trace_records_uniforms
        fn.restype, fn.argtypes = C.c_int, [C.c_void_p, C.POINTER(TraceParams), C.c_void_p, C.c_void_p]
        _lib.check(fn(self.handle, C.byref(p), u.ctypes.data_as(C.c_void_p), buf.ctypes.data_as(C.c_void_p)))
        return buf

    def trace_histogram(self, n_rays: int, seed: int = 299792458, ray_id_offset: int = 0, flags: int | None = None,
                        image_n: int = 256, accumulate: bool = False):
        """Fused trace + prepareHeatmap(256,256,norm=1) + flux sum + counters.  Returns (image[ny][nx], summary
        dict keyed like include/sart.h SART_ACC_*)."""
        p = self.trace_params(n_rays, seed, ray_id_offset, flags, image_n, accumulate)
        img = np.empty((image_n, image_n))
        summ = Summary()
        _lib.check(self.lib.sart_trace_histogram(self.handle, C.byref(p), _lib.as_dp(img), C.byref(summ)))
        return img, {k: summ.v[i] for k, i in _lib.ACC.items()}

    def trace_image(self, n_rays: int, nx: int, ny: int, x_range=None, y_range=None, seed: int = 299792458,
                    ray_id_offset: int = 0, flags: int | None = None):
        """prepareHeatmap with any binning / window (raytracer.nim:818-842): e.g. the 3000 x 3000 maps of
        generateResultPlots (:2626, :2630).  Returns (image[ny][nx], summary)."""
        p = self.trace_params(n_rays, seed, ray_id_offset, flags, 1, False)
        p.image_nx, p.image_ny = int(nx), int(ny)
        if x_range is not None:
            p.image_x_min, p.image_x_max = float(x_range[0]), float(x_range[1])
        if y_range is not None:
            p.image_y_min, p.image_y_max = float(y_range[0]), float(y_range[1])
        img = np.empty((ny, nx))
        summ = Summary()
        _lib.check(self.lib.sart_trace_histogram(self.handle, C.byref(p), _lib.as_dp(img), C.byref(summ)))
        return img, {k: summ.v[i] for k, i in _lib.ACC.items()}

    def y_slice_histogram(self, n_rays: int, half_width: float = 0.05, bin_width: float = 0.001, **kw):
        """`y_{year}.pdf` of generateResultPlots (raytracer.nim:2551-2559): weighted histogram (bin 0.001 mm) of the
        y-positions of the rays with |x - ChipCenterX| < 0.05 mm — one image column.  Returns (bin edges, flux per bin)."""
        s = self.full.setup
        cx = 0.5 * s.chip_x_max
        ny = int(round(s.chip_y_max / bin_width))
        img, _ = self.trace_image(n_rays, 1, ny, x_range=(cx - half_width, cx + half_width), y_range=(0.0, s.chip_y_max), **kw)
        return np.linspace(0.0, s.chip_y_max, ny + 1), img[:, 0].copy()

    def trace_spectra(self, n_rays: int, seed: int = 299792458, ray_id_offset: int = 0, flags: int | None = None,
                      image_n: int = 256, n_radial_bins: int = 10_000, radial_max: float = 10.0, accumulate: bool = False):
        """trace_histogram plus the post-processing histograms of generateResultPlots accumulated on the device:
        radial distribution (bin 0.001 mm by default, raytracer.nim:2386) and per-energy-index spectra.  Returns
        (image, summary, spectra dict)."""
        p = self.trace_params(n_rays, seed, ray_id_offset, flags, image_n, accumulate)
        p.spectra, p.n_radial_bins, p.radial_max = 1, n_radial_bins, radial_max
        n_e1 = self.full.energies.size + 1
        img = np.empty((image_n, image_n))
        summ = Summary()
        spec = np.empty(2 * n_radial_bins + 3 * n_e1)
        _lib.check(self.lib.sart_trace_histogram_spectra(self.handle, C.byref(p), _lib.as_dp(img), C.byref(summ), _lib.as_dp(spec)))
        return img, {k: summ.v[i] for k, i in _lib.ACC.items()}, split_spectra(spec, n_radial_bins, n_e1, radial_max)

    def trace_histogram_device(self, params: TraceParams, accumulator_ptr: int):
        """Asynchronous form: adds into a device accumulator (e.g. ``torch_tensor.data_ptr()``)."""
        _lib.check(self.lib.sart_trace_histogram_device(self.handle, C.byref(params), C.c_void_p(accumulator_ptr)))

    # -- fused axion-mass scan (gas stage; include/sart.h "fused axion-mass scan") -------------
    def trace_mass_scan(self, masses_ev, n_rays: int, seed: int = 299792458, ray_id_offset: int = 0, flags: int | None = None):
        """Every ray of [ray_id_offset, ray_id_offset + n_rays) traced ONCE and weighed for every axion mass.  Returns
        (per-mass dict of arrays SUM_WEIGHTS / SUM_WEIGHTS_SQ / N_PASSED, dict of the mass-independent counters)."""
        masses = np.ascontiguousarray(masses_ev, dtype=np.float64)
        p = self.trace_params(n_rays, seed, ray_id_offset, flags)
        out = np.empty(mass_scan_len(masses.size))
        _lib.check(self.lib.sart_trace_mass_scan(self.handle, C.byref(p), _lib.as_dp(masses), masses.size, _lib.as_dp(out)))
        return split_mass_scan(out, masses.size)

    def trace_mass_scan_device(self, params: TraceParams, masses_ev, scan_acc_ptr: int):
        """Asynchronous form: adds into a device scan accumulator of mass_scan_len(n) 8-byte slots (raw int64 in fixed64 mode)."""
        masses = np.ascontiguousarray(masses_ev, dtype=np.float64)
        _lib.check(self.lib.sart_trace_mass_scan_device(self.handle, C.byref(params), _lib.as_dp(masses), masses.size,
                                                        C.c_void_p(scan_acc_ptr)))

    def finalize_mass_scan_device(self, params: TraceParams, masses_ev, raw_ptr: int, out_ptr: int | None = None):
        """Raw FIXED64 scan accumulator (device) -> doubles (device; in place by default).  Asynchronous; what the conversion
        finds wrong (unresolved weights, wrapped slots) is raised by the next ``synchronize()``."""
        masses = np.ascontiguousarray(masses_ev, dtype=np.float64)
        _lib.check(self.lib.sart_finalize_mass_scan_device(self.handle, C.byref(params), _lib.as_dp(masses), masses.size,
                                                           C.c_void_p(raw_ptr), C.c_void_p(out_ptr if out_ptr is not None else raw_ptr)))

    # -- fused angular scan (include/sart.h "fused angular scan") -------------------------------
    def trace_angular_scan(self, turned_y_deg, n_rays: int, seed: int = 299792458, ray_id_offset: int = 0, flags: int | None = None):
        """Every ray of [ray_id_offset, ray_id_offset + n_rays) sampled and taken through bore and pipes ONCE and turned through
        every telescope angle.  Returns (per-angle dict of arrays SUM_WEIGHTS / SUM_WEIGHTS_SQ / N_PASSED / N_SHELL_SELECTED /
        N_HIT_NICKEL / N_PASSED_TILL_WINDOW, dict of the angle-independent counters)."""
        angles = np.ascontiguousarray(turned_y_deg, dtype=np.float64)
        p = self.trace_params(n_rays, seed, ray_id_offset, flags)
        out = np.empty(angular_scan_len(angles.size))
        _lib.check(self.lib.sart_trace_angular_scan(self.handle, C.byref(p), _lib.as_dp(angles), angles.size, _lib.as_dp(out)))
        return split_angular_scan(out, angles.size)

    def trace_angular_scan_device(self, params: TraceParams, turned_y_deg, scan_acc_ptr: int):
        """Asynchronous form: adds into a device scan accumulator of angular_scan_len(n) 8-byte slots (raw int64 in fixed64 mode)."""
        angles = np.ascontiguousarray(turned_y_deg, dtype=np.float64)
        _lib.check(self.lib.sart_trace_angular_scan_device(self.handle, C.byref(params), _lib.as_dp(angles), angles.size,
                                                           C.c_void_p(scan_acc_ptr)))

    def finalize_angular_scan_device(self, params: TraceParams, n_angles: int, raw_ptr: int, out_ptr: int | None = None):
        """Raw FIXED64 scan accumulator (device) -> doubles (device; in place by default).  Asynchronous."""
        _lib.check(self.lib.sart_finalize_angular_scan_device(self.handle, C.byref(params), int(n_angles), C.c_void_p(raw_ptr),
                                                              C.c_void_p(out_ptr if out_ptr is not None else raw_ptr)))

    def trace_flux(self, n_rays: int, seed: int = 299792458, ray_id_offset: int = 0, flags: int | None = None):
        """Flux-only launch (image_nx = image_ny = 0, include/sart.h): the summary of trace_histogram without an image."""
        p = self.trace_params(n_rays, seed, ray_id_offset, flags, 0, False)
        summ = Summary()
        _lib.check(self.lib.sart_trace_histogram(self.handle, C.byref(p), None, C.byref(summ)))
        return {k: summ.v[i] for k, i in _lib.ACC.items()}

    # -- measurement --------------------------------------------------------------------------
    def enable_kernel_timing(self, enable: bool = True):
        _lib.check(self.lib.sart_enable_kernel_timing(self.handle, 1 if enable else 0))

    def kernel_timing(self):
        ms, n = C.c_double(), C.c_int64()
        _lib.check(self.lib.sart_get_kernel_timing(self.handle, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def device_info(self):
        ncu, ws = C.c_int32(), C.c_int32()
        name = C.create_string_buffer(256)
        _lib.check(self.lib.sart_device_info(self.handle, C.byref(ncu), C.byref(ws), name, 256))
        return {"n_cu": ncu.value, "wave_size": ws.value, "name": name.value.decode()}


def split_spectra(spec: np.ndarray, n_radial_bins: int, n_e1: int, radial_max: float) -> dict:
    """Names the pieces of the spectra block of the accumulator (include/sart.h: sart_accumulator_len_spectra)."""
    o = 0
    out = {}
    for name, n in (("radial_counts", n_radial_bins), ("radial_weights", n_radial_bins), ("energy_counts", n_e1),
                    ("energy_weights", n_e1), ("energy_reflect", n_e1)):
        out[name] = spec[o:o + n].copy()
        o += n
    out["radial_max"] = radial_max
    return out


def containment_radii(spectra: dict):
    """rSigma1, rSigma2 (unweighted) and rSigma1W, rSigma2W (weighted) of generateResultPlots (raytracer.nim:2459-2527)."""
    host = _lib.load_host()
    r = [C.c_double() for _ in range(4)]
    rc, rw = np.ascontiguousarray(spectra["radial_counts"]), np.ascontiguousarray(spectra["radial_weights"])
    _lib.check(host.sart_host_containment_radii(_lib.as_dp(rc), _lib.as_dp(rw), rc.size, spectra["radial_max"],
                                                *[C.byref(x) for x in r]), host=True)
    return tuple(x.value for x in r)


def write_image_csv(path: str, image: np.ndarray, chip_max: float, r_sigma1: float, r_sigma2: float) -> float:
    """plotHeatmap's `axion_image_{year}{suffix}.csv` (raytracer.nim:887-921).  Returns the total flux."""
    host = _lib.load_host()
    img = np.ascontiguousarray(image, dtype=np.float64)
    flux = C.c_double()
    _lib.check(host.sart_host_write_image_csv(path.encode(), _lib.as_dp(img), img.shape[0], chip_max, r_sigma1, r_sigma2,
                                              C.byref(flux)), host=True)
    return flux.value


def accumulator_len(image_n: int = 256) -> int:
    return image_n * image_n + _lib.SART_ACC_COUNT


def mass_scan_len(n_masses: int) -> int:
    """sart_mass_scan_len: 8-byte slots of a scan accumulator."""
    return (int(n_masses) + 1) * _lib.SCAN_ROW


def split_mass_scan(acc: np.ndarray, n_masses: int):
    """(per-mass dict of arrays, shared-counter dict) from a finalized scan accumulator."""
    rows = np.asarray(acc, dtype=np.float64).reshape(n_masses + 1, _lib.SCAN_ROW)
    per_mass = {k: rows[:n_masses, i].copy() for k, i in _lib.SCAN.items()}
    shared = {k: float(rows[n_masses, i]) for k, i in _lib.SCAN_SHARED.items()}
    return per_mass, shared


def angular_scan_len(n_angles: int) -> int:
    """sart_angular_scan_len: 8-byte slots of an angular-scan accumulator."""
    return (int(n_angles) + 1) * _lib.ASCAN_ROW


def split_angular_scan(acc: np.ndarray, n_angles: int):
    """(per-angle dict of arrays, shared-counter dict) from a finalized angular-scan accumulator."""
    rows = np.asarray(acc, dtype=np.float64).reshape(n_angles + 1, _lib.ASCAN_ROW)
    per_angle = {k: rows[:n_angles, i].copy() for k, i in _lib.ASCAN.items()}
    shared = {k: float(rows[n_angles, i]) for k, i in _lib.ASCAN_SHARED.items()}
    return per_angle, shared


def calculateFluxFractions(tracer: RayTracer, n_rays: int = 1_000_000, seed: int = 299792458,
                           ray_id_offset: int = 0):
    """calculateFluxFractions (raytracer.nim:2755-2776) in histogram form: NumberOfPointsSun rays ->
    256x256 focal-plane image (heatmaptable2, :2629) + counters (:2252-2257) + total flux (:885)."""
    return tracer.trace_histogram(n_rays, seed, ray_id_offset)


def performAxionMassScan(tracer: RayTracer, masses_ev, n_rays_per_mass: int = 1_000_000, seed: int = 299792458,
                         flags: int | None = None, ray_id_offset: int = 0, errors: bool = False):
    """m_a scan (BASELINE configs[4]) through the C++ host driver: flux (sum of weights) per axion mass.  Gas stage: the
    fused scan kernel - every ray of [ray_id_offset, ray_id_offset + n_rays_per_mass) is traced once and weighed for every
    mass (common random numbers).  ``errors``: also return sqrt(sum of squared weights) and the passed-ray counts."""
    host = _lib.load_host()
    masses = np.ascontiguousarray(masses_ev, dtype=np.float64)
    fluxes, sq, n_pass = np.empty_like(masses), np.empty_like(masses), np.empty_like(masses)
    fl = tracer.full.flags if flags is None else flags
    _lib.check(host.sart_host_axion_mass_scan(tracer.handle, _lib.as_dp(masses), masses.size, n_rays_per_mass, seed,
                                              ray_id_offset, fl, _lib.as_dp(fluxes), _lib.as_dp(sq), _lib.as_dp(n_pass)), host=True)
    return (fluxes, np.sqrt(sq), n_pass) if errors else fluxes


def performAxionMassScanHostLoop(tracer: RayTracer, masses_ev, n_rays_per_mass: int = 1_000_000, seed: int = 299792458,
                                 flags: int | None = None, ray_id_offset: int = 0, same_rays: bool = True):
    """The reference-shaped scan (a host loop: set the mass, re-trace, sum) - what the fused kernel replaces; kept as the
    comparison of the parity tests and of bench.py.  ``same_rays``: every mass on the same ray ids (what the fused scan
    computes), else mass i on its own block of ids = the C++ host driver sart_host_perform_axion_mass_scan."""
    masses = np.ascontiguousarray(masses_ev, dtype=np.float64)
    out = np.empty_like(masses)
    if not same_rays:
        host = _lib.load_host()
        fl = tracer.full.flags if flags is None else flags
        _lib.check(host.sart_host_perform_axion_mass_scan(tracer.handle, _lib.as_dp(masses), masses.size, n_rays_per_mass, seed,
                                                          ray_id_offset, fl, _lib.as_dp(out)), host=True)
        return out
    m0 = tracer.full.setup.m_axion
    try:
        for i, m in enumerate(masses):
            tracer.set_axion_mass(float(m))
            off = ray_id_offset + (0 if same_rays else i * n_rays_per_mass)
            out[i] = tracer.trace_histogram(n_rays_per_mass, seed, off, flags)[1]["SUM_WEIGHTS"]
    finally:
        tracer.set_axion_mass(m0)
    return out


def performAngularScan(tracer: RayTracer, angularScanMin: float, angularScanMax: float, numAngularScanPoints: int = 50,
                       n_rays_per_angle: int = 1_000_000, seed: int = 299792458, flags: int | None = None,
                       angles=None, ray_id_offset: int = 0, fused: bool = False, errors: bool = False):
    """performAngularScan (raytracer.nim:2778-2802) through the C++ host driver: returns (angles, fluxes,
    relative fluxes).  ``angles`` overrides the linspace (used when angle bins are sharded over GPUs).
    ``fused = False``: the reference's shape - a host loop, angle i on its own block of fresh ray ids (flux-only launches).
    ``fused = True``: the fused scan kernel - every ray of [ray_id_offset, ray_id_offset + n_rays_per_angle) is sampled once
    and turned through every angle (common random numbers); with ``errors`` also sqrt(sum of squared weights) and the
    passed-ray counts."""
    host = _lib.load_host()
    angles = np.linspace(angularScanMin, angularScanMax, numAngularScanPoints) if angles is None else \
        np.ascontiguousarray(angles, dtype=np.float64)
    fluxes = np.empty_like(angles)
    rel = np.empty_like(angles)
    fl = tracer.full.flags if flags is None else flags
    if fused:
        sq, n_pass = np.empty_like(angles), np.empty_like(angles)
        _lib.check(host.sart_host_angular_scan(tracer.handle, _lib.as_dp(angles), angles.size, n_rays_per_angle, seed, ray_id_offset,
                                               fl, _lib.as_dp(fluxes), _lib.as_dp(rel), _lib.as_dp(sq), _lib.as_dp(n_pass)), host=True)
        return (angles, fluxes, rel, np.sqrt(sq), n_pass) if errors else (angles, fluxes, rel)
    _lib.check(host.sart_host_perform_angular_scan(tracer.handle, _lib.as_dp(angles), angles.size, n_rays_per_angle,
                                                   seed, ray_id_offset, fl, _lib.as_dp(fluxes), _lib.as_dp(rel)), host=True)
    return angles, fluxes, rel
