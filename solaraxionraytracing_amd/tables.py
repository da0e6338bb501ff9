"""Input tables of the hot path: loaders for the reference's file formats and the documented
synthetic stand-ins for the inputs the reference does not ship.

What the reference reads and what is used here when it is absent:

* ``solar_model_dataframe.csv`` (Radius, Energy [keV], emRates; raytracer.nim:2647-2668) — absent
  (needs the un-shipped OPCD opacity data).  Stand-in **E1**: an OPCD-free Primakoff emission table
  evaluated on the AGSS09 solar model (formula of readOpacityFile.nim:394-413) on the reference's grid
  (1968 radii 0.0015..0.985 step 0.0005, 1500 energies linspace(1e-3, 15) keV).
* ``gold_0.25microns_reflectivities.h5`` / ``llnl_layer_reflectivities.h5`` (raytracer.nim:1160-1231) —
  missing large blobs.  Stand-in **G1**: the 71 Henke gold scans of resources/reflectivity.zip resampled
  onto the H5 grid (1000 angles 0..1.5 deg x 1000 energies 0.03..15 keV), clamped outside the scanned
  range; **L1**: G1 replicated for the four LLNL coatings with distinct scale factors.
* the four transmission TSVs (raytracer.nim:1499-1527) — shipped; stored as data/detector_tables.npz.

CDF construction and the 1-D table algebra are done by the C++ host library (libsart_host.so).
"""
from __future__ import annotations

import os
from dataclasses import dataclass

import numpy as np

from . import _lib

DATA_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")

N_RADII = 1968
N_ENERGIES = 1500


def solar_grid(n_radii: int = N_RADII, n_energies: int = N_ENERGIES):
    """Radii (fractions of R_sun) and energies (keV) of the reference's emission table
    (readOpacityFile.nim:608-609, :787; raytracer.nim:438)."""
    radii = 0.0015 + 0.0005 * np.arange(n_radii, dtype=np.float64)
    energies = np.linspace(1e-3, 15.0, n_energies)
    return radii, energies


def primakoff_emission_table(n_radii: int = N_RADII, n_energies: int = N_ENERGIES) -> np.ndarray:
    """E1: Primakoff emission rate [n_radii][n_energies] from the AGSS09 plasma profile
    (Raffelt form used by readOpacityFile.nim:394-413; constant prefactors dropped — the CDFs are
    normalised).  Rows that would be identically zero (cold outermost shells) get a flat spectrum so that
    every per-radius CDF is defined; their radial weight stays negligible."""
    prof = np.load(os.path.join(DATA_DIR, "solar_profile.npz"))
    if n_radii > prof["radius"].shape[0]:
        raise ValueError("solar profile holds only %d radii" % prof["radius"].shape[0])
    _, energies = solar_grid(n_radii, n_energies)
    alpha, me = 1.0 / 137.0, 510.998
    T = prof["temp_kev"][:n_radii, None]
    ne = prof["n_e_kev3"][:n_radii, None]
    ndens = ne + prof["n_h_kev3"][:n_radii, None] + 4.0 * prof["n_he_kev3"][:n_radii, None]
    ks2 = prof["debye_ks2"][:n_radii, None]
    E = energies[None, :]
    om_pl2 = 4.0 * alpha * np.pi * ne / me
    om2 = E * E
    x = om2 / om_pl2
    ok = x > 1.0
    with np.errstate(all="ignore"):
        phase = 2.0 / (np.sqrt(1.0 - 1.0 / x) * np.expm1(E / T))
        s = 2.0 * E * np.sqrt(om2 - om_pl2)
        t = ks2 / s
        u = (2.0 * om2 - om_pl2) / s
        v = u + t
        br = np.where(u > 1.0, (u * u - 1.0) * np.log((u - 1.0) / (u + 1.0)), 0.0)
        br = br - np.where(v > 1.0, (v * v - 1.0) * np.log((v - 1.0) / (v + 1.0)), 0.0)
        br = br * 0.5 / t - 1.0
        rate = phase * ndens * br
    rate = np.where(ok & np.isfinite(rate) & (rate > 0.0), rate, 0.0)
    dead = rate.sum(axis=1) <= 0.0
    rate[dead, :] = 1e-300
    return np.ascontiguousarray(rate)


def legacy_emission_table():
    """E2: the one non-synthetic solar input shipped with the reference — `emission_rates_Hz.txt` [397 radii][233 energies]
    and `energies.txt` (written by ReadSolarModel/ReadSolarModelEmRate.cc:470-590).  Returns
    (radii, energies_keV, emission rates): radii are the first 397 rows of the AGSS09 table (0.0015 .. 0.1995 R_sun), the
    energy grid is the legacy code's non-uniform one (0.36 .. 11.4 keV)."""
    d = np.load(os.path.join(DATA_DIR, "legacy_emission.npz"))
    em = np.ascontiguousarray(d["emission_rates_hz"], dtype=np.float64)
    radii = 0.0015 + 0.0005 * np.arange(em.shape[0], dtype=np.float64)
    return radii, np.ascontiguousarray(d["energies_kev"], dtype=np.float64), em


def llnl_effective_area():
    """(energy keV, effective area cm^2) of the CAST / LLNL telescope for parallel light
    (resources/llnl_xray_telescope_cast_effective_area_parallel_light_DTU_thesis.csv)."""
    d = np.load(os.path.join(DATA_DIR, "reference_curves.npz"))
    return d["llnl_energy_kev"], d["llnl_effective_area_cm2"]


def flat_emission_table(n_radii: int = N_RADII, n_energies: int = N_ENERGIES) -> np.ndarray:
    """E3 (stress): emRate = 1 everywhere."""
    return np.ones((n_radii, n_energies))


SOLAR_MODEL_ELEMENTS = ("H1", "He4", "He3", "C12", "C13", "N14", "N15", "O16", "O17", "O18", "Ne", "Na", "Mg", "Al", "Si", "P", "S",
                        "Cl", "Ar", "K", "Ca", "Sc", "Ti", "V", "Cr", "Mn", "Fe", "Co", "Ni")   # `elements`, readOpacityFile.nim:125-128


def read_solar_model(path: str) -> dict:
    """readSolarModel (readSolarModel.nim:3-7): the blank-separated AGSS09 table with a `#` header line.  Returns the columns
    `calculateOpacities` uses: {"radius", "temp_K", "rho", "mass_fractions"[n][29]} (Radius, Temp, Rho and the 29 species)."""
    with open(path) as f:
        header = f.readline().lstrip("#").split()
    data = np.loadtxt(path, skiprows=1, ndmin=2)
    if data.shape[1] != len(header):
        raise ValueError("%s: %d columns of data for %d column names" % (path, data.shape[1], len(header)))
    col = {n: data[:, i] for i, n in enumerate(header)}
    missing = [n for n in ("Radius", "Temp", "Rho") + SOLAR_MODEL_ELEMENTS if n not in col]
    if missing:
        raise KeyError("%s: no column %s" % (path, ", ".join(missing)))   # the data frame's KeyError in the reference
    return {"radius": np.ascontiguousarray(col["Radius"]), "temp_K": np.ascontiguousarray(col["Temp"]),
            "rho": np.ascontiguousarray(col["Rho"]),
            "mass_fractions": np.ascontiguousarray(np.stack([col[n] for n in SOLAR_MODEL_ELEMENTS], axis=1))}


def read_solar_model_csv(path: str):
    """Reads the reference's ``solar_model_dataframe.csv`` (columns Radius, Energy [keV], emRates; rows
    grouped by radius; raytracer.nim:2647-2668).  Returns (radii, energies, emRates[nR][nE])."""
    data = np.loadtxt(path, delimiter=",", skiprows=1)
    radii = np.unique(data[:, 0])
    energies = np.unique(data[:, 1])
    if data.shape[0] != radii.size * energies.size:
        raise ValueError("solar model CSV is not a full radius x energy grid")
    order = np.lexsort((data[:, 1], data[:, 0]))
    em = data[order, 2].reshape(radii.size, energies.size)
    return radii, energies, np.ascontiguousarray(em)


def write_solar_model_csv(path: str, radii: np.ndarray, energies: np.ndarray, em_rates: np.ndarray):
    """Writes an emission table in the layout `calculateOpacities` returns and `main` saves
    (readOpacityFile.nim:853-854, :986: per radius one block with the columns Radius, Energy [keV], emRates) — the file
    `initFullSetup` reads back (raytracer.nim:2647-2668)."""
    em = np.asarray(em_rates, dtype=np.float64)
    if em.shape != (len(radii), len(energies)):
        raise ValueError("em_rates must be [n_radii][n_energies]")
    with open(path, "w") as f:
        f.write("Radius,Energy [keV],emRates\n")
        for i, r in enumerate(radii):
            rs = repr(float(r))
            f.write("".join("%s,%s,%s\n" % (rs, repr(float(e)), repr(float(v))) for e, v in zip(energies, em[i])))


def build_cdfs(em_rates: np.ndarray, radii: np.ndarray, energies: np.ndarray):
    """fluxRadiusCDF / diffFluxCDFs of initFullSetup (raytracer.nim:2670-2705), via libsart_host."""
    host = _lib.load_host()
    em = np.ascontiguousarray(em_rates, dtype=np.float64)
    radii = np.ascontiguousarray(radii, dtype=np.float64)
    energies = np.ascontiguousarray(energies, dtype=np.float64)
    n_r, n_e = em.shape
    rcdf = np.empty(n_r)
    ecdf = np.empty((n_r, n_e))
    _lib.check(host.sart_host_build_cdfs(_lib.as_dp(em), _lib.as_dp(radii), _lib.as_dp(energies), n_r, n_e,
                                         _lib.as_dp(rcdf), _lib.as_dp(ecdf)), host=True)
    return rcdf, ecdf


@dataclass
class ReflectivityGrid:
    """What initReflectivity reads from the H5 files (raytracer.nim:1174-1186, :1196-1209)."""
    data: np.ndarray  # [n_coatings][n_angles][n_energies]
    angle_min: float
    angle_max: float
    energy_min: float
    energy_max: float


def gold_reflectivity_grid(n_angles: int = 1000, n_energies: int = 1000) -> ReflectivityGrid:
    """G1: Henke gold 0.25 um scans resampled onto the gold H5 grid
    (angles linspace(0, 1.5) deg, energies linspace(0.03, 15) keV; tools/llnl_layer_reflectivity.nim:50-51)."""
    h = np.load(os.path.join(DATA_DIR, "gold_henke.npz"))
    a_src, e_src, r_src = h["angles_deg"], h["energy_ev"] / 1000.0, h["reflectivity"]
    angles = np.linspace(0.0, 1.5, n_angles)
    energies = np.linspace(0.03, 15.0, n_energies)
    tmp = np.stack([np.interp(energies, e_src, r_src[i]) for i in range(a_src.size)])  # [71][nE], clamped
    out = np.empty((n_angles, n_energies))
    for j in range(n_energies):
        out[:, j] = np.interp(angles, a_src, tmp[:, j])
    return ReflectivityGrid(np.ascontiguousarray(out[None]), 0.0, 1.5, 0.03, 15.0)


def llnl_reflectivity_grids(n_angles: int = 1000, n_energies: int = 1000) -> ReflectivityGrid:
    """L1: G1 replicated for the 4 LLNL coatings (/Reflectivity0..3) with distinct scale factors."""
    g = gold_reflectivity_grid(n_angles, n_energies)
    scales = np.array([1.0, 0.95, 0.9, 0.85])
    return ReflectivityGrid(np.ascontiguousarray(g.data[0][None] * scales[:, None, None]), g.angle_min, g.angle_max,
                            g.energy_min, g.energy_max)


def read_reflectivity_h5(path: str) -> ReflectivityGrid:
    """Reads `gold_0.25microns_reflectivities.h5` / `llnl_layer_reflectivities.h5` (schema of raytracer.nim:1174-1209)
    through libhdf5 (C++ host library)."""
    import ctypes as C
    host = _lib.load_host()
    nc, na, ne = C.c_int32(), C.c_int32(), C.c_int32()
    lim = [C.c_double() for _ in range(4)]
    _lib.check(host.sart_host_h5_reflectivity_info(path.encode(), C.byref(nc), C.byref(na), C.byref(ne),
                                                   *[C.byref(x) for x in lim]), host=True)
    data = np.empty((nc.value, na.value, ne.value))
    _lib.check(host.sart_host_h5_read_reflectivity(path.encode(), _lib.as_dp(data)), host=True)
    return ReflectivityGrid(data, *[x.value for x in lim])


def write_reflectivity_h5(path: str, grid: ReflectivityGrid):
    """Writes a reflectivity grid in the reference's H5 schema (what tools/convert_reflectivities_to_h5.nim and
    tools/llnl_layer_reflectivity.nim produce)."""
    host = _lib.load_host()
    nc, na, ne = grid.data.shape
    angles = np.linspace(grid.angle_min, grid.angle_max, na)
    energies = np.linspace(grid.energy_min, grid.energy_max, ne)
    _lib.check(host.sart_host_h5_write_reflectivity(path.encode(), nc, na, ne, _lib.as_dp(angles), _lib.as_dp(energies),
                                                    _lib.as_dp(np.ascontiguousarray(grid.data))), host=True)


def henke_directory_to_grid(directory: str, pattern: str = "*degGold0.25microns.csv") -> ReflectivityGrid:
    """tools/convert_reflectivities_to_h5.nim:9-32: the gold reflectivity scans downloaded from henke.lbl.gov by
    tools/download_henke_files.nim - one file per grazing angle, named `<angle>degGold0.25microns.csv` (:14, :155-161), two `#`
    header lines (column names `PhotonEnergy(eV) Reflectivity ...`, then Henke's description), one blank-separated row per
    energy - collected (walkFiles), sorted by the angle in the file name (sortedByIt) and stacked into [angle][energy].
    The energy axis is what the reference writes whatever the files say: linspace(0.03, 15.0, rows) keV (:23) - checked here
    against the files' own first column (eV), a mismatch raises."""
    import glob
    import re
    files = []
    for f in glob.glob(os.path.join(directory, pattern)):
        m = re.match(r"([-+]?[0-9]*\.?[0-9]+(?:[eE][-+]?[0-9]+)?)deg", os.path.basename(f))   # scanTuple(..., "$fdeg")
        if not m:
            raise IOError("Could not parse input gold file: " + f)
        files.append((float(m.group(1)), f))
    if len(files) < 2:
        raise IOError("%s: fewer than two files match %s" % (directory, pattern))
    files.sort(key=lambda t: t[0])
    rows = []
    energy_ev = None
    for angle, f in files:
        with open(f) as fh:
            names = fh.readline().lstrip("#").split()
        if "Reflectivity" not in names:
            raise KeyError("%s: no column Reflectivity in %s" % (f, names))
        data = np.loadtxt(f, comments="#", ndmin=2)
        if rows and data.shape[0] != rows[0].size:
            raise ValueError("%s: %d rows, the first file has %d" % (f, data.shape[0], rows[0].size))
        rows.append(np.ascontiguousarray(data[:, names.index("Reflectivity")]))
        if energy_ev is None:
            energy_ev = data[:, 0]
    n_e = rows[0].size
    energies = np.linspace(0.03, 15.0, n_e)
    if not np.allclose(energy_ev / 1000.0, energies, rtol=1e-4, atol=1e-6):
        raise ValueError("%s: the energy column is not linspace(30, 15000, %d) eV - the reference's converter assumes that axis" % (directory, n_e))
    angles = np.array([a for a, _ in files])
    if not np.allclose(angles, np.linspace(angles[0], angles[-1], angles.size), rtol=0, atol=2e-6):   # (file names carry six decimals)
        raise ValueError("%s: the angles of the files are not equidistant - the raytracer reads (min, max) of /Angles only" % directory)
    return ReflectivityGrid(np.ascontiguousarray(np.stack(rows)[None]), float(angles[0]), float(angles[-1]), 0.03, 15.0)


def convert_henke_directory_to_h5(directory: str, out_path: str, pattern: str = "*degGold0.25microns.csv") -> ReflectivityGrid:
    """The whole of tools/convert_reflectivities_to_h5.nim but its plot: `henke_download/` -> `gold_0.25microns_reflectivities.h5`
    in the schema initReflectivity reads (:34-48: /Energy (nE, 1), /Angles (nA, 1), /Reflectivity declared (nE, nA), laid out
    [angle][energy])."""
    grid = henke_directory_to_grid(directory, pattern)
    write_reflectivity_h5(out_path, grid)
    return grid


def write_henke_directory(directory: str, grid: ReflectivityGrid, thickness_microns: float = 0.25):
    """The inverse, for tests and fixtures: one `<angle>degGold<t>microns.csv` per angle of coating 0 in the format
    tools/download_henke_files.nim stores (:139-161)."""
    os.makedirs(directory, exist_ok=True)
    _, n_a, n_e = grid.data.shape
    angles = np.linspace(grid.angle_min, grid.angle_max, n_a)
    energies_ev = np.linspace(grid.energy_min, grid.energy_max, n_e) * 1000.0
    for i, a in enumerate(angles):
        with open(os.path.join(directory, "%.6fdegGold%.2fmicrons.csv" % (a, thickness_microns)), "w") as f:
            f.write("#PhotonEnergy(eV) Reflectivity Transmission\n# Au %g.nm on SiO2 at %.4fdeg, P=0.\n" % (thickness_microns * 1000.0, a))
            for e, r in zip(energies_ev, grid.data[0, i]):
                f.write("%s %s 0.0\n" % (repr(float(e)), repr(float(r))))


def analytic_reflectivity_grid(n_coatings: int = 1, n_angles: int = 1000, n_energies: int = 1000) -> ReflectivityGrid:
    """G2 (stress): R = exp(-alpha / 0.5 deg) * exp(-E / 10 keV)."""
    angles = np.linspace(0.0, 1.5, n_angles)[:, None]
    energies = np.linspace(0.03, 15.0, n_energies)[None, :]
    base = np.exp(-angles / 0.5) * np.exp(-energies / 10.0)
    scales = 1.0 - 0.05 * np.arange(n_coatings)
    return ReflectivityGrid(np.ascontiguousarray(base[None] * scales[:, None, None]), 0.0, 1.5, 0.03, 15.0)


@dataclass
class DetectorTables:
    """The three newLinear1D interpolators of newDetectorSetup (raytracer.nim:1522-1527)."""
    x_kev: np.ndarray
    strongback: np.ndarray
    window: np.ndarray
    gas_x_kev: np.ndarray
    gas_absorption: np.ndarray


def read_transmission_tsv(path: str):
    """Space-separated, one header line, eV + transmission (raytracer.nim:1503-1506)."""
    arr = np.loadtxt(path, skiprows=1)
    return np.ascontiguousarray(arr[:, 0]), np.ascontiguousarray(arr[:, 1])


def detector_tables(raw: dict | None = None) -> DetectorTables:
    """strongback = Si*Al, window = Si3N4*Al, gas absorption = 1 - T_Ar, x = eV/1000 (raytracer.nim:1509-1527).
    ``raw`` may hold the TSV columns (keys as in data/detector_tables.npz); default: the shipped copy."""
    host = _lib.load_host()
    if raw is None:
        raw = np.load(os.path.join(DATA_DIR, "detector_tables.npz"))
    cols = {k: np.ascontiguousarray(raw[k], dtype=np.float64) for k in
            ("energy_ev", "t_si3n4", "t_si", "t_al", "argon_energy_ev", "t_argon")}
    n, na = cols["energy_ev"].size, cols["argon_energy_ev"].size
    x, sb, win = np.empty(n), np.empty(n), np.empty(n)
    gx, ga = np.empty(na), np.empty(na)
    _lib.check(host.sart_host_detector_tables(
        _lib.as_dp(cols["energy_ev"]), _lib.as_dp(cols["t_si3n4"]), _lib.as_dp(cols["t_si"]), _lib.as_dp(cols["t_al"]),
        n, _lib.as_dp(cols["argon_energy_ev"]), _lib.as_dp(cols["t_argon"]), na, _lib.as_dp(x), _lib.as_dp(sb),
        _lib.as_dp(win), _lib.as_dp(gx), _lib.as_dp(ga)), host=True)
    return DetectorTables(x, sb, win, gx, ga)


def reference_curves():
    """McXtrace points and the XMM 'theory' curve the reference overlays on its angular scan
    (raytracer.nim:2805-2813)."""
    return np.load(os.path.join(DATA_DIR, "reference_curves.npz"))
