"""Absorption coefficients of the solar plasma from the OPCD 3.3 monochromatic opacity files
(readOpacityFile.nim:146-296 the file readers, :731-745 the file selection, :790-823 the coefficient itself).

The files are read by the C++ host library (``sart_host_opcd_*``, include/sart_host.h), the coefficients are computed by the
HIP kernel behind ``sart_emission_abs_coefs`` (include/sart_emission.h).  ctypes plumbing only; raises without a GPU where a
kernel is involved: there is no CPU path.  The OPCD data is not redistributable and not part of the reference repository."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib
from .tables import DATA_DIR

# proton numbers `calculateOpacities` looks an opacity up for (:827-831) / adds to the sum (:833)
LOOKED_UP_ELEMENTS = (1, 2, 6, 7, 8, 10, 11, 12, 13, 14, 16, 18, 20, 24, 25, 26, 28)
SUMMED_ELEMENTS = LOOKED_UP_ELEMENTS[2:]


def mono_dir(opcd_path: str) -> str:
    """``<opcdPath>/OPCD_3.3/mono`` (:290, :739)"""
    return os.path.join(opcd_path, "OPCD_3.3", "mono")


def number_densities(profile: dict | None = None, n_radii: int | None = None) -> np.ndarray:
    """n_Z[n_radii][29] in 1/cm^3, indexed by proton number (:655-679), from the AGSS09 columns shipped in
    data/solar_profile.npz (or ``profile`` = {"rho", "mass_fractions"[n][29]})."""
    host = _lib.load_host()
    if profile is None:
        profile = np.load(os.path.join(DATA_DIR, "solar_profile.npz"))
    rho = np.ascontiguousarray(profile["rho"], dtype=np.float64)
    frac = np.ascontiguousarray(profile["mass_fractions"], dtype=np.float64)
    if n_radii is None:
        n_radii = rho.shape[0]
    if n_radii > rho.shape[0] or frac.shape != (rho.shape[0], 29):
        raise ValueError("solar profile: need %d radii and 29 mass fractions per radius" % n_radii)
    out = np.empty((n_radii, 29))
    _lib.check(host.sart_host_solar_number_densities(_lib.as_dp(rho), _lib.as_dp(frac), n_radii, _lib.as_dp(out)), host=True)
    return out


def read_mesh(path: str) -> np.ndarray:
    """Column ``u`` of a mesh file (readMeshDataFile :278-282)."""
    host = _lib.load_host()
    n = C.c_int32()
    _lib.check(host.sart_host_opcd_read_mesh(path.encode(), None, 0, C.byref(n)), host=True)
    u = np.empty(n.value)
    _lib.check(host.sart_host_opcd_read_mesh(path.encode(), _lib.as_dp(u), u.size, C.byref(n)), host=True)
    return u


def file_info(path: str):
    """(element, temperature index, [density index ...], [table lines ...]) of one ``fmZZ.TTT`` file."""
    host = _lib.load_host()
    el, temp, n = C.c_int32(), C.c_int32(), C.c_int32()
    _lib.check(host.sart_host_opcd_file_info(path.encode(), C.byref(el), C.byref(temp), C.byref(n), None, None, 0), host=True)
    dens = (C.c_int32 * max(n.value, 1))()
    lens = (C.c_int32 * max(n.value, 1))()
    _lib.check(host.sart_host_opcd_file_info(path.encode(), C.byref(el), C.byref(temp), C.byref(n), dens, lens, n.value), host=True)
    return el.value, temp.value, list(dens[:n.value]), list(lens[:n.value])


def read_table(path: str, density: int):
    """(abscissae, opacities) of the table of one density index (parseDensityTab :184-217)."""
    host = _lib.load_host()
    n = C.c_int32()
    _lib.check(host.sart_host_opcd_read_table(path.encode(), density, None, None, 0, C.byref(n)), host=True)
    x, y = np.empty(n.value), np.empty(n.value)
    _lib.check(host.sart_host_opcd_read_table(path.encode(), density, _lib.as_dp(x), _lib.as_dp(y), n.value, C.byref(n)), host=True)
    return x, y


class OpcdSet:
    """The mesh and the opacity tables the zones of one solar model need (``sart_host_opcd_load``)."""

    def __init__(self, opcd_path: str, zones, n_threads: int = 0):
        host = _lib.load_host()
        self._host = host
        self._h = C.c_void_p()
        _lib.check(host.sart_host_opcd_load(opcd_path.encode(), zones, len(zones), n_threads, C.byref(self._h)), host=True)
        self.tables = host.sart_host_opcd_tables(self._h)   # POINTER(OpacityTables), valid until close()
        self.n_radii = len(zones)

    def close(self):
        if self._h:
            self._host.sart_host_opcd_free(self._h)
            self._h = C.c_void_p()
            self.tables = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # views for tests and tools
    @property
    def n_slots(self) -> int:
        return int(self.tables.contents.n_slots)

    def slot(self, k: int):
        t, ne = C.c_int32(), C.c_int32()
        _lib.check(self._host.sart_host_opcd_slot(self._h, k, C.byref(t), C.byref(ne)), host=True)
        return t.value, ne.value

    def slot_of_zone(self) -> np.ndarray:
        return np.ctypeslib.as_array(self.tables.contents.slot_of_zone, shape=(self.n_radii,)).copy()

    def table(self, slot: int, element_column: int):
        """(abscissae, opacities) as the kernel sees them."""
        T = self.tables.contents
        cell = slot * T.n_elements + element_column
        n, yb, xb = T.table_len[cell], T.table_y_begin[cell], T.table_x_begin[cell]
        y = np.ctypeslib.as_array(T.table_y, shape=(T.n_table_y,))[yb:yb + n].copy()
        x = np.arange(1, n + 1, dtype=np.float64) if xb < 0 else np.ctypeslib.as_array(T.table_x, shape=(T.n_table_x,))[xb:xb + n].copy()
        return x, y


def abs_coefs(zones, n_z, energies, tables, device: int = 0) -> np.ndarray:
    """absCoef[n_radii][n_energies] in keV (:790-823) from the HIP kernel; ``tables`` = OpcdSet or POINTER(OpacityTables)."""
    lib = _lib.load_sart()
    energies = np.ascontiguousarray(energies, dtype=np.float64)
    n_z = np.ascontiguousarray(n_z, dtype=np.float64)
    if n_z.shape != (len(zones), 29):
        raise ValueError("n_z must be [n_radii][29]")
    t = tables.tables if isinstance(tables, OpcdSet) else tables
    out = np.empty((len(zones), energies.size))
    ctx = C.c_void_p()
    _lib.check(lib.sart_create(device, C.byref(ctx)))
    try:
        _lib.check(lib.sart_emission_abs_coefs(ctx, zones, len(zones), _lib.as_dp(n_z), _lib.as_dp(energies), energies.size, t,
                                               _lib.as_dp(out)))
    finally:
        lib.sart_destroy(ctx)
    return out


# ---- writers of the two file formats (what the readers above accept; used to build stand-in data where the OPCD files are
# ---- not available, e.g. by the tests and by tools/opcd_time.py) --------------------------------------------------------

def write_mesh(path: str, u) -> None:
    """``fm01.mesh``: blank-separated columns with a header line, one of them ``u``."""
    u = np.asarray(u, dtype=np.float64)
    with open(path, "w") as f:
        f.write("i u\n")
        f.write("\n".join("%d %.17g" % (i, v) for i, v in enumerate(u)))
        f.write("\n")


def write_opcd_file(path: str, tables, header: str = "monochromatic opacities (stand-in data)", fmt: str = "%.6E") -> None:
    """``fmZZ.TTT``: ``tables`` = [(density index, abscissae or None, opacities)].  A table of 10000 lines without abscissae
    is written with the line count ``0`` (the reference reads that as 10000, readOpacityFile.nim:256)."""
    with open(path, "w") as f:
        f.write(" " + header + "\n")
        for density, x, y in tables:
            y = np.asarray(y, dtype=np.float64)
            f.write("%5d  density index\n" % density)
            f.write(" second header line\n")
            if x is None:
                if y.size != 10000:
                    raise ValueError("a table without abscissae has 10000 lines")
                f.write("0\n")
                f.write("\n".join(np.char.mod(fmt, y)))
            else:
                x = np.asarray(x, dtype=np.float64)
                f.write("%d\n" % y.size)
                f.write("\n".join(np.char.add(np.char.add(np.char.mod(" " + fmt, x), " "), np.char.mod(fmt, y))))
            f.write("\n")


def write_stand_in_tree(opcd_path: str, zones, densities_extra=(), seed: int = 1, explicit_abscissa_elements=()) -> str:
    """A directory of the OPCD 3.3 layout holding smooth made-up opacities for every (temperature, density) pair the zones
    need: NOT physical data, only the right shape (for tests and timing).  Returns the ``mono`` directory."""
    d = mono_dir(opcd_path)
    os.makedirs(d, exist_ok=True)
    lines = np.arange(10001, dtype=np.float64)
    write_mesh(os.path.join(d, "fm01.mesh"), 20.0 * (lines / 10000.0) ** 1.2)
    want = {}
    for z in zones:
        want.setdefault(int(z.temp_index), set()).add(int(z.ne_index))
    rng = np.random.default_rng(seed)
    x = np.arange(1, 10001, dtype=np.float64)
    for temp, nes in sorted(want.items()):
        for el in LOOKED_UP_ELEMENTS:
            tabs = []
            for ne in sorted(nes | set(densities_extra)):
                a, b, c = rng.uniform(0.5, 2.0), rng.uniform(1e-4, 1e-3), rng.uniform(0.0, 6.28)
                y = a * el * 1e-2 * np.exp(-b * x) * (1.2 + np.sin(x * 2e-3 * (1 + el % 5) + c)) * (1.0 + 0.01 * (ne - 74)) + 1e-6
                if el in explicit_abscissa_elements:
                    sel = np.unique(np.concatenate([[0, 9999], rng.choice(10000, 700, replace=False)]))
                    tabs.append((ne, x[sel], y[sel]))
                else:
                    tabs.append((ne, None, y))
            write_opcd_file(os.path.join(d, "fm%02d.%d" % (el, temp)), tabs)
    return d
