"""Absorption coefficients from the OPCD monochromatic opacity files (readOpacityFile.nim:146-296, :731-745, :790-823).

The OPCD data is not redistributable and not in the reference repository: every test writes files of the same format
(`opacity.write_*`) with made-up, smooth numbers and checks
  * the C++ readers against an independent parse of the same text in Python (CPU),
  * the loader's file selection / error behaviour against the reference's (CPU),
  * the oracle's restatement of :790-823 against a numpy evaluation (CPU),
  * the HIP kernel against the oracle (GPU) and the chain OPCD -> absCoef -> emission table -> sampling tables (GPU)."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import oracle as O
from solaraxionraytracing_amd import _lib
from solaraxionraytracing_amd import emission as em
from solaraxionraytracing_amd import opacity as op
from solaraxionraytracing_amd.tables import DATA_DIR, solar_grid


def _profile():
    return np.load(os.path.join(DATA_DIR, "solar_profile.npz"))


def _zones(indices):
    z = em.solar_zones()
    return (_lib.SolarZone * len(indices))(*[z[i] for i in indices])


def _python_parse(path):
    """Independent reader of the fmZZ.TTT format: header line, then (line 1 starting with the density index, line 2, the
    number of lines, the lines)."""
    with open(path) as f:
        lines = f.read().split("\n")
    out, i = {}, 1
    while i < len(lines) and lines[i].strip():
        density = int(lines[i].split()[0])
        n = int(lines[i + 2].strip()) or 10000
        rows = [ln.replace("D", "E").split() for ln in lines[i + 3:i + 3 + n]]
        y = np.array([float(r[-1]) for r in rows])
        x = np.arange(1, n + 1, dtype=float) if n == 10000 else np.array([float(r[0]) for r in rows])
        out[density] = (x, y)
        i += 3 + n
    return out


# ---------------------------------------------------------------------------------------------------------------- CPU ----

def test_number_densities_follow_the_reference_indexing():
    p = _profile()
    rho, x = p["rho"], p["mass_fractions"]
    n_z = op.number_densities()
    assert n_z.shape == (rho.size, 29)
    assert np.array_equal(n_z, O.emission_number_densities(rho, x))          # same expressions, same rounding
    amu = 1.6605e-24
    a = [1.0078, 4.0026, 3.0160, 12.0, 13.0033, 14.0030, 15.0001, 15.9949, 16.9991, 17.9991, 20.1797]
    assert np.allclose(n_z[:, 1], x[:, 0] / a[0] * rho / amu, rtol=1e-15)                                     # :661
    he = (x[:, 1] + x[:, 2]) ** 2 / (a[1] * x[:, 1] + a[2] * x[:, 2]) * rho / amu
    assert np.allclose(n_z[:, 2], he, rtol=1e-14)                                                             # :662-667
    assert np.allclose(n_z[:, 6], (x[:, 3] + x[:, 4]) ** 2 / (a[3] * x[:, 3] + a[4] * x[:, 4]) * rho / amu, rtol=1e-14)   # carbon
    assert np.allclose(n_z[:, 7], (x[:, 5] + x[:, 6]) ** 2 / (a[5] * x[:, 5] + a[6] * x[:, 6]) * rho / amu, rtol=1e-14)   # nitrogen
    o = x[:, 7] + x[:, 8] + x[:, 9]
    assert np.allclose(n_z[:, 8], o * o / (x[:, 7] * a[7] + x[:, 8] * a[8] + x[:, 9] * a[9]) * rho / amu, rtol=1e-14)     # :672-675
    assert np.allclose(n_z[:, 10], x[:, 10] / a[10] * rho / amu, rtol=1e-15)                                  # neon: column 10 = Z 10
    assert np.all(n_z[:, [0, 3, 4, 5, 9]] == 0.0)                                                             # never written
    # the zones' hydrogen and helium densities are the same numbers (:661-667)
    z = em.solar_zones()
    assert np.array_equal(n_z[:, 1], np.array([q.n_H for q in z]))
    assert np.array_equal(n_z[:, 2], np.array([q.n_He for q in z]))


def test_file_reader_against_an_independent_parse(tmp_path):
    rng = np.random.default_rng(5)
    y0, y1 = rng.uniform(1e-6, 3.0, 10000), rng.uniform(1e-6, 3.0, 10000)
    x2 = np.sort(rng.uniform(1.0, 1e4, 500))
    y2 = rng.uniform(1e-6, 3.0, 500)
    path = str(tmp_path / "fm26.250")
    op.write_opcd_file(path, [(74, None, y0), (96, None, y1), (80, x2, y2)])
    assert op.file_info(path) == (26, 250, [74, 96, 80], [10000, 10000, 500])
    ref = _python_parse(path)
    for density in (74, 96, 80):
        x, y = op.read_table(path, density)
        assert np.array_equal(x, ref[density][0]) and np.array_equal(y, ref[density][1])     # correctly rounded, bit for bit
    assert np.array_equal(op.read_table(path, 74)[0], np.arange(1, 10001))                   # energy = line count + 1 (:206-208)
    with pytest.raises(_lib.SartError, match="no table of density 76"):
        op.read_table(path, 76)


def test_file_reader_accepts_what_the_files_may_hold(tmp_path):
    # a 10000-line table that names its count and carries a first column: the column is ignored (:204-208); Fortran `D`
    # exponents, a leading `+`, DOS line ends and blank lines after the last table
    y = np.linspace(0.5, 1.5, 10000)
    path = str(tmp_path / "fm08.200")
    with open(path, "w") as f:
        f.write(" header\r\n   88 first\r\n second\r\n 10000\r\n")
        f.write("".join(" %.5E +%s\r\n" % (7.0, ("%.8E" % v).replace("E", "D")) for v in y))
        f.write("\r\n  \r\n")
    x, got = op.read_table(path, 88)
    assert np.array_equal(x, np.arange(1, 10001))
    assert np.array_equal(got, np.array([float("%.8E" % v) for v in y]))
    # a later table of the same density replaces the earlier one (densityTab[h1.density] = ..., :258)
    path2 = str(tmp_path / "fm08.202")
    op.write_opcd_file(path2, [(90, [1.0, 2.0, 3.0], [1.0, 1.0, 1.0]), (90, [1.0, 5.0], [2.0, 4.0])])
    x, got = op.read_table(path2, 90)
    assert np.array_equal(x, [1.0, 5.0]) and np.array_equal(got, [2.0, 4.0])


def test_file_reader_errors(tmp_path):
    good = [(74, [1.0, 2.0, 3.0], [0.1, 0.2, 0.3])]
    p = str(tmp_path / "fm06.250")
    op.write_opcd_file(p, good)
    text = open(p).read()
    cases = {
        "fm06.251": (text.rsplit("\n", 2)[0] + "\n", "ends after 2 of 3 lines"),                         # cut short
        "fm06.252": (text.replace("   74  density index", " density 74"), "Could not parse header line 1"),   # :175
        "fm06.253": (text.replace(" 2.000000E+00 2.000000E-01", " 2.0 0.2 0.3"), "Parsing opacity table in line"),   # :161
        "fm06.254": (text.replace("\n3\n", "\nthree\n"), "number of table lines"),
        "fm06.255": (text.replace(" 2.000000E+00 2.000000E-01", " 2.000000E-01"), "no abscissa column"),
    }
    for name, (content, message) in cases.items():
        q = str(tmp_path / name)
        with open(q, "w") as f:
            f.write(content)
        with pytest.raises(_lib.SartError, match=message):
            op.read_table(q, 74)
    with pytest.raises(_lib.SartError, match="Could not open file"):                                      # IOError :266
        op.read_table(str(tmp_path / "fm06.299"), 74)
    for bad in ("opacity.dat", "fm6.250", "fm06_250", "fm06.25a"):                                          # parseInt raises on these (:238-242)
        with open(tmp_path / bad, "w") as f:
            f.write(text)
        with pytest.raises(_lib.SartError, match="expected fmZZ.TTT"):
            op.read_table(str(tmp_path / bad), 74)


def test_mesh_reader(tmp_path):
    u = np.sort(np.random.default_rng(2).uniform(0.0, 20.0, 300))
    p = str(tmp_path / "fm01.mesh")
    with open(p, "w") as f:
        f.write("line u du\n" + "".join("%d %.17g %.3f\n" % (i, v, 0.1) for i, v in enumerate(u)))
    assert np.array_equal(op.read_mesh(p), u)
    with open(p, "w") as f:
        f.write("line v\n0 1.0\n")
    with pytest.raises(_lib.SartError, match="no column `u`"):
        op.read_mesh(p)
    with pytest.raises(_lib.SartError, match="Could not read mesh file"):                                  # :292-294
        op.read_mesh(str(tmp_path / "nothing.mesh"))


def test_loader_selects_the_files_and_tables_the_zones_need(tmp_path):
    zones = _zones([0, 80, 81, 1, 700])               # (288,104) (286,104) (286,102) (288,104) (270,98)
    d = op.write_stand_in_tree(str(tmp_path), zones, densities_extra=(74,), explicit_abscissa_elements=(8,))
    with op.OpcdSet(str(tmp_path), zones, n_threads=3) as s:
        assert [s.slot(k) for k in range(s.n_slots)] == [(288, 104), (286, 104), (286, 102), (270, 98)]   # order of first use
        assert s.slot_of_zone().tolist() == [0, 1, 2, 0, 3]
        T = s.tables.contents
        assert T.n_mesh == 10001 and [T.element_z[k] for k in range(T.n_elements)] == list(op.SUMMED_ELEMENTS)
        assert np.array_equal(np.ctypeslib.as_array(T.u_mesh, shape=(10001,)), op.read_mesh(os.path.join(d, "fm01.mesh")))
        for slot in range(s.n_slots):
            temp, ne = s.slot(slot)
            for col, z in enumerate(op.SUMMED_ELEMENTS):
                ref = _python_parse(os.path.join(d, "fm%02d.%d" % (z, temp)))[ne]
                x, y = s.table(slot, col)
                assert np.array_equal(x, ref[0]) and np.array_equal(y, ref[1])
                assert (T.table_x_begin[slot * T.n_elements + col] >= 0) == (z == 8)      # only oxygen carries abscissae here
    # an element the cell loop looks up without a file is the reference's KeyError (:831) - hydrogen included (:827-833)
    os.rename(os.path.join(d, "fm01.270"), os.path.join(d, "hidden"))
    with pytest.raises(_lib.SartError, match=r"no opacity file .*fm01\.270"):
        op.OpcdSet(str(tmp_path), zones)
    os.rename(os.path.join(d, "hidden"), os.path.join(d, "fm01.270"))
    # a file without the density table a zone needs
    op.write_opcd_file(os.path.join(d, "fm26.270"), [(74, None, np.ones(10000))])
    with pytest.raises(_lib.SartError, match=r"fm26\.270: no table of density 98"):
        op.OpcdSet(str(tmp_path), zones)
    # the mesh must have its 10001 lines (doAssert :281)
    op.write_mesh(os.path.join(d, "fm01.mesh"), np.linspace(0, 20, 100))
    with pytest.raises(_lib.SartError, match="expected 10001"):
        op.OpcdSet(str(tmp_path), zones)
    with pytest.raises(_lib.SartError, match="Could not read mesh file"):
        op.OpcdSet(str(tmp_path / "elsewhere"), zones)


def _numpy_abs_coefs(zones, n_z, energies, s):
    """:790-823 with numpy's interpolation (an independent evaluation of the oracle's restatement)."""
    T = s.tables.contents
    u = np.ctypeslib.as_array(T.u_mesh, shape=(T.n_mesh,))
    out = np.zeros((len(zones), energies.size))
    slot_of = s.slot_of_zone()
    for r, z in enumerate(zones):
        t_table = 10.0 ** (z.temp_index * 0.025) * 8.617e-8
        w = energies / t_table
        inside = (w < 20.0) & (w > 0.0732)
        line = np.interp(w[inside], u, np.arange(T.n_mesh, dtype=float))
        total = np.zeros(line.size)
        for col, el in enumerate(op.SUMMED_ELEMENTS):
            x, y = s.table(int(slot_of[r]), col)
            total += n_z[r, el] * np.interp(line, x, y)
        out[r, inside] = total * 1.97327e-8 * 0.528e-8 * 0.528e-8 * (1.0 - np.exp(-energies[inside] / (z.temp_K * 8.617e-8)))
    return out


def test_oracle_abs_coefs_against_numpy(tmp_path):
    idx = [0, 300, 900, 1500, 1960]
    zones = _zones(idx)
    n_z = op.number_densities()[idx]
    op.write_stand_in_tree(str(tmp_path), zones, explicit_abscissa_elements=(6, 26))
    _, energies = solar_grid(8, 1500)
    with op.OpcdSet(str(tmp_path), zones) as s:
        got, n_outside = O.emission_abs_coefs(zones, n_z, energies, s.tables)
        want = _numpy_abs_coefs(zones, n_z, energies, s)
    assert n_outside == 0
    assert np.all(got >= 0.0) and got.max() > 0.0
    # outside 0.0732 < w < 20 the coefficient is exactly 0 (:801-808): the hot core at low energies, the cool edge at high ones
    w = energies[None, :] / np.array([10.0 ** (z.temp_index * 0.025) * 8.617e-8 for z in zones])[:, None]
    off = (w >= 20.0) | (w <= 0.0732)
    assert off.any() and (~off).any() and np.all(got[off] == 0.0) and np.all(got[~off] > 0.0)
    assert np.allclose(got, want, rtol=2e-13, atol=0.0)


def test_oracle_counts_cells_that_leave_a_table(tmp_path):
    zones = _zones([0, 1900])
    n_z = op.number_densities()[[0, 1900]]
    d = op.write_stand_in_tree(str(tmp_path), zones)
    # a mesh that sends w = 0.0732 .. 0.2 below line 1, where the 10000-line tables begin: numericalnim raises there
    op.write_mesh(os.path.join(d, "fm01.mesh"), np.linspace(0.0, 2000.0, 10001))
    _, energies = solar_grid(8, 1500)
    with op.OpcdSet(str(tmp_path), zones) as s:
        got, n_outside = O.emission_abs_coefs(zones, n_z, energies, s.tables)
    assert n_outside > 0 and int(np.isnan(got).sum()) == n_outside


# ---------------------------------------------------------------------------------------------------------------- GPU ----

@pytest.mark.gpu
def test_abs_coefs_kernel_against_the_oracle(tmp_path):
    idx = list(range(0, 1968, 41))                    # 48 zones over the whole model: every kind of slot
    zones = _zones(idx)
    n_z = op.number_densities()[idx]
    op.write_stand_in_tree(str(tmp_path), zones, densities_extra=(74,), explicit_abscissa_elements=(7, 26))
    _, energies = solar_grid(8, 1500)
    with op.OpcdSet(str(tmp_path), zones) as s:
        want, n_outside = O.emission_abs_coefs(zones, n_z, energies, s.tables)
        got = op.abs_coefs(zones, n_z, energies, s)
    assert n_outside == 0
    assert np.array_equal(got == 0.0, want == 0.0)
    # same operations in the same order without contraction; the device's exp() may differ from glibc's in the last place
    # of exp(-E/T), which the subtraction 1 - exp(-E/T) magnifies by 1 / (1 - exp(-E/T)) at small E/T
    nz = want != 0.0
    f = (1.0 - np.exp(-energies[None, :] / np.array([z.temp_K * 8.617e-8 for z in zones])[:, None]))[nz]
    assert np.max(np.abs(got[nz] - want[nz]) / want[nz] * f) < 4e-16
    assert np.mean(got == want) > 0.9


@pytest.mark.gpu
def test_abs_coefs_kernel_refuses_what_the_reference_raises_on(tmp_path):
    zones = _zones([0, 1900])
    n_z = op.number_densities()[[0, 1900]]
    d = op.write_stand_in_tree(str(tmp_path), zones)
    op.write_mesh(os.path.join(d, "fm01.mesh"), np.linspace(0.0, 2000.0, 10001))
    _, energies = solar_grid(8, 1500)
    with op.OpcdSet(str(tmp_path), zones) as s:
        _, n_outside = O.emission_abs_coefs(zones, n_z, energies, s.tables)
        with pytest.raises(_lib.SartError, match="%d cells evaluate a table outside" % n_outside):
            op.abs_coefs(zones, n_z, energies, s)
        # descriptions that point outside their pools never reach the GPU
        T = s.tables.contents
        bad = _lib.OpacityTables.from_buffer_copy(T)
        bad.n_table_y = 5
        with pytest.raises(_lib.SartError, match="lies outside the opacity pool"):
            op.abs_coefs(zones, n_z, energies, C.pointer(bad))
        bad = _lib.OpacityTables.from_buffer_copy(T)
        bad.n_slots = 1
        with pytest.raises(_lib.SartError, match="slot_of_zone"):
            op.abs_coefs(zones, n_z, energies, C.pointer(bad))


@pytest.mark.gpu
def test_emission_table_with_opcd_coefficients(tmp_path):
    """The chain of calculateOpacities (:731-860): files -> absCoef -> the three terms that use it, against the oracle's."""
    idx = [0, 150, 400, 800, 1200, 1600, 1950]
    zones = _zones(idx)
    n_z = op.number_densities()[idx]
    op.write_stand_in_tree(str(tmp_path), zones)
    _, energies = solar_grid(8, 1500)
    energies = energies[::10]
    params = em.default_params()
    with op.OpcdSet(str(tmp_path), zones) as s:
        absc = op.abs_coefs(zones, n_z, energies, s)
        o_abs, _ = O.emission_abs_coefs(zones, n_z, energies, s.tables)
    total, comp = em.emission_table(zones, energies, abs_coefs=absc, params=params, components=True)
    o_total, o_comp = O.emission_table(zones, energies, params, abs_coefs=o_abs, components=True)
    for k in (1, 5, 6):                                # term1, longitudinal and transverse plasmon
        assert np.any(o_comp[k] != 0.0)
        assert np.allclose(comp[k], o_comp[k], rtol=1e-9, atol=0.0), _lib.EM_TERMS[k]
    assert np.allclose(total, o_total, rtol=1e-9)
    without = em.emission_table(zones, energies, params=params)
    assert np.all(total >= without) and np.any(total > without * (1 + 1e-9))


@pytest.mark.gpu
def test_opcd_to_sampling_tables_on_the_device(tmp_path):
    """absCoef left on the device -> sart_emission_to_solar_tables: the same CDFs as the host route."""
    import torch
    from solaraxionraytracing_amd import raytracer as rt
    idx = list(range(0, 1968, 164))
    zones = _zones(idx)
    n_z = np.ascontiguousarray(op.number_densities()[idx])
    op.write_stand_in_tree(str(tmp_path), zones)
    radii, energies = solar_grid(len(idx), 300)
    radii = np.array([z.radius_frac for z in zones])
    params = em.default_params()
    lib = _lib.load_sart()
    full = rt.initFullSetup(n_radii=16, n_energies=32)
    with op.OpcdSet(str(tmp_path), zones) as s, rt.RayTracer(full) as tracer:
        d_abs = torch.empty(len(idx) * energies.size, dtype=torch.float64, device="cuda")
        _lib.check(lib.sart_emission_abs_coefs_device(tracer.handle, zones, len(idx), _lib.as_dp(n_z), _lib.as_dp(energies), energies.size,
                                                      s.tables, C.c_void_p(d_abs.data_ptr())))
        torch.cuda.synchronize()
        host_abs = op.abs_coefs(zones, n_z, energies, s)
        assert np.array_equal(d_abs.cpu().numpy().reshape(len(idx), -1), host_abs)
        _lib.check(lib.sart_emission_to_solar_tables(tracer.handle, zones, len(idx), _lib.as_dp(energies), energies.size,
                                                     C.c_void_p(d_abs.data_ptr()), C.byref(params)))
        tracer.full.energies, tracer._n_radii_set = energies, len(idx)
        got = tracer.solar_tables()
    rates = em.emission_table(zones, energies, abs_coefs=host_abs, params=params)
    host = _lib.load_host()
    rcdf, ecdf = np.empty(len(idx)), np.empty((len(idx), energies.size))
    _lib.check(host.sart_host_build_cdfs(_lib.as_dp(rates), _lib.as_dp(radii), _lib.as_dp(energies), len(idx), energies.size,
                                         _lib.as_dp(rcdf), _lib.as_dp(ecdf)), host=True)
    assert np.array_equal(got[0], rcdf) and np.array_equal(got[1], ecdf)


# ------------------------------------------------------------------------------------------- the pre-processor's driver ----

def _write_solar_model(path, rows):
    """The shipped AGSS09 columns back into the layout of AGSS09_solar_model_stripped.dat (`#` header, blank-separated)."""
    from solaraxionraytracing_amd import tables
    p = _profile()
    names = ["Mass", "Radius", "Temp", "Rho", "Pres", "Lumi"] + list(tables.SOLAR_MODEL_ELEMENTS)
    with open(path, "w") as f:
        f.write("#  " + "  ".join(names) + "\n")
        for i in rows:
            vals = [0.0, p["radius"][i], p["temp_K"][i], p["rho"][i], 0.0, 0.0] + list(p["mass_fractions"][i])
            f.write("  ".join(repr(float(v)) for v in vals) + "\n")


def _write_config(tmp_path, opcd="OPCD"):
    (tmp_path / "config").mkdir()
    (tmp_path / "resources").mkdir()
    cfg = tmp_path / "config" / "config.toml"
    cfg.write_text('[Resources]\nresourcePath = "../resources"\noutputPath = "../out"\nrawSolarModel = "model.dat"\n'
                   'solarModelFile = "solar_model_dataframe.csv"\n[ReadOpacityFile]\nsolarModelFile = "solar_model_dataframe.csv"\n'
                   'opcdPath = "%s"\n[Setup]\nexperimentSetup = "BabyIAXO"\ndetectorSetup = "InGridIAXO"\nstageSetup = "vacuum"\n'
                   'telescopeSetup = "XMM"\n' % opcd)
    return str(cfg)


def test_solar_model_reader_and_opcd_path(tmp_path):
    from solaraxionraytracing_amd import config as cfgmod, tables
    rows = [0, 1, 2, 700, 1967]
    _write_solar_model(str(tmp_path / "model.dat"), rows)
    got = tables.read_solar_model(str(tmp_path / "model.dat"))
    p = _profile()
    for key in ("radius", "temp_K", "rho", "mass_fractions"):
        assert np.array_equal(got[key], p[key][rows]), key
    with open(tmp_path / "short.dat", "w") as f:
        f.write("# Radius Temp\n0.1 1e7\n")
    with pytest.raises(KeyError, match="no column Rho"):
        tables.read_solar_model(str(tmp_path / "short.dat"))
    # opcdPath: as given if it is a directory (the reference's behaviour), else beside the config file
    cfg = cfgmod.load_config(_write_config(tmp_path))
    base = str(tmp_path / "config")
    assert cfgmod.resolve_opcd_path(cfg, base) == os.path.join(base, "OPCD")
    assert cfgmod.resolve_opcd_path({"ReadOpacityFile": {"opcdPath": str(tmp_path)}}, base) == str(tmp_path)
    assert cfgmod.resolve_opcd_path({}, base) is None


@pytest.mark.gpu
def test_read_opacity_file_driver(tmp_path, capsys):
    """`readOpacityFile` end to end on a 12-zone model: config -> model + OPCD files -> solar_model_dataframe.csv, and the
    raytracer's config path picking the same OPCD directory up when the CSV is not there."""
    from solaraxionraytracing_amd import config as cfgmod, read_opacity_file as rof, tables
    rows = list(range(0, 1968, 164))
    cfg = _write_config(tmp_path, opcd="../OPCD")
    _write_solar_model(str(tmp_path / "resources" / "model.dat"), rows)
    profile = tables.read_solar_model(str(tmp_path / "resources" / "model.dat"))
    zones = em.solar_zones(profile=profile)
    op.write_stand_in_tree(str(tmp_path / "OPCD"), zones)
    assert rof.main(["--config", cfg]) == 0
    out = capsys.readouterr().out
    assert "OPCD: " in out and "57Fe Flux" in out
    radii, energies, rates = tables.read_solar_model_csv(str(tmp_path / "out" / "solar_model_dataframe.csv"))
    assert rates.shape == (12, 1500) and np.allclose(radii, 0.0015 + 0.0005 * np.arange(12))    # radius = 0.0015 + R 0.0005 (:751)
    n_z = op.number_densities(profile)
    with op.OpcdSet(str(tmp_path / "OPCD"), zones) as s:
        o_abs, _ = O.emission_abs_coefs(zones, n_z, energies, s.tables)
    want = O.emission_table(zones, energies, em.default_params(), abs_coefs=o_abs, e_stride=25)
    sel = ~np.isnan(want)
    assert np.allclose(rates[sel], want[sel], rtol=1e-9)
    flux = np.loadtxt(str(tmp_path / "out" / "diff_flux.csv"), delimiter=",", skiprows=1)
    assert flux.shape == (1500, 10) and np.allclose(flux[:, 1], em.flux_spectrum(rates, energies), rtol=1e-12)
    # without the OPCD files the run goes on with absCoef = 0 and says so
    os.rename(str(tmp_path / "OPCD"), str(tmp_path / "OPCD_away"))
    assert rof.main(["--config", cfg]) == 0
    assert "absorption coefficients set to 0" in capsys.readouterr().out
    bare = tables.read_solar_model_csv(str(tmp_path / "out" / "solar_model_dataframe.csv"))[2]
    assert np.all(bare <= rates) and np.any(bare < rates * (1 - 1e-9))
    os.rename(str(tmp_path / "OPCD_away"), str(tmp_path / "OPCD"))
