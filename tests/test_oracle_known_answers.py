"""Pins the CPU oracle against every known-answer value the reference's own text holds for the hot path
(SURVEY.md 8c (i)-(ix)).  The reference ships no runnable tests or golden vectors for this path and cannot
be built here (Nim), so these are what ties oracle -> reference."""
import ctypes as C
import math

import numpy as np
import pytest

from oracle import oracle as O
from solaraxionraytracing_amd import _lib as L
import solaraxionraytracing_amd as sa

_dp = C.POINTER(C.c_double)


@pytest.fixture(scope="module")
def lib():
    return O.load("f64")


def v3(*a):
    return (C.c_double * 3)(*a)


def test_eff_photon_mass2_table(lib):
    # axionMassforMagnet.nim:115-119 `when isMainModule` comments (p mbar, 10 m, 0.35 m, 100 K)
    for p, m in [(36.61, 0.0853), (109.8, 0.1477), (183.05, 0.1907), (366.1, 0.2698)]:
        assert abs(lib.sart_oracle_eff_photon_mass2(p, 10.0, 0.35, 100.0) - m) < 1e-4  # 4 printed digits
    # raytracer.nim:255 comment: 0.26978249412621896 eV "corresponds to set p and T gas values"
    assert abs(lib.sart_oracle_eff_photon_mass2(366.1, 10.0, 0.35, 100.0) - 0.26978249412621896) < 2e-5


def test_vacuum_conversion_probability(lib):
    # raytracer.nim:363-365 with g = 1e-12 GeV^-1: CAST 9 T x 9.26 m ~ 1.70e-21, BabyIAXO 2 T x 11 m ~ 1.19e-22
    assert lib.sart_oracle_conversion_prob(9.0, 1e-12, 9260.0) == pytest.approx(1.70e-21, rel=5e-3)
    assert lib.sart_oracle_conversion_prob(2.0, 1e-12, 11000.0) == pytest.approx(1.19e-22, rel=5e-3)
    # P scales as (B L)^2
    r = lib.sart_oracle_conversion_prob(4.0, 1e-12, 5000.0) / lib.sart_oracle_conversion_prob(2.0, 1e-12, 2500.0)
    assert r == pytest.approx(16.0, rel=1e-12)


def test_gas_conversion_reduces_to_vacuum_form(lib):
    # axionMassforMagnet.nim:75-101: for Gamma -> 0 and q -> 0, P -> (g B L / 2)^2 (in its own unit constants)
    L_m, B, g = 10.0, 2.0, 1e-12
    p = lib.sart_oracle_axion_conversion_prob2(0.0, 4.2, 1e-12, 100.0, L_m, 0.35, g, B)
    expect = (g * 1e-9 * (B * 1e3 / 1.444) / 2.0) ** 2 * (L_m / 1.97e-7) ** 2
    assert p == pytest.approx(expect, rel=1e-6)
    # absorption is 1 without gas and decreases with pressure
    assert lib.sart_oracle_intensity_suppression2(4.2, 10.0, 5.0, 0.0, 100.0, 293.15) == 1.0
    assert lib.sart_oracle_intensity_suppression2(1.0, 10.0, 5.0, 300.0, 100.0, 293.15) < \
        lib.sart_oracle_intensity_suppression2(1.0, 10.0, 5.0, 30.0, 100.0, 293.15) < 1.0


def test_window_strip_geometry():
    # calcWindowVals raytracer.nim:1431-1462 for R = 7, 4 strips, 0.838 (also calculateWindowValues.nim)
    host = L.load_host()
    w, d = C.c_double(), C.c_double()
    assert host.sart_host_calc_window_vals(7.0, 4, 0.838, C.byref(w), C.byref(d)) == 0
    assert w.value == pytest.approx(0.500418, abs=2e-6)
    assert d.value == pytest.approx(2.299582, abs=2e-6)
    # the two strip half-lengths the reference echoes (:1453)
    dw = 14.0 / 5.0
    assert 2 * math.sqrt(49 - (0.5 * dw) ** 2) == pytest.approx(13.7171, abs=1e-4)
    assert 2 * math.sqrt(49 - (1.5 * dw) ** 2) == pytest.approx(11.2, abs=1e-4)


def test_coating_map_llnl():
    # layers = [2, 5, 9, 14]; lowerBound(hitLayer) => [0,0,0,1,1,1,2,2,2,2,3,3,3,3] (raytracer.nim:1167, :1573)
    layers = np.array([2.0, 5.0, 9.0, 14.0])
    lib = O.load("f64")
    got = [lib.sart_oracle_lower_bound(layers.ctypes.data_as(_dp), 4, float(h)) for h in range(14)]
    assert got == [0, 0, 0, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3]


def test_length_telescope(lib):
    # raytracer.nim:1883-1884: 599.9616 mm (XMM), 454.0551 mm (LLNL)
    xmm = sa.newFullSetup(L.ES_BABYIAXO, L.DK_INGRIDIAXO, L.SK_VACUUM, L.TK_XMM)
    llnl = sa.newFullSetup(L.ES_CAST, L.DK_INGRID2018, L.SK_VACUUM, L.TK_LLNL)
    assert lib.sart_oracle_length_telescope(C.byref(xmm)) == pytest.approx(599.9616, abs=1e-4)
    assert lib.sart_oracle_length_telescope(C.byref(llnl)) == pytest.approx(454.0551, abs=1e-4)


def test_std_helpers(lib):
    a = np.array([0.1, 0.2, 0.2, 0.7, 1.0])
    p = a.ctypes.data_as(_dp)
    # std/algorithm.lowerBound: first index with a[i] >= key
    assert [lib.sart_oracle_lower_bound(p, 5, k) for k in (0.0, 0.1, 0.15, 0.2, 0.21, 1.0, 1.1)] == [0, 0, 1, 1, 3, 4, 5]
    # std/math.almostEqual, 4 ulp
    assert lib.sart_oracle_almost_equal(1.0, 1.0 + 2 ** -52) == 1
    assert lib.sart_oracle_almost_equal(1.0, 1.0 + 2 ** -48) == 0
    assert lib.sart_oracle_almost_equal(-475.0, -475.0) == 1


def test_interpolators(lib):
    # numericalnim linear1D: exact on nodes, linear in between, end intervals at the ends
    xs = np.array([0.0, 1.0, 3.0, 4.0]); ys = np.array([0.0, 2.0, 2.0, 6.0])
    f = lambda x: lib.sart_oracle_linear1d(xs.ctypes.data_as(_dp), ys.ctypes.data_as(_dp), 4, x)
    assert [f(0.0), f(0.5), f(1.0), f(2.0), f(3.5), f(4.0)] == [0.0, 1.0, 2.0, 2.0, 4.0, 6.0]
    # numericalnim bilinear on a uniform grid: reproduces a bilinear function exactly
    nx, ny = 7, 5
    gx, gy = np.linspace(0.0, 1.5, nx), np.linspace(0.03, 15.0, ny)
    z = np.ascontiguousarray(2.0 + 3.0 * gx[:, None] - 0.5 * gy[None, :] + 0.25 * gx[:, None] * gy[None, :])
    for x, y in [(0.0, 0.03), (0.4, 3.0), (1.5, 15.0), (1.49, 14.9), (0.77, 0.031)]:
        got = lib.sart_oracle_bilinear(z.ctypes.data_as(_dp), nx, ny, 0.0, 1.5, 0.03, 15.0, x, y)
        assert got == pytest.approx(2.0 + 3.0 * x - 0.5 * y + 0.25 * x * y, rel=1e-13)


def test_xmm_shell_geometry(lib):
    # SURVEY 8(a) derived check values: XMM shell r1 = 247.2855, beta = 0.467 deg, f = 7500
    r1, beta, l, f = 247.2855, math.radians(0.467), 300.0, 7500.0
    for t, z1, z2, a1, a2 in [(1e-7, 158.867, 438.495, 0.46483, 0.47354), (1e-3, None, None, 0.52185, 0.41776)]:
        pcb, pxrt = v3(246 - 475 * t, 1e-7, -475.0), v3(246.0, 0.0, 0.0)
        m1, va, m2 = v3(0, 0, 0), v3(0, 0, 0), v3(0, 0, 0)
        lib.sart_oracle_find_pos(1, pxrt, pcb, r1, beta, l, 0.0, f, m1)
        lib.sart_oracle_vector_after_mirror(pxrt, pcb, m1, beta, r1, l, f, 1, va)
        after = v3(*[m1[i] + 200.0 * va[i] for i in range(3)])
        lib.sart_oracle_find_pos(2, after, m1, r1, 3 * beta, l, math.cos(beta) * l, f, m2)
        alpha1 = lib.sart_oracle_mirror_angle_deg(pxrt, pcb, m1, beta, r1, l, f, 1)
        alpha2 = lib.sart_oracle_mirror_angle_deg(after, m1, m2, 3 * beta, r1, l, f, 2)
        if z1 is not None:
            assert m1[2] == pytest.approx(z1, abs=2e-3) and m2[2] == pytest.approx(z2, abs=2e-3)
        assert alpha1 == pytest.approx(a1, abs=2e-5) and alpha2 == pytest.approx(a2, abs=2e-5)


def test_llnl_shell7_testmirrors_scenario(lib):
    # TestMirrors.nim:82-121 inputs (LLNL shell r1 = 83.405, beta = 0.767 deg, xSep = 4.284, axis-parallel ray at
    # x = 81.5); SURVEY 8(a): hits (81.5, 0, 142.297) and (76.958, 0, 311.885), alpha1 = alpha2 = 0.767 deg
    r1, xsep, beta, l = 83.405, 4.284, math.radians(0.767), 225.0
    r2 = r1 - l * math.sin(beta); r3 = r2 - 0.5 * xsep * math.tan(beta); r4 = r3 - 0.5 * xsep * math.tan(3 * beta)
    pcb, pxrt = v3(81.5, 0.0, -239.36), v3(81.5, 0.0, 0.0)
    m1, va, m2 = v3(0, 0, 0), v3(0, 0, 0), v3(0, 0, 0)
    lib.sart_oracle_find_pos(0, pxrt, pcb, r1, beta, l, 0.0, 1485.0, m1)
    assert (m1[0], m1[1]) == (81.5, 0.0) and m1[2] == pytest.approx(142.297, abs=2e-3)
    lib.sart_oracle_vector_after_mirror(pxrt, pcb, m1, beta, r1, l, 1485.0, 0, va)
    after = v3(*[m1[i] + 200.0 * va[i] for i in range(3)])
    lib.sart_oracle_find_pos(0, after, m1, r4, 3 * beta, l, math.cos(beta) * (xsep + l), 1485.0, m2)
    assert m2[0] == pytest.approx(76.958, abs=2e-3) and m2[2] == pytest.approx(311.885, abs=2e-3)
    a1 = lib.sart_oracle_mirror_angle_deg(pxrt, pcb, m1, beta, r1, l, 1485.0, 0)
    a2 = lib.sart_oracle_mirror_angle_deg(after, m1, m2, 3 * beta, r1, l, 1485.0, 0)
    assert a1 == pytest.approx(0.767, abs=1e-10) and a2 == pytest.approx(0.767, abs=1e-10)
    # the (stale) assertion of TestMirrors.nim:121 in today's API: 2 beta - alpha1/2... holds trivially here
    assert abs(2 * 0.767 - 0.5 * 2 * a1 - 0.5 * 2 * a2) < 1e-3


def test_parabola_misses_exactly_axis_parallel_ray(lib):
    # raytracer.nim:673: a = 0 for an exactly axis-parallel ray => inf/NaN roots => the input point is returned
    pcb, pxrt, out = v3(246.0, 0.0, -475.0), v3(246.0, 0.0, 0.0), v3(0, 0, 0)
    lib.sart_oracle_find_pos(1, pxrt, pcb, 247.2855, math.radians(0.467), 300.0, 0.0, 7500.0, out)
    assert tuple(out) == (246.0, 0.0, -475.0)


def test_axion_record_layout():
    # `Axion` object raytracer.nim:192-221 => C struct offsets of SURVEY 8(a) a1
    off = {n: getattr(L.Axion, n).offset for n, _ in L.Axion._fields_}
    assert C.sizeof(L.Axion) == 208
    assert (off["passed"], off["passedTillWindow"], off["hitNickel"], off["pointdataX"], off["weights"]) == (0, 1, 2, 8, 48)
    assert (off["kinds"], off["kindsWindow"], off["transProbWindow"], off["shellNumber"], off["reflect"]) == (128, 129, 136, 176, 200)


def test_survival_fractions_babyiaxo():
    # SURVEY Appendix C plausibility numbers (scratch restatement of the survey, 2-3e5 rays, different tables):
    # pipe cuts ~0.547, after glass fronts ~0.329; both are pure geometry of the solar disc + apertures.
    from tests.conftest import make_setup
    full = make_setup("babyiaxo_xmm")
    _, s, _ = O.Oracle(full).trace_histogram(200_000, seed=11)
    assert s["N_REACHED_TELESCOPE"] / 2e5 == pytest.approx(0.547, abs=0.01)
    assert s["N_SHELL_SELECTED"] / 2e5 == pytest.approx(0.329, abs=0.01)
    assert 0.02 < s["N_HIT_NICKEL"] / 2e5 < 0.06


def test_baseline_config0_default_toml_1e5_rays_on_the_cpu_path(tmp_path):
    """BASELINE configs[0]: "config/config_default.toml, 1e5 rays on the Nim CPU path (plumbing, no GPU)".  The keys and values
    of the reference's config_default.toml:1-50 through the TOML layer (config.py), initFullSetup on the DEFAULT table sizes
    (1968 x 1500 CDFs, 1000 x 1000 reflectivity) and 1e5 rays through the CPU oracle — BabyIAXO / InGridIAXO / vacuum / XMM, all
    flags off; survival chain of SURVEY App. C."""
    from solaraxionraytracing_amd import config
    from solaraxionraytracing_amd import _lib as L
    cfgdir = tmp_path / "config"
    cfgdir.mkdir()
    (cfgdir / "config.toml").write_text("""
[Resources]
resourcePath   = "../resources"
outputPath     = "../out"
llnlEfficiency = "llnl_xray_telescope_cast_effective_area_parallel_light_DTU_thesis.csv"
goldFilePrefix = "henke_download/"
rawSolarModel  = "AGSS09_solar_model_stripped.dat"
solarModelFile = "solar_model_dataframe.csv"
llnlReflFile   = "llnl_layer_reflectivities.h5"
goldReflFile   = "gold_0.25microns_reflectivities.h5"
[ReadOpacityFile]
solarModelFile = "solar_model_dataframe.csv"
opcdPath       = "OPCD"
[Setup]
experimentSetup = "BabyIAXO"
detectorSetup   = "InGridIAXO"
stageSetup      = "vacuum"
telescopeSetup  = "XMM"
[Magnet]
useConfig = false
B = 2.0
radiusCB = 350.0
lengthColdbore = 11300.0
lengthB = 11000.0
pGasRoom = 1.0
tGas = 100.0
[TestXraySource]
useConfig = false
active = true
parallel = false
energy = 1.0
distance = 2000.0
radius = 350.0
offAxisUp = 0.0
offAxisLeft = 0.0
activity = 0.125
lengthCol = 0.021
[DetectorInstallation]
useConfig = false
distanceDetectorXRT = 1485.0
distanceWindowFocalPlane = 0.0
lateralShift = 0.0
transversalShift = 0.0
""")
    full = config.init_full_setup_from_config(str(cfgdir / "config.toml"))
    s = full.setup
    assert (s.experiment, s.stage, s.telescope_kind, s.detector_kind) == (L.ES_BABYIAXO, L.SK_VACUUM, L.TK_XMM, L.DK_INGRIDIAXO)
    assert s.magnet_radiusCB == 500.0 and s.distance_detector_xrt == 7500.0      # useConfig = false: the built-in blocks (:1113-1123, :1401-1407)
    assert full.diffFluxCDFs.shape == (1968, 1500) and full.reflectivity.data.shape == (1, 1000, 1000) and full.flags == 0
    n = 100_000
    img, sm, _ = O.Oracle(full).trace_histogram(n, seed=299792458)
    assert sm["N_RAYS"] == n
    assert sm["N_REACHED_TELESCOPE"] / n == pytest.approx(0.547, abs=0.01)       # SURVEY App. C survival chain
    assert sm["N_SHELL_SELECTED"] / n == pytest.approx(0.329, abs=0.01)
    assert sm["N_PASSED"] / n == pytest.approx(0.214, abs=0.01)
    assert img.sum() == pytest.approx(sm["SUM_WEIGHTS"], rel=1e-12) and sm["SUM_WEIGHTS"] > 0


def test_ray_uniforms_are_uniform_and_uncorrelated():
    """The six uniforms of a ray (one Philox4x32-7 block keyed by seed and ray id, cut into 32 / 21 / 22-bit fractions; u3 from
    the word stream that four consecutive rays share) behave like independent U[0,1) samples: moments, lag correlations along the
    ray id (including inside a group of four rays that share a stream block), cross-correlations - among them u0 / u4 and u1 / u4,
    which are cut out of the same two words -, and a chi-square of u3."""
    import ctypes as C
    from oracle import oracle as O
    lib = O.load("f64")
    n = 200_000
    u = np.empty((n, 6))
    buf = (C.c_double * 6)()
    for i in range(n):
        lib.sart_oracle_uniforms(12345, 7_000_000_001 + i, buf)
        u[i] = buf[:]
    assert (u >= 0.0).all() and (u < 1.0).all()
    tol = 5.0 / np.sqrt(n)
    assert np.abs(u.mean(axis=0) - 0.5).max() < tol * 0.29          # sigma of U[0,1) = 0.2887
    assert np.abs(u.var(axis=0) - 1.0 / 12.0).max() < tol * 0.08
    c = np.corrcoef(u.T)
    assert np.abs(c - np.eye(6)).max() < tol                          # between the six uniforms of one ray
    for lag in (1, 2, 3, 4, 5):
        for k in range(6):
            assert abs(np.corrcoef(u[:-lag, k], u[lag:, k])[0, 1]) < tol, (lag, k)
    # rays 4g .. 4g+3 take the four words of one block for the high word of u3: pairwise correlation inside the groups
    g = u[: n // 4 * 4, 3].reshape(-1, 4)
    assert np.abs(np.corrcoef(g.T) - np.eye(4)).max() < 2.0 * tol
    for k in range(6):
        hist, _ = np.histogram(u[:, k], bins=256, range=(0.0, 1.0))
        chi2 = ((hist - n / 256.0) ** 2 / (n / 256.0)).sum()
        assert chi2 < 255.0 + 5.0 * np.sqrt(2.0 * 255.0), (k, chi2)
    # the cut of the 160 bits: 32-bit fractions for the CDF draws and the disc radius, 21 bits for the solar point's angles, 22 for
    # the disc angle (the bits the other two leave over in their words: jointly uniform with each of them)
    for k, bits in ((2, 32), (5, 32), (3, 32), (0, 21), (1, 21), (4, 22)):
        scaled = u[:, k] * 2.0 ** bits
        assert (scaled == np.floor(scaled)).all() and not (u[:, k] * 2.0 ** (bits - 1) == np.floor(u[:, k] * 2.0 ** (bits - 1))).all(), (k, bits)
    for a, b in ((0, 4), (1, 4), (2, 5), (0, 1)):
        h2, _, _ = np.histogram2d(u[:, a], u[:, b], bins=16, range=((0, 1), (0, 1)))
        chi2 = ((h2 - n / 256.0) ** 2 / (n / 256.0)).sum()
        assert chi2 < 255.0 + 5.0 * np.sqrt(2.0 * 255.0), (a, b, chi2)
    # different seeds and far-apart ids give different streams; the same (seed, id) the same numbers
    lib.sart_oracle_uniforms(12345, 7_000_000_001, buf)
    assert list(buf) == list(u[0])
    lib.sart_oracle_uniforms(12346, 7_000_000_001, buf)
    assert all(abs(a - b) > 0 for a, b in zip(buf, u[0]))
