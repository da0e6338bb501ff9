"""Fused axion-mass scan (include/sart.h: sart_trace_mass_scan; BASELINE configs[4]) on the MI355X box.

The reference has one constant axion mass (raytracer.nim:255); in the gas stage the mass enters a ray's weight through
axionConversionProb2 alone (computeMagnetTransmission :1599-1625, axionMassforMagnet.nim:75-113).  The scan kernel traces
every ray once and evaluates that probability for K masses.  Demanded here, per mass:
  * SART_ACCUM_FIXED64: the raw integers equal those of a single-mass launch (sart_set_axion_mass + trace) on the same ray
    ids - bit for bit, for every kernel variant that can run the gas stage, for any split of the rays and of the masses;
  * SART_ACCUM_F64: the flux equals the single-mass launch to 1e-12 (summation order);
  * the flux equals the CPU oracle (80-bit build) with that mass to 1e-6, on small and on full-size AGSS09 tables."""
import os

import ctypes as C

import numpy as np
import pytest

import solaraxionraytracing_amd as sa
from solaraxionraytracing_amd import _lib as L

from tests.conftest import make_setup

pytestmark = pytest.mark.gpu

M_GAMMA = 0.008235          # eV, literal-units gas stage of BabyIAXO
N_IMG = 256 * 256


def masses(k):
    """k masses around the resonance, including it, zero and far-off points (two groups of masses when k > 32)."""
    m = np.concatenate([[0.0, M_GAMMA], np.linspace(0.002, 0.02, max(0, k - 3)), [0.05]])
    return np.ascontiguousarray(m[:k])


class env:
    def __init__(self, **kv):
        self.kv = kv

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kv}
        os.environ.update(self.kv)

    def __exit__(self, *exc):
        for k, v in self.old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v


def raw_single(rt, torch, m, n, seed, off=0, flags=None):
    """Raw FIXED64 accumulator of one single-mass launch."""
    acc = torch.zeros(sa.accumulator_len(256), dtype=torch.int64, device="cuda")
    rt.set_axion_mass(float(m))
    p = rt.trace_params(n, seed=seed, ray_id_offset=off, flags=flags, accumulate=False)
    rt.trace_histogram_device(p, acc.data_ptr())
    rt.synchronize()
    return acc.cpu().numpy()[N_IMG:]


def raw_scan(rt, torch, ms, pieces, seed, flags=None):
    acc = torch.zeros(sa.mass_scan_len(len(ms)), dtype=torch.int64, device="cuda")
    for lo, hi in pieces:
        p = rt.trace_params(hi - lo, seed=seed, ray_id_offset=lo, flags=flags, accumulate=True)
        rt.trace_mass_scan_device(p, ms, acc.data_ptr())
    rt.synchronize()
    return acc.cpu().numpy().reshape(len(ms) + 1, L.SCAN_ROW)


VARIANTS = {
    # kernel variant 6: specialised gas stage, constant path (stage A0 on, scan accumulators in ring 1's path column)
    "gas_pathc": ("babyiaxo_xmm_gas", {}, None),
    # variant 3: the path travels through ring 1, scan accumulators in ring 0 (stage A0 off for the scan)
    "gas_ring_path": ("babyiaxo_xmm_gas", {"SART_NO_PATH_CONST": "1"}, None),
    # variant 1: generic instantiation (stage read at run time)
    "generic": ("babyiaxo_xmm_gas", {"SART_FORCE_GENERIC": "1"}, None),
    # variant 2: generic, rotated telescope
    "generic_rotated": ("babyiaxo_xmm_gas", {}, "rot"),
    # variant 1 with the X-ray test source (one energy row, no exposure factor)
    "xray_test_source": ("babyiaxo_xmm_gas", {}, "xray"),
    # cone optics, four coatings, a 43 mm bore that rays do enter through the wall: stage A0 has no zones, the path varies per ray
    "cast_llnl_gas": ("cast_llnl", {}, "gas"),
    # the third telescope: CAST + Abrixas (27 Wolter shells, six spokes)
    "cast_abrixas_gas": ("cast_abrixas", {}, "gas"),
}


def variant_setup(name):
    setup_name, knobs, tweak = VARIANTS[name]
    flags = None
    if tweak == "xray":
        full = sa.initFullSetup(stage=L.SK_GAS, flags=L.CF_XRAY_TEST, n_radii=400, n_energies=300, refl_n_angles=200, refl_n_energies=200)
    else:
        full = make_setup(setup_name)
    if tweak == "rot":
        full.setup.telescope_turned_x_deg, full.setup.telescope_turned_y_deg = 0.01, 0.03
    if tweak == "gas":
        full.setup.stage = L.SK_GAS
    return full, knobs, flags


@pytest.mark.parametrize("name", sorted(VARIANTS))
def test_fixed64_scan_equals_single_mass_launches_bit_for_bit(name):
    import torch
    full, knobs, flags = variant_setup(name)
    ms = masses(37)                      # two groups of masses: 32 + 5
    n, seed = 3_000_000, 17
    with env(**knobs):
        with sa.RayTracer(full) as rt:
            rt.set_accumulation_mode("fixed64")
            scan = raw_scan(rt, torch, ms, [(0, n)], seed, flags)
            split = raw_scan(rt, torch, ms, [(0, 1_000_001), (1_000_001, n)], seed, flags)   # rays in two accumulating calls
            singles = [raw_single(rt, torch, m, n, seed, flags=flags) for m in ms]
    assert np.array_equal(scan, split)
    shared = scan[len(ms)]
    assert shared[L.SCAN_SHARED["N_RAYS"]] == n
    assert scan[:, L.SCAN["N_PASSED"]][:len(ms)].min() > 1000
    for k, s in enumerate(singles):
        row = scan[k]
        assert row[L.SCAN["SUM_WEIGHTS"]] == s[L.ACC["SUM_WEIGHTS"]] and row[L.SCAN_HI["SUM_WEIGHTS"]] == s[L.ACC_HI["SUM_WEIGHTS"]], (name, k)
        assert row[L.SCAN["SUM_WEIGHTS_SQ"]] == s[L.ACC["SUM_WEIGHTS_SQ"]] and row[L.SCAN_HI["SUM_WEIGHTS_SQ"]] == s[L.ACC_HI["SUM_WEIGHTS_SQ"]], (name, k)
        assert row[L.SCAN["N_PASSED"]] == s[L.ACC["N_PASSED"]], (name, k)
        for key in ("N_REACHED_TELESCOPE", "N_SHELL_SELECTED", "N_HIT_NICKEL"):
            assert shared[L.SCAN_SHARED[key]] == s[L.ACC[key]], (name, key)
        assert 0 <= row[L.SCAN["SUM_WEIGHTS"]] < 2 ** L.FIXED_LIMB_BITS


@pytest.mark.parametrize("flags", [L.CF_IGNORE_DET_WINDOW | L.CF_IGNORE_GAS_ABS, L.CF_IGNORE_REFLECTION, L.CF_IGNORE_DET_WINDOW | L.CF_IGNORE_REFLECTION | L.CF_IGNORE_GAS_ABS])
def test_fixed64_scan_equals_single_mass_launches_under_the_ignore_flags(flags):
    """The command-line switches of the reference (--ignoreDetWindow, --ignoreGasAbs, --ignoreReflection; raytracer.nim:2842-2849)
    change which factors the mass-independent part of the weight has (and the FIXED64 weight bound): the scan still equals the
    single-mass launches bit for bit."""
    import torch
    full = make_setup("babyiaxo_xmm_gas")
    ms = masses(6)
    n, seed = 1_000_000, 23
    with sa.RayTracer(full) as rt:
        rt.set_accumulation_mode("fixed64")
        scan = raw_scan(rt, torch, ms, [(0, n)], seed, flags)
        for k, m in enumerate(ms):
            s = raw_single(rt, torch, m, n, seed, flags=flags)
            for a, b in (("SUM_WEIGHTS", "SUM_WEIGHTS"), ("SUM_WEIGHTS_SQ", "SUM_WEIGHTS_SQ")):
                assert scan[k][L.SCAN[a]] == s[L.ACC[b]] and scan[k][L.SCAN_HI[a]] == s[L.ACC_HI[b]], (flags, k, a)
            assert scan[k][L.SCAN["N_PASSED"]] == s[L.ACC["N_PASSED"]] > 0


def test_fixed64_scan_finalize_equals_finalized_single_launches():
    """The per-mass quanta are a function of (setup, tables, flags, headroom, mass): finalize of the raw scan gives the very
    doubles the blocking single-mass call returns, and the blocking scan call returns them, too."""
    full = make_setup("babyiaxo_xmm_gas")
    ms = masses(7)
    n, seed = 2_000_000, 3
    with sa.RayTracer(full) as rt:
        rt.set_accumulation_mode("fixed64")
        per_mass, shared = rt.trace_mass_scan(ms, n, seed=seed)
        for k, m in enumerate(ms):
            rt.set_axion_mass(float(m))
            s = rt.trace_histogram(n, seed=seed)[1]
            for key in ("SUM_WEIGHTS", "SUM_WEIGHTS_SQ", "N_PASSED"):
                assert np.float64(per_mass[key][k]).view(np.uint64) == np.float64(s[key]).view(np.uint64), (k, key, per_mass[key][k], s[key])
        assert shared["N_RAYS"] == n and shared["N_ON_DETECTOR"] >= per_mass["N_PASSED"].max()
        # the far-off-resonance point keeps its resolution: its quantum follows the mass
        assert per_mass["SUM_WEIGHTS"][-1] < 0.05 * per_mass["SUM_WEIGHTS"][1]
        assert np.all(np.isfinite(per_mass["SUM_WEIGHTS_SQ"]))


@pytest.mark.parametrize("name", ["gas_pathc", "gas_ring_path", "generic_rotated"])
def test_f64_scan_equals_single_mass_launches(name):
    full, knobs, flags = variant_setup(name)
    ms = masses(35)
    n, seed = 3_000_000, 5
    with env(**knobs):
        with sa.RayTracer(full) as rt:
            per_mass, shared = rt.trace_mass_scan(ms, n, seed=seed, flags=flags)
            loop = sa.performAxionMassScanHostLoop(rt, ms, n, seed=seed, flags=flags)
            n_pass = []
            for m in ms:
                rt.set_axion_mass(float(m))
                s = rt.trace_histogram(n, seed=seed, flags=flags)[1]
                n_pass.append(s["N_PASSED"])
                assert per_mass["SUM_WEIGHTS_SQ"][len(n_pass) - 1] == pytest.approx(s["SUM_WEIGHTS_SQ"], rel=1e-11)
    assert np.array_equal(per_mass["N_PASSED"], np.array(n_pass))
    assert np.abs(per_mass["SUM_WEIGHTS"] / loop - 1.0).max() < 1e-12


@pytest.mark.parametrize("tables", ["small", "small_xray"])
def test_scan_matches_the_oracle_per_mass(tables):
    from oracle.oracle import Oracle
    full = variant_setup("xray_test_source")[0] if tables == "small_xray" else make_setup("babyiaxo_xmm_gas")
    ms = masses(9)
    n, seed = 100_000, 4
    with sa.RayTracer(full) as rt:
        per_mass, shared = rt.trace_mass_scan(ms, n, seed=seed)
    o = Oracle(full, "ld")
    for k, m in enumerate(ms):
        s = full.setup.copy()
        s.m_axion = float(m)
        want = o.trace_histogram(n, seed=seed, setup=s)[1]
        # (a ray within rounding of a cut may fall on the other side in the 80-bit oracle: then the flux differs by that ray)
        same_rays = per_mass["N_PASSED"][k] == want["N_PASSED"]
        assert abs(per_mass["N_PASSED"][k] - want["N_PASSED"]) <= 2
        assert per_mass["SUM_WEIGHTS"][k] == pytest.approx(want["SUM_WEIGHTS"], rel=1e-6 if same_rays else 1e-4), (k, m)
        for key in ("N_REACHED_TELESCOPE", "N_SHELL_SELECTED", "N_HIT_NICKEL"):
            assert abs(shared[key] - want[key]) <= 2


def test_scan_on_full_size_agss09_tables_matches_the_oracle_and_the_host_loop():
    """BASELINE configs[4] at its table sizes: full AGSS09 emission (all terms, made on the device), 1968 x 1500 CDFs, 1000 x 1000
    reflectivity; 40 masses (two groups: 32 + 8), the scan through the C++ host driver."""
    from oracle.oracle import Oracle
    full = sa.initFullSetup(stage=L.SK_GAS, emission="agss09-device")
    ms = np.linspace(0.0, 0.02, 40)
    n, seed = 4_000_000, 11
    with sa.RayTracer(full) as rt:
        flux, err, n_pass = sa.performAxionMassScan(rt, ms, n, seed=seed, errors=True)
        loop = sa.performAxionMassScanHostLoop(rt, ms[[0, 13, 39]], n, seed=seed)
        full.fetch_solar_tables(rt)
    assert np.abs(flux[[0, 13, 39]] / loop - 1.0).max() < 1e-12
    assert np.all(err > 0) and np.all(err < 0.05 * flux) and n_pass.min() > 1e5
    k_res = int(np.argmin(np.abs(ms - M_GAMMA)))
    assert int(np.argmax(flux)) == k_res                         # the resonance m_a = m_gamma
    o = Oracle(full, "ld")
    n_o = 400_000
    with sa.RayTracer(full) as rt:
        per_mass, _ = rt.trace_mass_scan(ms[[0, k_res, 39]], n_o, seed=seed)
    for j, k in enumerate((0, k_res, 39)):
        s = full.setup.copy()
        s.m_axion = float(ms[k])
        want = o.trace_histogram(n_o, seed=seed, setup=s)[1]["SUM_WEIGHTS"]
        assert per_mass["SUM_WEIGHTS"][j] == pytest.approx(want, rel=1e-6), k


def test_scan_flags_accumulation_and_errors():
    full = make_setup("babyiaxo_xmm_gas")
    ms = masses(5)
    with sa.RayTracer(full) as rt:
        # ignoreConvProb: the mass no longer matters
        # (f64: every mass adds the same weights, but the sixteen waves of a workgroup reach a mass's LDS cells in an order of their
        # own per mass - equal up to the summation order, 1e-15 per DESIGN.md 3.1; the demand for identical bits held by luck until
        # the round-6 stream shifted the timing.  FIXED64: integers - identical to the last bit)
        pm, _ = rt.trace_mass_scan(ms, 500_000, seed=2, flags=L.CF_IGNORE_CONV_PROB)
        assert pm["SUM_WEIGHTS"][0] > 0
        np.testing.assert_allclose(pm["SUM_WEIGHTS"], pm["SUM_WEIGHTS"][0], rtol=1e-13)
        assert np.all(pm["N_PASSED"] == pm["N_PASSED"][0])
        want = rt.trace_histogram(500_000, seed=2, flags=L.CF_IGNORE_CONV_PROB)[1]
        assert pm["SUM_WEIGHTS"][0] == pytest.approx(want["SUM_WEIGHTS"], rel=1e-12) and pm["N_PASSED"][0] == want["N_PASSED"]
        rt.set_accumulation_mode("fixed64")
        pmx, _ = rt.trace_mass_scan(ms, 500_000, seed=2, flags=L.CF_IGNORE_CONV_PROB)
        assert np.all(pmx["SUM_WEIGHTS"] == pmx["SUM_WEIGHTS"][0]) and np.all(pmx["SUM_WEIGHTS_SQ"] == pmx["SUM_WEIGHTS_SQ"][0])
        assert pmx["SUM_WEIGHTS"][0] == pytest.approx(want["SUM_WEIGHTS"], rel=1e-9)
        rt.set_accumulation_mode("f64")
        # no rays: zeros
        pm, sh = rt.trace_mass_scan(ms, 0, seed=2)
        assert not pm["SUM_WEIGHTS"].any() and sh["N_RAYS"] == 0
        # the context's own mass is untouched by a scan
        assert rt.full.setup.m_axion == full.setup.m_axion
        with pytest.raises(L.SartError) as e:
            rt.trace_mass_scan(np.array([0.01, -1.0]), 1000)
        assert e.value.code == -1
        with pytest.raises(L.SartError):
            rt.trace_mass_scan(np.array([np.nan]), 1000)
    # vacuum stage: the entry point refuses (nothing depends on m_a), the host driver serves every mass from one launch
    vac = make_setup("babyiaxo_xmm")
    with sa.RayTracer(vac) as rt:
        with pytest.raises(L.SartError) as e:
            rt.trace_mass_scan(ms, 1000)
        assert e.value.code == -1 and "vacuum" in str(e.value)
        flux = sa.performAxionMassScan(rt, ms, 300_000, seed=2)
        assert np.all(flux == flux[0]) and flux[0] == rt.trace_histogram(300_000, seed=2)[1]["SUM_WEIGHTS"]


def test_scan_cli_writes_the_curve(tmp_path):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / "mass.csv")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "scan.py"), "mass", "--points", "9", "--rays", "1e6", "--massMin", "0.004",
                        "--massMax", "0.012", "--emission", "primakoff", "--out", out], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    rows = [l.split(",") for l in open(out).read().splitlines()]
    assert rows[0][:3] == ["m_a [eV]", "flux", "relative flux"] and len(rows) == 10
    rel = np.array([float(x[2]) for x in rows[1:]])
    assert rel.max() == 1.0 and int(np.argmax(rel)) in (4, 5)     # 0.008 / 0.009 eV bracket m_gamma = 0.008235 eV


def test_host_loop_driver_keeps_its_own_ray_block_per_mass():
    """sart_host_perform_axion_mass_scan had kept its name and signature while changing meaning (fused: every mass on
    the same rays).  It is the host loop of rounds 1-3 again - mass i on the ray ids [offset + i n, offset + (i + 1) n), flux-only
    launches, the context's mass back afterwards; the fused scan lives under sart_host_axion_mass_scan alone."""
    full = make_setup("babyiaxo_xmm_gas")
    ms = masses(4)
    n, seed, off = 400_000, 9, 5000
    with sa.RayTracer(full) as rt:
        loop = sa.performAxionMassScanHostLoop(rt, ms, n, seed=seed, ray_id_offset=off, same_rays=False)
        for i, m in enumerate(ms):
            rt.set_axion_mass(float(m))
            want = rt.trace_histogram(n, seed=seed, ray_id_offset=off + i * n)[1]["SUM_WEIGHTS"]
            assert loop[i] == pytest.approx(want, rel=1e-12), i
        rt.set_axion_mass(full.setup.m_axion)
        fused = sa.performAxionMassScan(rt, ms, n, seed=seed, ray_id_offset=off)
        s = L.Setup()
        L.check(rt.lib.sart_get_setup(rt.handle, C.byref(s)))
        assert s.m_axion == full.setup.m_axion
    assert fused[0] == pytest.approx(loop[0], rel=1e-12)          # mass 0 sits on the same block in both
    assert abs(fused[1] / loop[1] - 1.0) > 1e-9                    # the others are other rays: close, not equal
    assert abs(fused[1] / loop[1] - 1.0) < 0.05
