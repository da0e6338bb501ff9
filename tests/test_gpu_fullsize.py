"""Ray-for-ray parity at BASELINE's own table sizes (run on the MI355X box).

tests/test_gpu_parity.py compares on shrunken tables (400 radii x 300 energies, 200 x 200 reflectivity) so that dozens of
setups stay cheap.  Here the tables are the ones `initFullSetup()` builds by default and `bench.py` runs on — 1968 x 1500
emission CDFs behind the log-spaced energy guide, 1000 x 1000 reflectivity re-tabulated to 1501 x 1000 per coating — for
the four BASELINE configurations that reach the GPU:

  configs[1]  CAST magnet + LLNL telescope, gold reflectivities
              ... and the two pairings the reference itself ships for that magnet: LLNL with its four multilayer coatings by
              shell group (raytracer.nim:1164-1187, computeReflectivity :1571-1580) and Abrixas (27 shells, :1320-1346)
  configs[2]  BabyIAXO magnet + XMM-Newton shells, vacuum
  configs[3]  XMM shells, telescope turned (one angle bin of the effective-area scan: chip 100 mm, effective-area flags)
  configs[4]  full AGSS09 emission (all terms, made by the emission kernel) + gas stage

Records are compared with the binary128 build of the oracle (every decision of every ray identical, positions to 1e-10 mm),
the fused histogram with the f64 oracle at 2e7 rays (counters to a few rays: the rays inside the f64 formulation's own
rounding noise of a cut edge, tests/test_gpu_parity.py::test_reference_formulation_noise_envelope).
"""
import numpy as np
import pytest

import solaraxionraytracing_amd as sa
from solaraxionraytracing_amd import _lib as L

pytestmark = pytest.mark.gpu

N_REC = 30_000
N_HIST = 20_000_000
EFFAREA_FLAGS = L.CF_IGNORE_DET_WINDOW | L.CF_IGNORE_GAS_ABS | L.CF_IGNORE_CONV_PROB

_cache = {}


def full_setup(name):
    """Default (full-size) tables; built once per session (the AGSS09 table runs the emission kernel)."""
    if name not in _cache:
        if name == "babyiaxo_xmm":                      # configs[2]
            full = sa.initFullSetup()
        elif name == "cast_llnl_gold":                  # configs[1]
            full = sa.initFullSetup(L.ES_CAST, L.DK_INGRID2018, L.SK_VACUUM, L.TK_LLNL, reflectivity="gold")
        elif name == "cast_llnl":                       # the reference's own LLNL pairing: 4 coatings, 48 MB re-tabulated
            full = sa.initFullSetup(L.ES_CAST, L.DK_INGRID2018, L.SK_VACUUM, L.TK_LLNL)
            assert full.reflectivity.data.shape[0] == 4 and full.setup.n_coatings == 4
        elif name == "cast_abrixas":                    # raytracer.nim:1320-1346
            full = sa.initFullSetup(L.ES_CAST, L.DK_INGRID2017, L.SK_VACUUM, L.TK_ABRIXAS)
            assert full.setup.n_shells == 27
        elif name == "babyiaxo_xmm_gas_agss09":         # configs[4]
            full = sa.initFullSetup(stage=L.SK_GAS, emission="agss09")
        elif name == "babyiaxo_xmm_rot_effarea":        # configs[3]
            full = sa.initFullSetup()
            full.setup.telescope_turned_x_deg = 0.02
            full.setup.telescope_turned_y_deg = 0.1
            full.setup.chip_x_max = full.setup.chip_y_max = 100.0
            full.flags = EFFAREA_FLAGS
        else:
            raise KeyError(name)
        assert full.diffFluxCDFs.shape == (1968, 1500) and full.reflectivity.data.shape[1:] == (1000, 1000)
        _cache[name] = full
    return _cache[name]


NAMES = ["babyiaxo_xmm", "cast_llnl_gold", "cast_llnl", "cast_abrixas", "babyiaxo_xmm_gas_agss09", "babyiaxo_xmm_rot_effarea"]


@pytest.mark.parametrize("name", NAMES)
def test_fullsize_records_match_binary128_oracle(name):
    from oracle.oracle import Oracle
    full = full_setup(name)
    with sa.RayTracer(full) as rt:
        rec = rt.traceAxionWrapper(N_REC, seed=4242)
    ref = Oracle(full, "q").trace_records(N_REC, seed=4242)
    for f in ("passed", "passedTillWindow", "hitNickel", "shellNumber", "kinds", "kindsWindow"):
        np.testing.assert_array_equal(rec[f], ref[f], err_msg=f)
    # the energy draw goes through the guide table + candidate gather on the device and through lowerBound in the oracle
    np.testing.assert_array_equal(rec["energiesPre"], ref["energiesPre"])
    np.testing.assert_array_equal(rec["energiesAx"], ref["energiesAx"])
    assert rec["passed"].sum() > 0.15 * N_REC
    m = rec["passedTillWindow"] == 1
    for f in ("pointdataX", "pointdataY", "pointdataR", "pointdataXBefore", "pointdataYBefore", "deviationDet", "yawAngles"):
        assert np.abs(rec[f] - ref[f]).max(initial=0.0) < 1e-10, f
    for f in ("weights", "reflect", "transmissionMagnet", "transProbWindow"):
        np.testing.assert_allclose(rec[f][m], ref[f][m], rtol=2e-8, atol=0, err_msg=f)


@pytest.mark.parametrize("name", NAMES)
def test_fullsize_histogram_matches_f64_oracle_to_a_few_rays(name):
    from bench import available_cpus
    from oracle.oracle import Oracle
    full = full_setup(name)
    with sa.RayTracer(full) as rt:
        img, s = rt.trace_histogram(N_HIST, seed=99)
    oimg, o, _ = Oracle(full).trace_histogram(N_HIST, seed=99, n_threads=available_cpus())
    assert s["N_RAYS"] == N_HIST == o["N_RAYS"]
    bound = max(10.0, 1.5e-6 * N_HIST)     # measured at 1e8 / 1e9 rays: <= 4.2e-7 N (profiles/r01_full_size_compare_*.json)
    for k in ("N_REACHED_TELESCOPE", "N_SHELL_SELECTED", "N_HIT_NICKEL", "N_PASSED_TILL_WINDOW", "N_PASSED"):
        assert abs(s[k] - o[k]) <= bound, (k, s[k], o[k])
    assert s["SUM_WEIGHTS"] == pytest.approx(o["SUM_WEIGHTS"], rel=2e-5)
    for k in ("SUM_X", "SUM_Y", "SUM_R", "SUM_WEIGHTS_SQ"):
        assert s[k] == pytest.approx(o[k], rel=2e-5), k
    assert s["N_OUTSIDE_IMAGE"] == o["N_OUTSIDE_IMAGE"] == 0
    assert img.sum() == pytest.approx(s["SUM_WEIGHTS"], rel=1e-11)
    # image: pixel-edge flips of rays inside the oracle's noise (0.055 mm pixels, ~1e-3 mm noise => a fraction of a few 1e-3
    # of the rays lands in the neighbouring pixel; at ~400 rays per pixel few of those flips cancel, at 1e9 rays most do:
    # L1 = 9e-4 there, profiles/r01_full_size_compare_1e9.json)
    assert float(np.abs(img - oimg).sum() / oimg.sum()) <= 1e-2
    c, oc = img.reshape(32, 8, 32, 8).sum(axis=(1, 3)), oimg.reshape(32, 8, 32, 8).sum(axis=(1, 3))
    assert float(np.abs(c - oc).max() / oc.max()) <= 5e-4     # 8 x 8-pixel blocks: 2.6e-5 at 1e9 rays, ~1e-4 at 2e7


def test_fullsize_golden_babyiaxo_xmm():
    """tests/golden/babyiaxo_xmm_full.npz: 4000 records + a 1e5-ray histogram of the f64 oracle on the default tables."""
    from tests.test_golden import FLAG_MISMATCH_MAX, POS_TOL_MM, WEIGHT_RTOL, _check_inputs, _load, compare_records
    g, n_rec, n_hist, seed, flags = _load("babyiaxo_xmm_full")
    full = full_setup("babyiaxo_xmm")
    _check_inputs(full, g)
    with sa.RayTracer(full) as rt:
        rec = rt.traceAxionWrapper(n_rec, seed=seed)
        img, summ = rt.trace_histogram(n_hist, seed=seed)
    compare_records(rec, g, POS_TOL_MM, WEIGHT_RTOL, FLAG_MISMATCH_MAX)
    gs = dict(zip([str(k) for k in g["summary_keys"]], g["summary_vals"]))
    for k in ("N_REACHED_TELESCOPE", "N_SHELL_SELECTED", "N_PASSED", "N_PASSED_TILL_WINDOW", "N_HIT_NICKEL"):
        assert abs(summ[k] - gs[k]) <= 2.0, (k, summ[k], gs[k])
    assert summ["SUM_WEIGHTS"] == pytest.approx(gs["SUM_WEIGHTS"], rel=1e-4)


def test_energy_draw_over_the_whole_guide_matches_lower_bound():
    """Every bucket kind of the energy guide (uniform, log-spaced towards u = 1, the widest buckets that fall back to the
    binary search): 4e5 rays on a table whose CDFs creep towards 1 (flat tail), energies compared exactly."""
    from oracle.oracle import Oracle
    from solaraxionraytracing_amd import tables
    radii, energies = tables.solar_grid(1968, 1500)
    em = tables.primakoff_emission_table(1968, 1500).copy()
    em[:, 900:] *= 1e-9                                     # 600 energies share < 1e-8 of every row's probability
    em[:, :40] *= 1e-7                                      # and 40 more sit below the first uniform bucket
    full = sa.initFullSetup(emission=em)
    n = 400_000
    with sa.RayTracer(full) as rt:
        rec = rt.traceAxionWrapper(n, seed=31)
    ref = Oracle(full).trace_records(n, seed=31)
    np.testing.assert_array_equal(rec["energiesPre"], ref["energiesPre"])


def test_energy_draw_on_the_cdf_entries_themselves():
    """The four-candidate count of the energy draw compares the upper 32 bits of floor(cdf 2^52) with those of the uniform and lets
    the f64 row decide ties (sart_kernels.hip: energy_draw_finish).  Here the uniforms ARE table entries and their f64
    neighbours - every draw is a tie in the upper bits - fed through the explicit-uniform test entry; the energy index must be
    std/algorithm.lowerBound's (raytracer.nim:464-468) for every one of them."""
    full = full_setup("babyiaxo_xmm")
    rcdf, ecdf, energies = full.fluxRadiusCDF, full.diffFluxCDFs, full.energies
    rng = np.random.default_rng(11)
    n = 30_000
    rows = rng.integers(1, 700, n)                              # radius rows that carry probability
    u2 = 0.5 * (rcdf[rows - 1] + rcdf[rows])                    # lowerBound(rcdf, u2) == rows
    assert np.array_equal(np.searchsorted(rcdf, u2, side="left"), rows)
    cols = rng.integers(0, energies.size - 1, n)
    entry = ecdf[rows, cols]
    kind = rng.integers(0, 5, n)
    u5 = np.where(kind == 0, entry, np.where(kind == 1, np.nextafter(entry, 0.0), np.where(kind == 2, np.nextafter(entry, 1.0),
                  np.where(kind == 3, entry * (1 - 2.0 ** -40), entry * (1 + 2.0 ** -40)))))
    u5 = np.clip(u5, 0.0, np.nextafter(1.0, 0.0))
    u = np.column_stack([np.full(n, 0.3), np.full(n, 0.4), u2, np.full(n, 1e-4), np.full(n, 0.7), u5])
    with sa.RayTracer(full) as rt:
        rec = rt.trace_records_uniforms(u)
    want_idx = np.array([min(np.searchsorted(ecdf[r], x, side="left"), energies.size - 1) for r, x in zip(rows, u5)])
    want = np.maximum(0.03, energies[want_idx])                 # :470-471
    np.testing.assert_array_equal(rec["energiesPre"], want)
    assert len(np.unique(want_idx)) > 1000                      # the whole energy range, not one corner of it


def test_radius_draw_on_the_cdf_entries_themselves():
    """The radius draw compares 32-bit words in LDS (T' = floor(cdf 2^32) + 1 against K = floor(u2 2^32)) inside a guide bracket -
    2048 buckets of 1/2048 and 1024 of 1/32768 for u2 >= 31/32 -, searches on in a bracket wider than four entries and lets the
    f64 table in device memory decide ties (sart_kernels.hip: phase_a_core).  Here the uniforms ARE table entries and their f64
    neighbours - every draw a tie in the upper bits -, drawn from the whole table: the crowded end near 1 (buckets of hundreds of
    entries), the first buckets, the bulk.  Fed through the explicit-uniform entry to the GPU and to the CPU oracle: the sampled
    radius sets where the ray starts on the Sun and which CDF row its energy comes from, so a wrong index moves the ray."""
    from oracle.oracle import Oracle
    full = full_setup("babyiaxo_xmm")
    rcdf = full.fluxRadiusCDF
    rng = np.random.default_rng(23)
    n = 40_000
    idx = np.concatenate([rng.integers(0, rcdf.size, n // 2), rng.integers(rcdf.size - 700, rcdf.size, n // 4), rng.integers(0, 40, n // 4)])
    entry = rcdf[idx]
    kind = rng.integers(0, 6, n)
    u2 = np.choose(kind, [entry, np.nextafter(entry, 0.0), np.nextafter(entry, 1.0), entry * (1 - 2.0 ** -40), entry * (1 + 2.0 ** -40),
                          entry - 2.0 ** -33])
    u2 = np.clip(u2, 0.0, np.nextafter(1.0, 0.0))
    want = np.minimum(np.searchsorted(rcdf, u2, side="left"), rcdf.size - 1)
    assert len(np.unique(want)) > 1500 and (want > rcdf.size - 600).sum() > 5000 and (want < 20).sum() > 2000
    u = np.column_stack([rng.random(n), rng.random(n), u2, rng.random(n), rng.random(n), rng.random(n)])
    with sa.RayTracer(full) as rt:
        rec = rt.trace_records_uniforms(u)
    ref = Oracle(full, "f64").trace_records_uniforms(u, n_threads=8)
    # the energy comes from row `want` of diffFluxCDFs with the same u5 on both sides
    e_want = np.maximum(0.03, full.energies[[min(np.searchsorted(full.diffFluxCDFs[r], x, side="left"), full.energies.size - 1)
                                              for r, x in zip(want, u[:, 5])]])
    np.testing.assert_array_equal(rec["energiesPre"], e_want)
    np.testing.assert_array_equal(rec["energiesPre"], ref["energiesPre"])
    for k in ("passed", "passedTillWindow", "hitNickel", "shellNumber"):
        assert (rec[k] != ref[k]).sum() <= 2, k                      # (the f64 oracle's own rounding noise at a cut edge)
    both = (rec["passedTillWindow"] != 0) & (ref["passedTillWindow"] != 0)
    assert both.sum() > 1000
    np.testing.assert_allclose(rec["pointdataXBefore"][both], ref["pointdataXBefore"][both], atol=2e-3)   # a wrong radius index moves the spot by more
