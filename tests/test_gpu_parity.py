"""GPU parity tests (run on the MI355X box): the HIP path, called through the C-ABI, against the CPU oracle
on the same seeded inputs, plus size-independent properties at full BASELINE sizes.

Two oracles are used (oracle/sart_oracle.c built twice from one source):
 * f64  — the literal restatement of the reference.  Its formulation carries ~1e-5 mm cancellation noise at the
          bore exit (points of magnitude 1e11..1.5e14 mm are subtracted), which becomes up to ~1.5e-3 mm in the
          focal plane.  HIP-vs-f64 differences are bounded by that envelope (tolerances below).
 * ld   — the same source in 80-bit long double, 2048x less of that noise.  HIP-vs-ld shows that what remains is
          not an algorithmic difference: positions agree to 5e-6 mm, weights to 1e-6 relative.
 * q    — the same source in IEEE binary128: HIP agrees to 1e-10 mm / 2e-8 relative with identical decisions for every
          ray, i.e. the HIP path evaluates the reference's algorithm more accurately than its own f64 formulation does.
"""
import ctypes as C

import numpy as np
import pytest

import solaraxionraytracing_amd as sa
from solaraxionraytracing_amd import _lib as L
from tests.conftest import SETUP_NAMES, make_setup
from tests.test_golden import FLAG_MISMATCH_MAX, POS_TOL_MM, WEIGHT_RTOL, compare_records

pytestmark = pytest.mark.gpu

N_REC = 60_000


def _small():
    from tests.conftest import SMALL
    return dict(SMALL)


def _as_gold(rec):
    return {"rec_" + n: rec[n] for n in rec.dtype.names}


def test_reference_formulation_noise_envelope():
    """Measures the envelope the tolerances rest on: f64 oracle vs its own long-double build (no GPU involved)."""
    from oracle.oracle import Oracle
    full = make_setup("babyiaxo_xmm")
    a = Oracle(full, "f64").trace_records(N_REC, seed=5)
    b = Oracle(full, "ld").trace_records(N_REC, seed=5)
    both = (a["passed"] == 1) & (b["passed"] == 1)
    dx = np.abs(a["pointdataX"][both] - b["pointdataX"][both])
    assert 1e-5 < dx.max() < POS_TOL_MM          # the noise exists, and the tolerance covers it
    assert dx.std() > 1e-5


@pytest.mark.parametrize("name", SETUP_NAMES)
def test_records_match_f64_oracle_within_its_noise(name):
    from oracle.oracle import Oracle
    full = make_setup(name)
    with sa.RayTracer(full) as rt:
        rec = rt.traceAxionWrapper(N_REC, seed=42)
    ref = Oracle(full, "f64").trace_records(N_REC, seed=42)
    compare_records(rec, _as_gold(ref), POS_TOL_MM, WEIGHT_RTOL, FLAG_MISMATCH_MAX, pos_outliers=2e-5)


@pytest.mark.parametrize("name", SETUP_NAMES)
def test_records_match_long_double_oracle_tightly(name):
    from oracle.oracle import Oracle
    full = make_setup(name)
    with sa.RayTracer(full) as rt:
        rec = rt.traceAxionWrapper(N_REC, seed=43)
    ref = Oracle(full, "ld").trace_records(N_REC, seed=43)
    # 5e-6 mm, with room for the 80-bit build's own rare outliers (its formulation is the reference's: differences of 1e14-mm
    # points; one ray in ~1e5 lands at 1e-5 mm - Abrixas, seed 43 of the round-6 stream): at most one ray per 5e4 up to 5e-5 mm,
    # and every such ray must agree with the binary128 build to 1e-10 mm - the noise is the oracle's, not the path's
    compare_records(rec, _as_gold(ref), 5e-6, 1e-6, 5e-5, pos_outliers=2e-5)
    both = (rec["passed"] == 1) & (ref["passed"] == 1)
    off = np.zeros(len(rec), dtype=bool)
    for f in ("pointdataX", "pointdataY", "pointdataR", "pointdataXBefore", "pointdataYBefore"):
        off |= both & (np.abs(rec[f] - ref[f]) >= 5e-6)
    if off.any():
        exact = Oracle(full, "q").trace_records(N_REC, seed=43)
        for f in ("pointdataX", "pointdataY", "pointdataR", "pointdataXBefore", "pointdataYBefore"):
            assert np.abs(rec[f][off] - exact[f][off]).max() < 1e-10, (f, np.flatnonzero(off))
    # every field of every record, including rays that did not pass (zero-initialised like newSeq[Axion])
    dead = (rec["passed"] == 0) & (ref["passed"] == 0)
    # (a ray may reach the end with weight == 0, e.g. behind the strongback: fields set, `passed` false)
    for f in ("pointdataX", "pointdataY", "pointdataR"):
        assert np.abs(rec[f][dead] - ref[f][dead]).max(initial=0.0) < 5e-6
    for f in ("weights", "weightsAll", "energiesAx", "shellNumber"):
        np.testing.assert_allclose(rec[f][dead].astype(float), ref[f][dead].astype(float), rtol=1e-6, atol=0)
    early = (rec["passedTillWindow"] == 0) & (ref["passedTillWindow"] == 0)
    for f in ("pointdataX", "pointdataY", "weights", "weightsAll", "energiesAx", "shellNumber", "pointdataR"):
        assert not rec[f][early].any()
    np.testing.assert_array_equal(rec["kinds"], ref["kinds"])
    np.testing.assert_allclose(rec["transProbArgon"], ref["transProbArgon"], rtol=1e-13)
    np.testing.assert_array_equal(rec["pixvalsX"], ref["pixvalsX"])


@pytest.mark.parametrize("name", SETUP_NAMES)
def test_records_match_binary128_oracle(name):
    """Third build of the same oracle source in IEEE binary128 (libquadmath): no rounding noise left in the
    reference's formulation.  Every decision of every ray is identical and positions agree to 1e-10 mm — what differs
    between the HIP path and the f64 reference is the reference's own rounding."""
    from oracle.oracle import Oracle
    full = make_setup(name)
    n = 40_000
    with sa.RayTracer(full) as rt:
        rec = rt.traceAxionWrapper(n, seed=44)
    ref = Oracle(full, "q").trace_records(n, seed=44)
    for f in ("passed", "passedTillWindow", "hitNickel", "shellNumber", "kinds", "kindsWindow"):
        np.testing.assert_array_equal(rec[f], ref[f])
    m = rec["passedTillWindow"] == 1
    for f in ("pointdataX", "pointdataY", "pointdataR", "pointdataXBefore", "pointdataYBefore", "deviationDet", "yawAngles"):
        assert np.abs(rec[f] - ref[f]).max(initial=0.0) < 1e-10, f
    for f in ("weights", "reflect", "transmissionMagnet", "transProbWindow", "energiesAx"):
        np.testing.assert_allclose(rec[f][m], ref[f][m], rtol=2e-8, atol=0)


@pytest.mark.parametrize("name", ["babyiaxo_xmm", "cast_llnl", "cast_abrixas", "babyiaxo_xmm_gas"])
def test_histogram_matches_oracle(name):
    from oracle.oracle import Oracle
    full = make_setup(name)
    n = 400_000
    with sa.RayTracer(full) as rt:
        img, summ = rt.trace_histogram(n, seed=9)
    oimg, osumm, _ = Oracle(full).trace_histogram(n, seed=9)
    assert summ["N_RAYS"] == n == osumm["N_RAYS"]
    assert summ["N_REACHED_TELESCOPE"] == osumm["N_REACHED_TELESCOPE"] or abs(summ["N_REACHED_TELESCOPE"] - osumm["N_REACHED_TELESCOPE"]) <= 3
    for k in ("N_SHELL_SELECTED", "N_PASSED", "N_PASSED_TILL_WINDOW", "N_HIT_NICKEL"):
        assert abs(summ[k] - osumm[k]) <= max(3.0, FLAG_MISMATCH_MAX * n), k
    for k in ("SUM_WEIGHTS", "SUM_X", "SUM_Y", "SUM_R", "SUM_WEIGHTS_SQ"):
        assert summ[k] == pytest.approx(osumm[k], rel=1e-3), k
    assert summ["N_OUTSIDE_IMAGE"] == 0
    # image: identical up to rays within the oracle's noise of a pixel edge (pixel = 0.055 mm, noise ~1e-3 mm)
    assert np.abs(img - oimg).sum() <= 2e-2 * oimg.sum()
    c, oc = img.reshape(32, 8, 32, 8).sum(axis=(1, 3)), oimg.reshape(32, 8, 32, 8).sum(axis=(1, 3))
    assert np.abs(c - oc).sum() <= 3e-3 * oc.sum()
    # the image holds exactly the passed flux
    assert img.sum() == pytest.approx(summ["SUM_WEIGHTS"], rel=1e-12)


@pytest.mark.parametrize("name", ["babyiaxo_xmm", "cast_llnl", "babyiaxo_xmm_gas"])
def test_image_and_flux_agree_with_the_reference_rng_stream_within_monte_carlo_error(name):
    """The acceptance sentence of the north star, literally: "results match the Nim CPU reference's focal-plane image ... within
    sqrt(N) Monte-Carlo error".  Two INDEPENDENT sets of rays: the HIP path on its Philox stream, and the oracle driven by the
    reference's own generator (std/random's xoroshiro128+ behind randomize(299792458), raytracer.nim:276, six draws per ray in the
    reference's order; oracle.trace_records_nim_stream).  Per pixel of a 16 x 16 image the difference of the two fluxes over
    its Monte-Carlo error (sum of squared weights of both) is a unit normal: chi^2 ~ n_pixels +- sqrt(2 n_pixels); the total flux
    and the three counters agree within five standard deviations.  (Every other parity test traces the SAME rays on both sides -
    stronger ray by ray, but blind to anything the two sides would share through the common stream.)"""
    from oracle.oracle import Oracle
    full = make_setup(name)
    n = 400_000
    with sa.RayTracer(full) as rt:
        g = rt.traceAxionWrapper(n, seed=20261005)
    o = Oracle(full).trace_records_nim_stream(n, init_variant=1)
    chip = full.setup.chip_x_max

    def binned(rec):
        p = rec[rec["passed"] != 0]
        ix = np.clip((p["pointdataX"] / chip * 16).astype(int), 0, 15)
        iy = np.clip((p["pointdataY"] / chip * 16).astype(int), 0, 15)
        w, w2, cnt = np.zeros((16, 16)), np.zeros((16, 16)), np.zeros((16, 16))
        np.add.at(w, (iy, ix), p["weights"]); np.add.at(w2, (iy, ix), p["weights"] ** 2); np.add.at(cnt, (iy, ix), 1.0)
        return w, w2, cnt, p

    gw, gw2, gc, gp = binned(g)
    ow, ow2, oc, op = binned(o)
    lit = (gc >= 30) & (oc >= 30)
    assert lit.sum() >= 16, "too few populated pixels to test anything"
    chi2 = ((gw - ow)[lit] ** 2 / (gw2 + ow2)[lit]).sum()
    k = int(lit.sum())
    assert chi2 < k + 5.0 * np.sqrt(2.0 * k), (chi2, k)
    assert chi2 > k - 5.0 * np.sqrt(2.0 * k), (chi2, k)     # (not suspiciously equal either: the rays ARE independent)
    z = (gp["weights"].sum() - op["weights"].sum()) / np.sqrt((gp["weights"] ** 2).sum() + (op["weights"] ** 2).sum())
    assert abs(z) < 5.0, z
    for f in ("passed", "passedTillWindow", "hitNickel"):
        pg, po = (g[f] != 0).mean(), (o[f] != 0).mean()
        assert abs(pg - po) < 5.0 * np.sqrt((pg * (1 - pg) + po * (1 - po)) / n) + 1e-12, (f, pg, po)


def test_effective_area_curve_agrees_with_the_reference_rng_stream_within_monte_carlo_error():
    """... and its effective-area curve: the fused angular scan on the Philox stream against one oracle run per angle on the
    reference's generator (performAngularScan's shape, raytracer.nim:2791-2800: fresh draws for every angle), chip 100 mm and the
    effective-area flags.  Per angle the two fluxes differ by less than five Monte-Carlo errors, and so do
    the max-normalised curves."""
    from oracle.oracle import Oracle
    full = make_setup("babyiaxo_xmm")
    full.setup.chip_x_max = full.setup.chip_y_max = 100.0
    flags = L.CF_IGNORE_DET_WINDOW | L.CF_IGNORE_GAS_ABS | L.CF_IGNORE_CONV_PROB
    angles = np.array([0.0, 0.1, 0.2, 0.3])
    n = 300_000
    with sa.RayTracer(full) as rt:
        per, shared = rt.trace_angular_scan(angles, n, seed=77, flags=flags)
    o = Oracle(full)
    of, of2 = np.zeros(len(angles)), np.zeros(len(angles))
    for i, a in enumerate(angles):
        s = full.setup.copy()
        s.telescope_turned_y_deg = float(a)
        rec = o.trace_records_nim_stream(n, ray_id_offset=i * n, flags=flags, init_variant=1, setup=s)   # the stream goes on from angle to angle
        w = rec["weights"][rec["passed"] != 0]
        of[i], of2[i] = w.sum(), (w ** 2).sum()
    gf, gf2 = per["SUM_WEIGHTS"], per["SUM_WEIGHTS_SQ"]
    z = (gf - of) / np.sqrt(gf2 + of2)
    assert np.all(np.abs(z) < 5.0), z
    assert gf[0] > gf[1] > gf[2] > gf[3] > 0.3 * gf[0]
    rel_g, rel_o = gf / gf.max(), of / of.max()
    err = np.sqrt(gf2 + of2) / gf.max() * 2.0       # (both normalisations carry an error of their own)
    assert np.all(np.abs(rel_g - rel_o) < 5.0 * err + 1e-12), (rel_g, rel_o)


def test_flags_change_weights_like_the_reference():
    from oracle.oracle import Oracle
    full = make_setup("babyiaxo_xmm")
    o = Oracle(full, "ld")
    with sa.RayTracer(full) as rt:
        for flags in (L.CF_IGNORE_DET_WINDOW, L.CF_IGNORE_GAS_ABS, L.CF_IGNORE_REFLECTION, L.CF_IGNORE_CONV_PROB, 0b1111):
            rec = rt.traceAxionWrapper(20_000, seed=3, flags=flags)
            ref = o.trace_records(20_000, seed=3, flags=flags)
            compare_records(rec, _as_gold(ref), 5e-6, 1e-6, 1e-4)
            if flags & L.CF_IGNORE_REFLECTION:
                assert np.all(rec["reflect"][rec["passedTillWindow"] == 1] == 1.0)


def test_ray_ids_make_results_independent_of_batching():
    """Philox counter = global ray id: one call over [0, N) == two calls over [0, N/2) + [N/2, N), ray for ray;
    accumulate mode adds; a second identical call is deterministic in every count."""
    full = make_setup("babyiaxo_xmm")
    n = 100_000
    with sa.RayTracer(full) as rt:
        a = rt.traceAxionWrapper(n, seed=8)
        b = np.zeros(n, dtype=a.dtype)   # (np.concatenate would repack the padded record dtype)
        b[:n // 2] = rt.traceAxionWrapper(n // 2, seed=8)
        b[n // 2:] = rt.traceAxionWrapper(n - n // 2, seed=8, ray_id_offset=n // 2)
        assert a.tobytes() == b.tobytes()
        img, s = rt.trace_histogram(n, seed=8)
        img1, s1 = rt.trace_histogram(n // 2, seed=8)
        img2, s2 = rt.trace_histogram(n - n // 2, seed=8, ray_id_offset=n // 2, accumulate=True)
        for k in ("N_RAYS", "N_PASSED", "N_HIT_NICKEL", "N_REACHED_TELESCOPE", "N_SHELL_SELECTED", "N_PASSED_TILL_WINDOW"):
            assert s2[k] == s[k], k
        np.testing.assert_allclose(img2, img, rtol=1e-11, atol=1e-30)
        # histogram == records binned on the host (prepareHeatmap, raytracer.nim:838-842)
        p = a[a["passed"] == 1]
        h = np.zeros((256, 256))
        np.add.at(h, (np.floor(p["pointdataY"] / (14.0 / 256)).astype(int), np.floor(p["pointdataX"] / (14.0 / 256)).astype(int)), p["weights"])
        np.testing.assert_allclose(img, h, rtol=1e-11, atol=1e-30)
        assert s["N_PASSED"] == len(p) and s["SUM_WEIGHTS"] == pytest.approx(p["weights"].sum(), rel=1e-12)
        # different seeds give different rays
        assert rt.traceAxionWrapper(1000, seed=9).tobytes() != a[:1000].tobytes()


def test_edge_cases():
    full = make_setup("babyiaxo_xmm")
    with sa.RayTracer(full) as rt:
        assert len(rt.traceAxionWrapper(0)) == 0                       # empty buffer
        img, s = rt.trace_histogram(0)
        assert s["N_RAYS"] == 0 and not img.any()
        r1 = rt.traceAxionWrapper(1, seed=1, ray_id_offset=2 ** 40 + 5)  # ragged size, ids beyond 32 bit
        r2 = rt.traceAxionWrapper(3, seed=1, ray_id_offset=2 ** 40 + 3)
        assert r1.tobytes() == r2[2:].tobytes()
        r3 = rt.traceAxionWrapper(257, seed=1)                           # not a multiple of the block size
        assert r3.tobytes() == rt.traceAxionWrapper(1000, seed=1)[:257].tobytes()
        # non-square / other image sizes
        p = rt.trace_params(50_000, seed=2, image_n=64)
        p.image_ny = 32
        acc = np.zeros(64 * 32 + L.SART_ACC_COUNT)
        summ = L.Summary()
        img = np.zeros((32, 64))
        L.check(rt.lib.sart_trace_histogram(rt.handle, C.byref(p), L.as_dp(img), C.byref(summ)))
        img256, s256 = rt.trace_histogram(50_000, seed=2)
        np.testing.assert_allclose(img, img256.reshape(32, 8, 64, 4).sum(axis=(1, 3)), rtol=1e-11, atol=1e-30)


def test_errors_are_reported_not_crashed():
    lib = L.load_sart()
    h = C.c_void_p()
    L.check(lib.sart_create(0, C.byref(h)))
    try:
        p = L.TraceParams(n_rays=10, image_nx=256, image_ny=256, image_x_max=14.0, image_y_max=14.0)
        buf = np.zeros(10, dtype=L.AXION_DTYPE)
        assert lib.sart_trace_records(h, C.byref(p), buf.ctypes.data_as(C.c_void_p)) == -3   # NOT_READY
        s = sa.newFullSetup(L.ES_BABYIAXO, L.DK_INGRIDIAXO, L.SK_VACUUM, L.TK_XMM)
        s.telescope_kind = L.TK_CUSTOM_BABYIAXO
        assert lib.sart_set_setup(h, C.byref(s)) == -4                                         # UNSUPPORTED (doAssert :1233)
        s = sa.newFullSetup(L.ES_BABYIAXO, L.DK_INGRIDIAXO, L.SK_VACUUM, L.TK_XMM)
        s.number_of_holes = 1 << 30                                                            # would be a 2^30-pass loop per ray
        assert lib.sart_set_setup(h, C.byref(s)) == -1 and b"number_of_holes" in lib.sart_last_error()
        s.number_of_holes, s.hole_type = 1, 17
        assert lib.sart_set_setup(h, C.byref(s)) == -1 and b"hole_type" in lib.sart_last_error()
        for field, j in (("pipe_vt3_xrt_radius", None), ("all_angles_deg", 40), ("optics_entrance", 2), ("chip_y_max", None)):
            s = sa.newFullSetup(L.ES_BABYIAXO, L.DK_INGRIDIAXO, L.SK_VACUUM, L.TK_XMM)
            if j is None:
                setattr(s, field, float("nan"))
            else:
                getattr(s, field)[j] = float("inf")
            assert lib.sart_set_setup(h, C.byref(s)) == -1 and field.encode() in lib.sart_last_error(), field
        s = sa.newFullSetup(L.ES_BABYIAXO, L.DK_INGRIDIAXO, L.SK_VACUUM, L.TK_XMM)
        s.all_r1[60] = float("nan")                                                            # behind n_shells = 58: not looked at
        assert lib.sart_set_setup(h, C.byref(s)) == 0
        assert lib.sart_set_axion_mass(h, float("nan")) == -1 and lib.sart_set_telescope_angles(h, float("inf"), 0.0) == -1
        assert lib.sart_set_telescope_angles(h, float("nan"), 0.01) == 0
        bad = np.array([0.1, 0.5, 0.9])
        assert lib.sart_set_solar_tables(h, L.as_dp(bad), L.as_dp(np.ones((3, 4))), L.as_dp(np.ones(4)), 3, 4) == -1
        assert b"1.0" in lib.sart_last_error()
    finally:
        lib.sart_destroy(h)


def test_angular_scan_curve_matches_oracle_rms():
    """BASELINE metric 2: effective-area curve RMS vs the CPU reference path, same seed family (config 4 style:
    XMM shells, chip enlarged to 100 mm, effective-area flags)."""
    from oracle.oracle import Oracle
    full = make_setup("babyiaxo_xmm")
    full.setup.chip_x_max = full.setup.chip_y_max = 100.0
    flags = L.CF_IGNORE_DET_WINDOW | L.CF_IGNORE_GAS_ABS | L.CF_IGNORE_CONV_PROB
    full.flags = flags
    n_per = 100_000
    with sa.RayTracer(full) as rt:
        angles, flux, rel = sa.performAngularScan(rt, 0.0, 0.3, 7, n_rays_per_angle=n_per, seed=21)
        rt.set_telescope_angles(turned_y_deg=0.0)
    o = Oracle(full)
    oflux = []
    for i, a in enumerate(angles):
        s = full.setup.copy()
        s.telescope_turned_y_deg = a
        _, summ, _ = o.trace_histogram(n_per, seed=21, ray_id_offset=i * n_per, setup=s)
        oflux.append(summ["SUM_WEIGHTS"])
    orel = np.array(oflux) / max(oflux)
    rms = np.sqrt(np.mean((rel - orel) ** 2))
    assert rms < 1e-3, (rel, orel)           # sqrt(N) Monte-Carlo error per point is ~3e-3; same seeds do far better
    assert rel[0] == 1.0 and np.all(np.diff(rel) < 0) and 0.2 < rel[-1] < 0.8   # shape of the XMM vignetting curve


def test_axion_mass_scan_in_gas_stage():
    from oracle.oracle import Oracle
    full = make_setup("babyiaxo_xmm_gas")
    o = Oracle(full, "ld")
    with sa.RayTracer(full) as rt:
        base = rt.trace_histogram(100_000, seed=4)[1]["SUM_WEIGHTS"]
        for m in (0.0, 0.008235, 0.05):
            rt.set_axion_mass(m)
            got = rt.trace_histogram(100_000, seed=4)[1]["SUM_WEIGHTS"]
            s = full.setup.copy(); s.m_axion = m
            want = o.trace_histogram(100_000, seed=4, setup=s)[1]["SUM_WEIGHTS"]
            assert got == pytest.approx(want, rel=1e-6), m
        assert got != base


@pytest.mark.parametrize("config", ["cast_llnl_gold_1e8", "babyiaxo_xmm_1e9"])
def test_full_size_properties(config):
    """BASELINE configs 2 and 3 at full size (1e8 / 1e9 rays, full-size tables): properties that do not need the
    oracle — conservation, additivity over ray-id shards, agreement of rates with a 1e6-ray oracle run."""
    from oracle.oracle import Oracle
    if config == "cast_llnl_gold_1e8":
        full = sa.initFullSetup(L.ES_CAST, L.DK_INGRID2018, L.SK_VACUUM, L.TK_LLNL, reflectivity="gold")
        n = 100_000_000
    else:
        full = sa.initFullSetup()
        n = 1_000_000_000
    with sa.RayTracer(full) as rt:
        img, s = rt.trace_histogram(n, seed=1)
        # same rays in 4 shards, accumulated
        q = n // 4
        for k in range(4):
            img4, s4 = rt.trace_histogram(q, seed=1, ray_id_offset=k * q, accumulate=(k > 0))
    assert s["N_RAYS"] == n
    for k in ("N_PASSED", "N_HIT_NICKEL", "N_REACHED_TELESCOPE", "N_SHELL_SELECTED", "N_PASSED_TILL_WINDOW"):
        assert s4[k] == s[k], k                                   # counts are exact and order-independent
    assert s4["SUM_WEIGHTS"] == pytest.approx(s["SUM_WEIGHTS"], rel=1e-10)
    np.testing.assert_allclose(img4, img, rtol=1e-9, atol=img.max() * 1e-13)
    assert img.sum() == pytest.approx(s["SUM_WEIGHTS"], rel=1e-10) and s["N_OUTSIDE_IMAGE"] == 0
    assert s["N_PASSED"] <= s["N_PASSED_TILL_WINDOW"] <= s["N_SHELL_SELECTED"] <= s["N_REACHED_TELESCOPE"] <= n
    # rates against the oracle on its first 1e6 rays (binomial error ~1e-3)
    m = 1_000_000
    _, os_, _ = Oracle(full).trace_histogram(m, seed=1)
    for k in ("N_PASSED", "N_REACHED_TELESCOPE", "N_SHELL_SELECTED"):
        assert s[k] / n == pytest.approx(os_[k] / m, abs=4e-3), k
    assert s["SUM_WEIGHTS"] / n == pytest.approx(os_["SUM_WEIGHTS"] / m, rel=2e-2)
    # image symmetry of an on-axis telescope: centroid at the chip centre (the LLNL image sits off-centre in x:
    # 2.75 deg pipe rotation + 83 mm shift, raytracer.nim:797-814)
    if config == "babyiaxo_xmm_1e9":
        assert s["SUM_X"] / s["N_PASSED"] == pytest.approx(7.0, abs=0.05)
    assert s["SUM_Y"] / s["N_PASSED"] == pytest.approx(7.0, abs=0.2)


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_two_rank_rehearsal(scaling):
    """bench.py's multi-rank path (ray-id sharding, shared stream, single reduce) with 2 ranks sharing this GPU over
    gloo — the RCCL run on 2/4/8 GPUs is the driver's; this catches ordering bugs between the launches and the reduce.
    weak: every rank traces --rays-per-step; strong: the step's total is split over the ranks (BASELINE configs[4])."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SART_BENCH_BACKEND="gloo", SART_BENCH_DEVICE="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29547" if scaling == "weak" else "29549", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3",
           "--warmup", "1", "--rays-per-step", "2e7", "--profile-run", "--scaling", scaling]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    total = (2 if scaling == "weak" else 1) * 3 * 2e7
    assert d["n_gpus"] == 2 and d["config"]["total_rays"] == total and d["scaling"] == scaling
    assert d["results"]["passed_fraction"] == pytest.approx(0.2144, abs=2e-3)
    assert d["results"]["image_sum"] == pytest.approx(d["results"]["flux"], rel=1e-9)
    assert d["value"] > 1e9
    # the result, not only the rate: the FIXED64 image of the ids [0, 2e8) reduced from the two ranks is rank 0's own image of
    # those ids slot for slot, and (same input tables as the host of the constants) the committed one-GPU image
    proof = d["bitwise_proof"]
    assert proof["ranks"] == 2 and proof["equal"] is True and proof["slots_that_differ"] == 0 and proof["n_rays"] == 200_000_000
    assert d["multi_rank_bitwise_equal_to_single_gpu"] is True
    assert proof["matches_committed_constants"] in (True, None), proof     # None: this host built other tables; False fails
    if proof["matches_committed_constants"] is None:
        assert "tables_sha256" in proof["committed"] or proof["committed"] is None


def test_bench_bitwise_constants_are_current():
    """tests/golden/bench_bitwise_fixed64.json (made by `bench.py --emit-bitwise-constants` on one MI355X) is what this build
    computes on one GPU: a --gpus N run of the driver is then held to the same bytes.  Skipped - loudly - when this host's
    libm built other input tables than the host of the constants (tables_sha256)."""
    import json
    import os
    import torch
    import bench
    from solaraxionraytracing_amd import distributed as D
    with open(bench.BITWISE_CONSTANTS) as f:
        want = json.load(f)
    full, flags = bench.make_setup("babyiaxo_xmm")
    if bench.tables_sha256(full) != want["tables_sha256"]:
        pytest.skip("this host built other input tables than the host of the committed constants")
    with sa.RayTracer(full) as rt:
        blk = bench.bitwise_proof(rt, full, flags, 0, 1, torch.device("cuda", 0), want["seed"], D)
    assert blk["equal"] and blk["matches_committed_constants"] is True, (blk, want)
    assert blk["image_sha256"] == want["image_sha256"] and blk["sum_weights_hex"] == want["sum_weights_hex"]


def test_trace_records_chunked_path_is_byte_identical():
    """sart_trace_records above one chunk: rays traced chunk by chunk into two device buffers, copied out on a second stream
    into a pre-faulted caller buffer.  Same bytes as one launch + one copy, for ragged sizes and unaligned destinations."""
    import os
    full = make_setup("cast_llnl")
    n = 150_001
    with sa.RayTracer(full) as rt:
        one = rt.traceAxionWrapper(n, seed=31, ray_id_offset=7)          # n < 1 Mi records: one launch, one copy
    for chunk, prefault in ((65_536, True), (40_000, False), (149_999, True)):
        os.environ["SART_RECORDS_CHUNK"] = str(chunk)
        if not prefault:
            os.environ["SART_NO_HOST_PREFAULT"] = "1"
        try:
            with sa.RayTracer(full) as rt:
                raw = np.empty(n * 208 + 24, dtype=np.uint8)              # destination 24 bytes off any page boundary
                view = raw[24:].view(L.AXION_DTYPE)
                p = rt.trace_params(n, seed=31, ray_id_offset=7)
                L.check(rt.lib.sart_trace_records(rt.handle, C.byref(p), view.ctypes.data_as(C.c_void_p)))
                again = rt.traceAxionWrapper(n, seed=31, ray_id_offset=7)
        finally:
            os.environ.pop("SART_RECORDS_CHUNK", None)
            os.environ.pop("SART_NO_HOST_PREFAULT", None)
        assert view.tobytes() == one.tobytes(), (chunk, prefault)
        assert again.tobytes() == one.tobytes()
        assert raw[:24].tobytes() != b"" and one["passed"].sum() > 0.5 * n


RCCL_ALONE = r'''
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, %r)
import solaraxionraytracing_amd as sa
from solaraxionraytracing_amd import _lib as L
from solaraxionraytracing_amd import distributed as D
rank, world, local = D.init_process_group_from_env("nccl")
assert (rank, world) == (0, 1)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
assert dist.get_backend() == "nccl"
n = sa.accumulator_len(256)
with sa.RayTracer(sa.initFullSetup()) as rt:
    p = rt.trace_params(3_000_000, seed=5)
    for mode, dtype in (("f64", torch.float64), ("fixed64", torch.int64)):
        rt.set_accumulation_mode(mode)
        acc = torch.zeros(n, dtype=torch.float64, device="cuda:0")
        rt.trace_histogram_device(p, acc.data_ptr())
        torch.cuda.synchronize()
        before = acc.clone()
        for dst in (0, None):
            out = D.reduce_accumulator(acc, dst=dst, fixed64=(mode == "fixed64"), even_alone=True)   # the RCCL call of an N-rank run
            torch.cuda.synchronize()
            assert out is acc and torch.equal(acc.view(torch.int64), before.view(torch.int64)), (mode, dst)
        if mode == "fixed64":
            rt.finalize_accumulator_device(p, acc.data_ptr())
        s = acc[256 * 256:].cpu().numpy()
        assert s[L.ACC["N_RAYS"]] == 3_000_000 and s[L.ACC["N_PASSED"]] > 5e5 and abs(acc[:256 * 256].sum().item() / s[L.ACC["SUM_WEIGHTS"]] - 1) < 1e-9
dist.barrier()
dist.destroy_process_group()
print("RCCL_ALONE_OK", flush=True)
'''


def test_rccl_carries_the_accumulator_in_a_one_rank_group():
    """What a one-GPU box can show of the RCCL leg (the N > 1 runs are the driver's): torch.distributed backend "nccl" comes up
    on this card, and the path's one collective - reduce / all-reduce of the fused accumulator, as f64 and as the int64 of a raw
    SART_ACCUM_FIXED64 accumulator - runs through the RCCL communicator on the accumulator the kernel has just filled and leaves
    it bit for bit as it was."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29561", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-c", RCCL_ALONE % root], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "RCCL_ALONE_OK" in out.stdout, (out.stdout[-500:], out.stderr[-3000:])


def test_bench_gpus_n_starts_its_own_ranks_and_refuses_to_lie():
    """`python bench.py --gpus 2` WITHOUT torchrun: the script starts two ranks itself (gloo rehearsal on this one card) and
    the line says n_gpus 2; `--gpus 8` on a one-GPU box exits non-zero without a line."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env = dict(base, SART_BENCH_BACKEND="gloo", SART_BENCH_DEVICE="0")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                          "--rays-per-step", "2e7", "--profile-run", "--scaling", "strong"], env=env, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["world_size"] == 2 and d["backend"] == "gloo" and d["scaling"] == "strong"
    assert d["config"]["rays_per_step_per_rank"] == [10_000_000, 10_000_000] and d["config"]["total_rays"] == 6e7
    assert d["reduce_ms"] > 0 and d["results"]["passed_fraction"] == pytest.approx(0.2144, abs=2e-3)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "1"], env=base,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 2 and "{" not in out.stdout and "refused" in out.stderr
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "1"], env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 2 and "{" not in out.stdout


@pytest.mark.parametrize("variant", ["vacuum", "gas", "rotated", "rotated_0.5", "rotated_3", "abrixas_rotated"])
def test_early_rejection_stage_is_exact(variant):
    """Stage A0 (rays classified from the bore-exit radius alone) must not change any counter or the image: same
    launch with the stage switched off (SART_NO_EARLY_REJECT, read when a context uploads its parameter block).
    vacuum / gas / rotated are the instantiations of the histogram kernel.  For a turned telescope the zones in terms of the
    radial distance at the entrance (inner disc, ring, beyond the last shell) carry the margin of the tilt (round 6,
    sart_api.hip: build_zones): 0.11, 0.5 and 3 degrees - a margin that were too small by any factor would retire live rays at
    one of them -, and Abrixas (the other telescope with an inner disc) behind the CAST magnet."""
    import os
    from solaraxionraytracing_amd import _lib as L
    # full-size BabyIAXO / XMM tables: the configuration the stage exists for
    if variant == "abrixas_rotated":
        full = sa.initFullSetup(L.ES_CAST, L.DK_INGRID2017, L.SK_VACUUM, L.TK_ABRIXAS)
        full.setup.telescope_turned_y_deg = 0.2
    else:
        full = sa.initFullSetup(stage=L.SK_GAS) if variant == "gas" else sa.initFullSetup()
    if variant == "rotated":
        full.setup.telescope_turned_y_deg = 0.1
        full.setup.telescope_turned_x_deg = -0.05
    elif variant.startswith("rotated_"):
        full.setup.telescope_turned_y_deg = float(variant.split("_")[1])
        full.setup.chip_x_max = full.setup.chip_y_max = 100.0
    n = 20_000_000
    with sa.RayTracer(full) as rt:
        img_a, s_a = rt.trace_histogram(n, seed=17)
    os.environ["SART_NO_EARLY_REJECT"] = "1"
    try:
        with sa.RayTracer(full) as rt:
            img_b, s_b = rt.trace_histogram(n, seed=17)
    finally:
        del os.environ["SART_NO_EARLY_REJECT"]
    for k in ("N_RAYS", "N_REACHED_TELESCOPE", "N_SHELL_SELECTED", "N_HIT_NICKEL", "N_PASSED_TILL_WINDOW", "N_PASSED", "N_OUTSIDE_IMAGE"):
        assert s_a[k] == s_b[k], k
    if variant != "abrixas_rotated":
        assert s_a["N_REACHED_TELESCOPE"] / n == pytest.approx(0.5475, abs=1e-3)
    np.testing.assert_allclose(img_a, img_b, rtol=1e-10, atol=img_b.max() * 1e-14)
    assert s_a["SUM_WEIGHTS"] == pytest.approx(s_b["SUM_WEIGHTS"], rel=1e-12)


def test_turned_telescope_has_the_zones_of_the_entrance_plane():
    """The zones of stage A0 for a turned telescope (round 6): beside the pipes' zone the inner disc and the annulus beyond the last
    shell, narrower by the tilt's margin; none of them with SART_NO_TILT_ZONES - checked on the zone table the launch would use."""
    import os
    import subprocess
    import sys
    code = ("import ctypes as C, json, solaraxionraytracing_amd as sa\n"
            "full = sa.initFullSetup(n_radii=400, n_energies=300, refl_n_angles=50, refl_n_energies=50)\n"
            "out = {}\n"
            "for y in (0.0, 0.1, 0.3, 40.0):\n"
            "    full.setup.telescope_turned_y_deg = y\n"
            "    with sa.RayTracer(full) as rt:\n"
            "        lo, hi, reached = (C.c_uint32 * 8)(), (C.c_uint32 * 8)(), C.c_uint32()\n"
            "        n = rt.lib.sart_internal_zones(rt.handle, lo, hi, C.byref(reached))\n"
            "        out[str(y)] = [n, reached.value, [(hi[i] - lo[i]) / 2.0 ** 32 for i in range(n)]]\n"
            "print(json.dumps(out))\n")
    import json
    res = {}
    for knob in (False, True):
        env = dict(os.environ)
        if knob:
            env["SART_NO_TILT_ZONES"] = "1"
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300,
                             cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        assert out.returncode == 0, out.stderr[-2000:]
        res[knob] = json.loads(out.stdout.strip().splitlines()[-1])
    on, off = res[False], res[True]
    assert on["0.0"] == off["0.0"] and on["0.0"][0] >= 3               # not turned: the zones of rounds 1-5, with or without the knob
    assert off["0.1"][0] == off["0.3"][0] == 1 and off["0.1"][1] == 0   # knob: only the pipes' zone for a turned telescope
    assert on["0.1"][0] == on["0.0"][0] and on["0.3"][0] == on["0.0"][0]
    area = lambda z: sum(z[2])                                         # fraction of the rays stage A0 retires (u3 is uniform)
    assert area(on["0.0"]) > area(on["0.1"]) > area(on["0.3"]) > area(off["0.3"]) + 0.05
    assert on["40.0"][0] == 1                                          # no radial zone survives a tilt of 40 degrees


def test_launch_splitting_beyond_32bit_ray_indices():
    """One call with more than 2^31 rays is split into launches internally; ray ids continue across the split."""
    full = make_setup("babyiaxo_xmm")
    n = (1 << 31) + 12_345
    with sa.RayTracer(full) as rt:
        _, s = rt.trace_histogram(n, seed=2)
        _, s1 = rt.trace_histogram(1 << 31, seed=2)
        _, s2 = rt.trace_histogram(12_345, seed=2, ray_id_offset=1 << 31, accumulate=True)
    assert s["N_RAYS"] == n == s2["N_RAYS"]
    for k in ("N_PASSED", "N_HIT_NICKEL", "N_REACHED_TELESCOPE", "N_SHELL_SELECTED"):
        assert s[k] == s2[k], k


@pytest.mark.parametrize("kw", [dict(emission="flat", reflectivity="analytic"),
                                dict(emission="flat", reflectivity="henke", n_radii=2048, n_energies=700),
                                dict(emission="primakoff", reflectivity="analytic", n_radii=9, n_energies=2, refl_n_angles=2, refl_n_energies=2)])
def test_other_table_shapes_against_binary128_oracle(kw):
    """Stress inputs of SURVEY 8(d): flat emission (E3: wide guide brackets, every energy equally likely), analytic
    reflectivity (G2), the largest radius table the LDS stage takes (2048) and degenerate tiny tables."""
    from oracle.oracle import Oracle
    args = dict(n_radii=300, n_energies=200, refl_n_angles=64, refl_n_energies=48)
    args.update(kw)
    for full in (sa.initFullSetup(**args), sa.initFullSetup(L.ES_CAST, L.DK_INGRID2018, L.SK_VACUUM, L.TK_LLNL, **args)):
        n = 30_000
        with sa.RayTracer(full) as rt:
            rec = rt.traceAxionWrapper(n, seed=3)
            img, s = rt.trace_histogram(n, seed=3)
        ref = Oracle(full, "q").trace_records(n, seed=3)
        for f in ("passed", "passedTillWindow", "hitNickel", "shellNumber", "kindsWindow"):
            np.testing.assert_array_equal(rec[f], ref[f])
        np.testing.assert_array_equal(rec["energiesPre"], ref["energiesPre"])
        m = rec["passed"] == 1
        assert np.abs(rec["pointdataX"] - ref["pointdataX"]).max(initial=0.0) < 1e-10
        np.testing.assert_allclose(rec["weights"][m], ref["weights"][m], rtol=2e-8)
        assert s["N_PASSED"] == m.sum() and s["SUM_WEIGHTS"] == pytest.approx(rec["weights"][m].sum(), rel=1e-12)


def test_reduce_across_devices_single_process_entry_point():
    """sart_reduce_across_devices: argument checking and the n == 1 path run here; the n > 1 RCCL
    path needs several GPUs in one process (one-GPU box: not exercised) — bench.py's per-process reduce is."""
    full = make_setup("babyiaxo_xmm")
    lib = L.load_sart()
    import torch
    with sa.RayTracer(full) as rt:
        acc = torch.zeros(sa.accumulator_len(256), dtype=torch.float64, device="cuda")
        p = rt.trace_params(50_000, seed=1)
        rt.trace_histogram_device(p, acc.data_ptr())
        ctxs = (C.c_void_p * 1)(rt.handle)
        accs = (C.c_void_p * 1)(acc.data_ptr())
        assert lib.sart_reduce_across_devices(ctxs, accs, 1, acc.numel(), 0) == 0     # also synchronises the stream
        assert acc[256 * 256 + L.ACC["N_RAYS"]].item() == 50_000
        assert lib.sart_reduce_across_devices(ctxs, accs, 1, acc.numel(), 3) == -1    # root out of range
        two = (C.c_void_p * 2)(rt.handle, rt.handle)
        accs2 = (C.c_void_p * 2)(acc.data_ptr(), acc.data_ptr())
        assert lib.sart_reduce_across_devices(two, accs2, 2, acc.numel(), 0) == -1    # same device twice
        assert b"distinct devices" in lib.sart_last_error()


def test_high_resolution_heatmap_and_y_slice():
    """The other binnings of generateResultPlots through the same entry point: the 3000 x 3000 heat map (:2626) and the
    y-slice histogram of the rays within 0.05 mm of the chip centre in x (:2551-2559), against oracle records."""
    from oracle.oracle import Oracle
    full = make_setup("babyiaxo_xmm")
    n = 400_000
    o = Oracle(full)
    rec = o.trace_records(n, seed=21)
    ok = rec["passed"].astype(bool)
    with sa.RayTracer(full) as rt:
        img, s = rt.trace_image(n, 3000, 3000, seed=21)
        edges, flux = rt.y_slice_histogram(n, seed=21)
        coarse, _ = rt.trace_histogram(n, seed=21)
    assert img.shape == (3000, 3000)
    assert img.sum() == pytest.approx(s["SUM_WEIGHTS"], rel=1e-11) and s["N_OUTSIDE_IMAGE"] == 0
    assert s["SUM_WEIGHTS"] == pytest.approx(rec["weights"][ok].sum(), rel=2e-4)    # edge rays within the f64 oracle's noise
    # 3000 = 12 x 250, 256-pixel map over the same 14 mm: compare on a common 2 x 2 grid of quadrants
    q = lambda a: np.array([[a[:a.shape[0] // 2, :a.shape[1] // 2].sum(), a[:a.shape[0] // 2, a.shape[1] // 2:].sum()],
                            [a[a.shape[0] // 2:, :a.shape[1] // 2].sum(), a[a.shape[0] // 2:, a.shape[1] // 2:].sum()]])
    np.testing.assert_allclose(q(img), q(coarse), rtol=1e-9)
    # y slice: oracle rays with |x - 7| < 0.05 (positions are stored as -x + 7 / y + 7, :2203-2204)
    x, y, w = rec["pointdataX"][ok], rec["pointdataY"][ok], rec["weights"][ok]
    sel = np.abs(x - 7.0) < 0.05
    want, _ = np.histogram(y[sel], bins=edges, weights=w[sel])
    assert flux.size == 14000 and sel.sum() > 50
    assert flux.sum() == pytest.approx(want.sum(), rel=2e-2)          # rays within the oracle's f64 noise of the slice edge may flip
    # same rays land in the same 0.001 mm bins up to that noise: the two cumulative distributions along y never differ
    # by more than a few rays' weight
    assert np.abs(np.cumsum(flux) - np.cumsum(want)).max() <= 4.0 * w[sel].max()


def test_xmm_on_axis_effective_area_matches_published_values():
    """End-to-end physics pin that does not go through the oracle: a parallel X-ray beam (the reference's `--xrayTest`,
    :1765-1806) filling the 350 mm aperture, detector / gas / conversion factors switched off, so that
    pi R^2 * sum(weights) / N is the on-axis effective area of the 58-shell XMM-Newton optic with 0.25 um gold (Henke scans
    shipped by the reference).  Published for one XMM mirror module: ~1500 cm^2 at 1.5 keV, ~600 cm^2 at 8 keV (real mirrors
    have roughness and support-structure losses; ideal gold sits a little above at high energy)."""
    area_cm2 = np.pi * 35.0 ** 2
    flags = L.CF_XRAY_TEST | L.CF_IGNORE_DET_WINDOW | L.CF_IGNORE_GAS_ABS | L.CF_IGNORE_CONV_PROB
    got = {}
    for energy in (1.5, 8.0):
        src = L.TestSourceConfig()
        src.active, src.parallel = 1, 1
        src.energy, src.distance, src.radius, src.activity = energy, 2000.0, 350.0, 0.125
        full = sa.initFullSetup(flags=flags, source_cfg=src)
        full.setup.chip_x_max = full.setup.chip_y_max = 100.0
        with sa.RayTracer(full) as rt:
            _, s = rt.trace_histogram(5_000_000, seed=4, flags=flags)
        got[energy] = area_cm2 * s["SUM_WEIGHTS"] / s["N_RAYS"]
        assert s["N_SHELL_SELECTED"] / s["N_RAYS"] == pytest.approx(0.672, abs=5e-3)     # geometric open fraction of the aperture
    assert 1400.0 < got[1.5] < 1650.0, got
    assert 550.0 < got[8.0] < 750.0, got


def test_command_line_full_run_and_angular_scan(tmp_path):
    """`python -m solaraxionraytracing_amd` = the reference's `raytracer` binary: full run writes axion_image_IAXO.csv with
    the reference's columns; --angularScanMin/Max runs performAngularScan; --massScanMin/Max the fused mass scan."""
    import os
    import subprocess
    import sys
    out = tmp_path / "out"
    r = subprocess.run([sys.executable, "-m", "solaraxionraytracing_amd", "--rays", "300000", "--outpath", str(out), "--noPlots"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "Passed axions" in r.stdout and "The total flux" in r.stdout
    lines = open(out / "axion_image_IAXO.csv").read().splitlines()
    assert lines[0].startswith("x,y,photon flux,yr0,yr02") and len(lines) == 1 + 256 * 256
    r = subprocess.run([sys.executable, "-m", "solaraxionraytracing_amd", "--rays", "200000", "--outpath", str(out), "--ignoreDetWindow",
                        "--ignoreGasAbs", "--ignoreConvProb", "--angularScanMax", "0.05", "--numAngularScanPoints", "3"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    scan = np.loadtxt(out / "angular_scan_telescope_y.csv", delimiter=",", skiprows=1)
    assert scan.shape == (3, 4) and scan[0, 2] == 1.0 and scan[2, 2] < 1.0 and np.all(np.isnan(scan[:, 3]))
    # the same scan through the fused kernel (--fusedAngularScan: the same rays for every angle, errors in the fourth column)
    r = subprocess.run([sys.executable, "-m", "solaraxionraytracing_amd", "--rays", "200000", "--outpath", str(out), "--ignoreDetWindow",
                        "--ignoreGasAbs", "--ignoreConvProb", "--angularScanMax", "0.05", "--numAngularScanPoints", "3", "--fusedAngularScan"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    fused = np.loadtxt(out / "angular_scan_telescope_y.csv", delimiter=",", skiprows=1)
    assert fused.shape == (3, 4) and fused[0, 2] == 1.0 and fused[2, 2] < 1.0 and np.all(fused[:, 3] > 0) and np.all(fused[:, 3] < 0.05 * fused[:, 1])
    assert fused[0, 1] == pytest.approx(scan[0, 1], rel=1e-9)  # angle 0: the same ray ids in both shapes (rotation by 0 rounds, the unrotated kernel does not)
    assert abs(fused[2, 1] / scan[2, 1] - 1.0) < 0.05         # the other angles: other rays in the host loop
    # the third mode (not in the reference): the fused axion-mass scan on a gas-stage config.toml
    cfg = tmp_path / "config.toml"
    sample = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "make_nim_parity_kit.py")).read()
    template = sample.split('CONFIG_TEMPLATE = """')[1].split('"""')[0]
    cfg.write_text(template % ("BabyIAXO", "InGridIAXO", "gas", "XMM"))
    r = subprocess.run([sys.executable, "-m", "solaraxionraytracing_amd", "--rays", "300000", "--outpath", str(out), "--config", str(cfg),
                        "--massScanMin", "0.004", "--massScanMax", "0.012", "--numMassScanPoints", "9"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    ms = np.loadtxt(out / "axion_mass_scan.csv", delimiter=",", skiprows=1)
    assert ms.shape == (9, 5) and ms[:, 4].max() == 1.0 and int(np.argmax(ms[:, 1])) == 4 and np.all(ms[:, 2] < 0.1 * ms[:, 1])
    assert "maximum at m_a = 0.008 eV" in r.stdout


@pytest.mark.parametrize("name", ["babyiaxo_xmm", "cast_llnl", "babyiaxo_xmm_gas", "babyiaxo_xmm_rot"])
def test_specialised_and_generic_kernel_variants_agree(name):
    """The compile-time specialised instantiations (solar source, no hole loop: vacuum / gas stage / rotated telescope) and the
    generic ones are the same source: same rays, same counters, same image (SART_FORCE_GENERIC is read when a context is
    created).  The vacuum specialisation carries the z extent of the path and multiplies pathCB^2 by 1 + slope^2 in phase B;
    the others carry the path length itself."""
    import os
    full = make_setup(name)
    n = 5_000_000
    with sa.RayTracer(full) as rt:
        img_a, s_a = rt.trace_histogram(n, seed=23, ray_id_offset=777)
        assert s_a["N_PASSED"] > 0.1 * n
    os.environ["SART_FORCE_GENERIC"] = "1"
    try:
        with sa.RayTracer(full) as rt:
            img_b, s_b = rt.trace_histogram(n, seed=23, ray_id_offset=777)
    finally:
        del os.environ["SART_FORCE_GENERIC"]
    for k in ("N_RAYS", "N_REACHED_TELESCOPE", "N_SHELL_SELECTED", "N_HIT_NICKEL", "N_PASSED_TILL_WINDOW", "N_PASSED", "N_OUTSIDE_IMAGE"):
        assert s_a[k] == s_b[k], k
    assert s_a["SUM_WEIGHTS"] == pytest.approx(s_b["SUM_WEIGHTS"], rel=1e-12)
    np.testing.assert_allclose(img_a, img_b, rtol=1e-9, atol=img_b.max() * 1e-13)


@pytest.mark.parametrize("case", range(10))
def test_randomised_geometries_against_binary128_oracle(case):
    """Parity on setups nobody wrote by hand: magnet, pipes, telescope attitude / position, detector installation and chip
    size are perturbed at random around the reference's three installations.  Exercises the host-built zone table of stage
    A0 and the shell look-up table on geometries they were not tuned for: every flag of every ray must equal the binary128
    oracle's, the counters of the histogram path the oracle's counts."""
    from oracle.oracle import Oracle
    rng = np.random.default_rng(1000 + case)
    base = ["babyiaxo_xmm", "cast_llnl", "cast_abrixas", "babyiaxo_xmm_gas"][case % 4]
    full = make_setup(base)
    s = full.setup
    s.magnet_radiusCB *= rng.uniform(0.6, 1.3)
    s.magnet_lengthB *= rng.uniform(0.8, 1.1)
    s.magnet_lengthColdbore = s.magnet_lengthB * rng.uniform(1.01, 1.08)
    s.pipe_cb_vt3_radius *= rng.uniform(0.7, 1.4)
    s.pipe_vt3_xrt_radius *= rng.uniform(0.7, 1.4)
    s.pipe_cb_vt3_length *= rng.uniform(0.5, 2.0)
    s.pipe_vt3_xrt_length *= rng.uniform(0.5, 2.0)
    if rng.random() < 0.5:
        s.telescope_turned_x_deg = rng.uniform(-0.05, 0.05)
        s.telescope_turned_y_deg = rng.uniform(-0.05, 0.05)
    if rng.random() < 0.5:
        s.optics_entrance[0] += rng.uniform(-5.0, 5.0)
        s.optics_entrance[1] += rng.uniform(-5.0, 5.0)
    s.lateral_shift = rng.uniform(-1.0, 1.0)
    s.transversal_shift = rng.uniform(-1.0, 1.0)
    s.distance_detector_xrt *= rng.uniform(0.97, 1.03)
    s.chip_x_max = s.chip_y_max = rng.uniform(10.0, 40.0)
    n = 40_000
    seed, off = 900 + case, int(rng.integers(0, 10_000))
    with sa.RayTracer(full) as rt:
        rec = rt.traceAxionWrapper(n, seed=seed, ray_id_offset=off)
        _, summ = rt.trace_histogram(n, seed=seed, ray_id_offset=off)
    o = Oracle(full, "q")
    ref = o.trace_records(n, seed=seed, ray_id_offset=off)
    compare_records(rec, _as_gold(ref), 1e-9, 2e-8, 0.0)
    for f in ("passed", "passedTillWindow", "hitNickel"):
        np.testing.assert_array_equal(rec[f], ref[f])
    assert summ["N_PASSED"] == int(ref["passed"].sum())
    assert summ["N_PASSED_TILL_WINDOW"] == int(ref["passedTillWindow"].sum())
    assert summ["N_HIT_NICKEL"] == int(ref["hitNickel"].sum())
    _, osum, _ = Oracle(full, "f64").trace_histogram(n, seed=seed, ray_id_offset=off)
    for k in ("N_REACHED_TELESCOPE", "N_SHELL_SELECTED"):
        assert abs(summ[k] - osum[k]) <= 2, (k, summ[k], osum[k])      # f64 oracle: a ray within its noise of an edge may flip


@pytest.mark.parametrize("hole_type,n_holes,size", [("HT_CIRCLE", 1, 20.0), ("HT_CROSS", 5, 3.0), ("HT_STAR", 3, 4.0),
                                                    ("HT_SQUARE", 1, 25.0), ("HT_DIAMOND", 5, 15.0)])
def test_hole_loop_of_the_xmm_inner_disc(hole_type, n_holes, size):
    """The hole loop of lineIntersectsOpaqueTelescopeStructures (:1675-1688, lineIntersectsObject :494-527) only runs for an
    XMM-type optic with a hole type other than htNone — no shipped setup has one, so it is exercised here with each type:
    rays through a hole in the inner disc go on (and then miss every shell), all others behave as before."""
    from oracle.oracle import Oracle
    full = make_setup("babyiaxo_xmm")
    full.setup.hole_type = getattr(L, hole_type)
    full.setup.number_of_holes = n_holes
    full.setup.hole_in_optics = size
    n = 60_000
    with sa.RayTracer(full) as rt:
        rec = rt.traceAxionWrapper(n, seed=31)
        _, summ = rt.trace_histogram(n, seed=31)
    with sa.RayTracer(make_setup("babyiaxo_xmm")) as rt:
        _, closed = rt.trace_histogram(n, seed=31)
    assert summ["N_SHELL_SELECTED"] > closed["N_SHELL_SELECTED"]      # some rays do go through the hole(s)
    ref = Oracle(full, "q").trace_records(n, seed=31)
    compare_records(rec, _as_gold(ref), 1e-9, 2e-8, 0.0)
    for f in ("passed", "passedTillWindow", "hitNickel"):
        np.testing.assert_array_equal(rec[f], ref[f])
    assert summ["N_PASSED"] == int(ref["passed"].sum())


@pytest.mark.parametrize("variant", ["divergent", "divergent_collimated", "cast_parallel_on_axis", "flags_all_ignored"])
def test_xray_test_source_variants(variant):
    """The branches of the X-ray test source (:1765-1806) the shipped defaults do not reach: a divergent source (points on
    the source disc joined to points on the bore exit, :1794-1797), the collimator cut (:1800), the CAST source moved onto
    the axis, and every ignore-flag at once."""
    from oracle.oracle import Oracle
    src = L.TestSourceConfig()
    src.active, src.activity = 1, 0.125
    flags = L.CF_XRAY_TEST
    if variant == "divergent":
        src.parallel, src.energy, src.distance, src.radius, src.lengthCol = 0, 2.5, 1.0e6, 100.0, 0.0
        full = sa.initFullSetup(flags=flags, source_cfg=src, **_small())
    elif variant == "divergent_collimated":
        src.parallel, src.energy, src.distance, src.radius, src.lengthCol = 0, 1.0, 5.0e5, 300.0, 2.5e5
        full = sa.initFullSetup(flags=flags, source_cfg=src, **_small())
    elif variant == "cast_parallel_on_axis":
        src.parallel, src.energy, src.distance, src.radius, src.lengthCol = 1, 3.0, 100.0, 10.0, 50.0
        full = sa.initFullSetup(L.ES_CAST, L.DK_INGRID2018, L.SK_VACUUM, L.TK_LLNL, flags=flags, source_cfg=src, **_small())
    else:
        flags |= L.CF_IGNORE_DET_WINDOW | L.CF_IGNORE_GAS_ABS | L.CF_IGNORE_CONV_PROB | L.CF_IGNORE_REFLECTION
        src.parallel, src.energy, src.distance, src.radius, src.lengthCol = 1, 4.0, 2000.0, 350.0, 0.0
        full = sa.initFullSetup(flags=flags, source_cfg=src, **_small())
    n = 50_000
    with sa.RayTracer(full) as rt:
        rec = rt.traceAxionWrapper(n, seed=77, flags=flags)
        _, summ = rt.trace_histogram(n, seed=77, flags=flags)
    ref = Oracle(full, "q").trace_records(n, seed=77, flags=flags)
    assert ref["passed"].sum() > 500, int(ref["passed"].sum())          # the variant does send rays to the detector
    compare_records(rec, _as_gold(ref), 1e-9, 2e-8, 0.0)
    for f in ("passed", "passedTillWindow", "hitNickel"):
        np.testing.assert_array_equal(rec[f], ref[f])
    assert summ["N_PASSED"] == int(ref["passed"].sum())
    if variant == "flags_all_ignored":
        np.testing.assert_allclose(rec["weights"][rec["passed"] == 1], 1.0, rtol=1e-5)   # only cos(yaw) is left (:1598)


@pytest.mark.parametrize("case", range(6))
def test_randomised_detector_side_against_binary128_oracle(case):
    """Same idea for the detector side: detector kind (window angle 30 / 20 degrees), number of strongback strips and open
    aperture (calcWindowVals), window radius, distance window - focal plane, pipe turn angle (LLNL), detector depth."""
    import ctypes as C
    from oracle.oracle import Oracle
    rng = np.random.default_rng(2000 + case)
    tel, exp = [(L.TK_XMM, L.ES_BABYIAXO), (L.TK_LLNL, L.ES_CAST), (L.TK_ABRIXAS, L.ES_CAST)][case % 3]
    det = [L.DK_INGRID2017, L.DK_INGRID2018, L.DK_INGRIDIAXO][int(rng.integers(0, 3))]
    full = sa.initFullSetup(exp, det, L.SK_VACUUM, tel, **_small())
    s = full.setup
    s.radius_window = rng.uniform(4.0, 9.0)
    s.number_of_strips = int(rng.integers(2, 9))
    s.open_aperture_ratio = rng.uniform(0.7, 0.95)
    w, d = C.c_double(), C.c_double()
    assert L.load_host().sart_host_calc_window_vals(s.radius_window, s.number_of_strips, s.open_aperture_ratio, C.byref(w), C.byref(d)) == 0
    s.strip_width_window, s.strip_dist_window = w.value, d.value
    s.distance_window_focal_plane = rng.uniform(-5.0, 5.0)
    s.depth_det = rng.uniform(10.0, 40.0)
    if tel == L.TK_LLNL:
        s.pipes_turned_deg = rng.uniform(2.0, 3.5)
    n, seed = 40_000, 300 + case
    with sa.RayTracer(full) as rt:
        rec = rt.traceAxionWrapper(n, seed=seed)
        _, summ = rt.trace_histogram(n, seed=seed)
    ref = Oracle(full, "q").trace_records(n, seed=seed)
    assert ref["passedTillWindow"].sum() > 1000
    compare_records(rec, _as_gold(ref), 1e-9, 2e-8, 0.0)
    for f in ("passed", "passedTillWindow", "hitNickel", "kindsWindow"):
        np.testing.assert_array_equal(rec[f], ref[f])
    assert summ["N_PASSED"] == int(ref["passed"].sum()) and summ["N_PASSED_TILL_WINDOW"] == int(ref["passedTillWindow"].sum())
    if ref["passed"].sum() > 2000:    # (a shifted window may sit beside the focus)
        assert len(np.unique(ref["kindsWindow"][ref["passed"] == 1])) == 2   # both materials (strongback, window) occur


@pytest.mark.parametrize("case", range(5))
def test_randomised_gas_stage_against_binary128_oracle(case):
    """Gas stage (computeMagnetTransmission :1599-1625, axionMassforMagnet.nim:4-113) with random pressure, temperature,
    field, axion mass and coupling: conversion probability and absorption through the per-energy tables."""
    from oracle.oracle import Oracle
    rng = np.random.default_rng(3000 + case)
    full = make_setup("babyiaxo_xmm_gas")
    s = full.setup
    s.magnet_pGasRoom = rng.uniform(0.2, 3.0)
    s.magnet_tGas = rng.uniform(1.7, 293.0)
    s.magnet_B = rng.uniform(1.0, 9.0)
    s.m_axion = rng.uniform(0.0, 0.03)
    s.g_agamma = 10.0 ** rng.uniform(-13.0, -10.0)
    n, seed = 40_000, 500 + case
    with sa.RayTracer(full) as rt:
        rec = rt.traceAxionWrapper(n, seed=seed)
        _, summ = rt.trace_histogram(n, seed=seed)
        rt.set_axion_mass(0.5 * s.m_axion)                    # the scan entry point re-hoists the per-energy tables
        rec2 = rt.traceAxionWrapper(n, seed=seed)
    o = Oracle(full, "q")
    ref = o.trace_records(n, seed=seed)
    compare_records(rec, _as_gold(ref), 1e-9, 2e-8, 0.0)
    assert summ["SUM_WEIGHTS"] == pytest.approx(float(ref["weights"][ref["passed"] == 1].sum()), rel=1e-9)
    s.m_axion *= 0.5
    ref2 = Oracle(full, "q").trace_records(n, seed=seed)
    compare_records(rec2, _as_gold(ref2), 1e-9, 2e-8, 0.0)
    both = (ref["passed"] == 1) & (ref2["passed"] == 1)
    assert not np.allclose(ref["weights"][both], ref2["weights"][both], rtol=1e-6)    # the mass does matter


def test_interleaved_contexts_and_extreme_seeds_and_ids():
    """Two contexts with different setups used alternately give what each gives alone (no shared state between contexts);
    64-bit seeds and ray ids are used in full (seed 2^64 - 1, ids beyond 2^63, offsets that are not multiples of 4 or 256)."""
    from oracle.oracle import Oracle
    fa, fb = make_setup("babyiaxo_xmm"), make_setup("cast_llnl")
    seed, off = 2 ** 64 - 1, 2 ** 63 + 123
    with sa.RayTracer(fa) as ra, sa.RayTracer(fb) as rb:
        ia1, sa1 = ra.trace_histogram(300_000, seed=seed, ray_id_offset=off)
        ib1, sb1 = rb.trace_histogram(300_000, seed=seed, ray_id_offset=off)
        ia2, sa2 = ra.trace_histogram(300_000, seed=seed, ray_id_offset=off)
        rec = ra.traceAxionWrapper(20_001, seed=seed, ray_id_offset=off + 77)
        ib2, sb2 = rb.trace_histogram(300_000, seed=seed, ray_id_offset=off)
    for k in ("N_PASSED", "N_REACHED_TELESCOPE", "N_SHELL_SELECTED", "N_HIT_NICKEL"):
        assert sa1[k] == sa2[k] and sb1[k] == sb2[k]
    np.testing.assert_allclose(ia1, ia2, rtol=1e-10, atol=ia1.max() * 1e-14)
    np.testing.assert_allclose(ib1, ib2, rtol=1e-10, atol=ib1.max() * 1e-14)
    assert sa1["N_PASSED"] != sb1["N_PASSED"]
    ref = Oracle(fa, "q").trace_records(20_001, seed=seed, ray_id_offset=off + 77)
    compare_records(rec, _as_gold(ref), 1e-9, 2e-8, 0.0)
    np.testing.assert_array_equal(rec["passed"], ref["passed"])
    _, so, _ = Oracle(fa).trace_histogram(300_000, seed=seed, ray_id_offset=off)
    assert abs(sa1["N_PASSED"] - so["N_PASSED"]) <= 2 and abs(sa1["N_REACHED_TELESCOPE"] - so["N_REACHED_TELESCOPE"]) <= 2


@pytest.mark.parametrize("host_loop", [False, True])
def test_scan_driver_sharded_over_two_ranks_equals_one_process(tmp_path, host_loop):
    """tools/scan.py mass: the fused scan (every rank traces its share of the ray ids once for all masses, one reduce of the
    scan accumulator; BASELINE configs[4]) and, --host-loop, the reference-shaped scan (a re-trace per mass point, sharded by
    bin in one process / by ray id over the ranks with one accumulator reduce per point) - two gloo ranks on this one GPU give
    the curve of the single process."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    a, b = str(tmp_path / "one.csv"), str(tmp_path / "two.csv")
    common = ["mass", "--points", "4", "--rays", "500000"] + (["--host-loop"] if host_loop else [])
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "scan.py")] + common + ["--out", a], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    env = dict(os.environ, SART_BENCH_BACKEND="gloo", SART_BENCH_DEVICE="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29571", os.path.join(root, "tools", "scan.py")] + common + ["--shard", "rays", "--out", b],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    ca, cb = np.loadtxt(a, delimiter=",", skiprows=1, usecols=(0, 1, 2)), np.loadtxt(b, delimiter=",", skiprows=1, usecols=(0, 1, 2))
    np.testing.assert_allclose(cb[:, 1], ca[:, 1], rtol=1e-9)
    assert ca[:, 1].min() > 0


def test_scan_driver_xray_test_source_fused_equals_the_loop_on_the_same_rays(tmp_path):
    """tools/scan.py angular --xrayTest (the parallel test source in front of the bore, raytracer.nim:1765-1806: SURVEY 8(d)'s
    cleaner effective-area probe): the fused scan and the re-trace per angle give the same falling curve on a common first bin."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    a, b = str(tmp_path / "loop.csv"), str(tmp_path / "fused.csv")
    common = ["angular", "--xrayTest", "--numAngularScanPoints", "5", "--angularScanMax", "0.2", "--rays", "400000"]
    for extra, out in (([], a), (["--fused", "--shard", "rays"], b)):
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "scan.py")] + common + extra + ["--out", out], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
    ca, cb = np.loadtxt(a, delimiter=",", skiprows=1, usecols=(0, 1)), np.loadtxt(b, delimiter=",", skiprows=1, usecols=(0, 1))
    assert ca[0, 1] > 0 and (np.diff(ca[:, 1]) < 0).all() and (np.diff(cb[:, 1]) < 0).all()          # effective area falls off axis
    # the loop traces fresh rays per angle, the fused scan the same rays for all: equal within the Monte-Carlo error of 4e5 rays
    np.testing.assert_allclose(cb[:, 1] / cb[0, 1], ca[:, 1] / ca[0, 1], atol=0.02)


def test_setup_change_between_async_launches_needs_no_explicit_sync():
    """trace_histogram_device is asynchronous; changing the setup / axion mass / telescope angles right after it
    re-uploads tables the running launch still reads.  The library orders the two itself (refresh_derived / sync_blob wait
    for the context's stream), so each launch sees exactly the setup that was current when it was queued."""
    import torch
    full = make_setup("babyiaxo_xmm_gas")
    n = 20_000_000
    dev = torch.device("cuda", 0)
    with sa.RayTracer(full) as rt:
        # reference values, fully synchronised
        _, s_a = rt.trace_histogram(n, seed=5)
        rt.set_axion_mass(0.5 * full.setup.m_axion)
        _, s_b = rt.trace_histogram(n, seed=5)
        rt.set_axion_mass(full.setup.m_axion)
        s2 = full.setup.copy()
        s2.magnet_B = 1.5 * full.setup.magnet_B
        # now back to back without any synchronisation in between
        acc = [torch.zeros(sa.accumulator_len(256), dtype=torch.float64, device=dev) for _ in range(3)]
        p = rt.trace_params(n, seed=5)
        rt.trace_histogram_device(p, acc[0].data_ptr())
        rt.set_axion_mass(0.5 * full.setup.m_axion)               # parameter blob
        rt.trace_histogram_device(p, acc[1].data_ptr())
        L.check(rt.lib.sart_set_setup(rt.handle, C.byref(s2)))    # shell / energy / reflectivity tables re-hoisted
        rt.trace_histogram_device(p, acc[2].data_ptr())
        rt.synchronize()
        k = 256 * 256 + L.ACC["SUM_WEIGHTS"]
        got = [float(a[k].item()) for a in acc]
    assert got[0] == pytest.approx(s_a["SUM_WEIGHTS"], rel=1e-12)
    assert got[1] == pytest.approx(s_b["SUM_WEIGHTS"], rel=1e-12)
    assert got[0] != pytest.approx(got[1], rel=1e-3)
    assert got[2] != pytest.approx(got[0], rel=1e-3) and got[2] > 0


def test_scan_drivers_leave_the_context_as_they_found_it():
    """performAngularScan / the mass scan work on a copy of fullSetup in the reference (raytracer.nim:2794-2797);
    here the context is restored, so the same tracer gives the same answer before and after a scan."""
    full = make_setup("babyiaxo_xmm_gas")
    with sa.RayTracer(full) as rt:
        _, before = rt.trace_histogram(300_000, seed=8)
        sa.performAngularScan(rt, 0.0, 0.2, 3, n_rays_per_angle=50_000, seed=1)
        sa.performAxionMassScan(rt, [0.001, 0.02], 50_000, seed=1)
        got = L.Setup()
        L.check(rt.lib.sart_get_setup(rt.handle, C.byref(got)))
        assert got.telescope_turned_y_deg == full.setup.telescope_turned_y_deg and got.m_axion == full.setup.m_axion
        _, after = rt.trace_histogram(300_000, seed=8)
    assert after["SUM_WEIGHTS"] == pytest.approx(before["SUM_WEIGHTS"], rel=1e-13) and after["N_PASSED"] == before["N_PASSED"]


def test_oversized_grid_and_image_limits(monkeypatch):
    """more workgroups per CU than the partial-sum buffer used to hold (SART_HIST_BLOCKS_PER_CU) gives the same
    counts; an image with 2^29 pixels or more is rejected (32-bit pixel offsets on the device)."""
    full = make_setup("babyiaxo_xmm")
    with sa.RayTracer(full) as rt:
        _, ref = rt.trace_histogram(3_000_000, seed=2)
        p = rt.trace_params(1000)
        p.image_nx, p.image_ny = 1 << 15, 1 << 14
        img = np.empty(1)
        rc = rt.lib.sart_trace_histogram(rt.handle, C.byref(p), None, None)
        assert rc == L.SART_ERR_INVALID_ARGUMENT and b"2^29" in rt.lib.sart_last_error()
    monkeypatch.setenv("SART_HIST_BLOCKS_PER_CU", "20")
    with sa.RayTracer(full) as rt:
        _, s = rt.trace_histogram(3_000_000, seed=2)
    for k in ("N_RAYS", "N_REACHED_TELESCOPE", "N_SHELL_SELECTED", "N_PASSED", "N_HIT_NICKEL"):
        assert s[k] == ref[k], k
    assert s["SUM_WEIGHTS"] == pytest.approx(ref["SUM_WEIGHTS"], rel=1e-12)


@pytest.mark.parametrize("shape", [(256, 256), (40, 30), (1, 14000), (3000, 3000)])
def test_lds_image_tile_equals_global_atomics(shape, monkeypatch):
    """Small focal spots (CAST / LLNL; stage A0 off) accumulate the centre of the spot in a per-workgroup LDS tile that is
    placed by a pilot launch and flushed at the end of the kernel.  Same rays with the tile switched off (SART_NO_IMAGE_TILE):
    identical counters, images equal up to summation order — for the standard binning, for images smaller than the tile, for a
    one-column image (y-slice histogram) and for the 3000 x 3000 maps of generateResultPlots."""
    nx, ny = shape
    full = make_setup("cast_llnl_gold")
    n = 3_000_000
    with sa.RayTracer(full) as rt:
        img_a, s_a = rt.trace_image(n, nx, ny, seed=12)
        img_a2, s_a2 = rt.trace_image(n, nx, ny, seed=12)          # second call: tile position comes from the cache
    monkeypatch.setenv("SART_NO_IMAGE_TILE", "1")
    with sa.RayTracer(full) as rt:
        img_b, s_b = rt.trace_image(n, nx, ny, seed=12)
    for k in ("N_RAYS", "N_REACHED_TELESCOPE", "N_SHELL_SELECTED", "N_HIT_NICKEL", "N_PASSED_TILL_WINDOW", "N_PASSED", "N_OUTSIDE_IMAGE"):
        assert s_a[k] == s_b[k] == s_a2[k], k
    assert s_a["N_PASSED"] > 0.8 * n
    scale = img_b.max()
    np.testing.assert_allclose(img_a, img_b, rtol=1e-10, atol=scale * 1e-13)
    np.testing.assert_allclose(img_a2, img_b, rtol=1e-10, atol=scale * 1e-13)
    assert img_a.sum() == pytest.approx(s_a["SUM_WEIGHTS"], rel=1e-11)
    # the setup moves the spot: the tile is re-placed (cache invalidated with the parameter blob), results stay right
    with sa.RayTracer(full) as rt:
        rt.trace_image(200_000, nx, ny, seed=1)
        rt.set_telescope_angles(turned_y_deg=0.02)
        img_c, s_c = rt.trace_image(n, nx, ny, seed=12)
    monkeypatch.delenv("SART_NO_IMAGE_TILE")
    with sa.RayTracer(full) as rt:
        rt.trace_image(200_000, nx, ny, seed=1)
        rt.set_telescope_angles(turned_y_deg=0.02)
        img_d, s_d = rt.trace_image(n, nx, ny, seed=12)
    assert s_c["N_PASSED"] == s_d["N_PASSED"]
    np.testing.assert_allclose(img_d, img_c, rtol=1e-10, atol=max(img_c.max(), 1e-300) * 1e-13)


@pytest.mark.gpu
@pytest.mark.parametrize("tables", ["small", "default"])
def test_constant_path_variant_equals_the_carried_path(tables, monkeypatch):
    """BabyIAXO-type setups (bore much wider than the pipes): no ray that survives the cuts behind the magnet entered
    through the bore wall (sart_api.hip: path_is_constant), so kernel variant 5 drops the path column of ring 1 and keeps an
    LDS image tile in its place while stage A0 runs.  Same rays through variant 0 (SART_NO_PATH_CONST) and through variant 5
    with the tile off: identical counters, fluxes to 1e-12, images equal up to summation order; and the record path (which
    carries the path of every ray) says the same about the rays: pathCB is not a record field, but transmissionMagnet =
    cos(ya) conv_k pathCB^2 is, and it equals the constant-path value for every surviving ray."""
    full = make_setup("babyiaxo_xmm") if tables == "small" else sa.initFullSetup()
    n = 4_000_000
    with sa.RayTracer(full) as rt:
        img_a, s_a = rt.trace_histogram(n, seed=77)
    monkeypatch.setenv("SART_NO_IMAGE_TILE", "1")
    with sa.RayTracer(full) as rt:
        img_t, s_t = rt.trace_histogram(n, seed=77)
    monkeypatch.delenv("SART_NO_IMAGE_TILE")
    monkeypatch.setenv("SART_NO_PATH_CONST", "1")
    with sa.RayTracer(full) as rt:
        img_b, s_b = rt.trace_histogram(n, seed=77)
        rec = rt.traceAxionWrapper(200_000, seed=77)
    for k in ("N_RAYS", "N_REACHED_TELESCOPE", "N_SHELL_SELECTED", "N_HIT_NICKEL", "N_PASSED_TILL_WINDOW", "N_PASSED", "N_OUTSIDE_IMAGE"):
        assert s_a[k] == s_b[k] == s_t[k], k
    assert s_a["N_PASSED"] > 0.2 * n
    for k in ("SUM_WEIGHTS", "SUM_WEIGHTS_SQ", "SUM_X", "SUM_Y", "SUM_R"):
        assert s_a[k] == pytest.approx(s_b[k], rel=1e-12) and s_t[k] == pytest.approx(s_b[k], rel=1e-12), k
    scale = img_b.max()
    np.testing.assert_allclose(img_a, img_b, rtol=1e-10, atol=scale * 1e-13)
    np.testing.assert_allclose(img_t, img_b, rtol=1e-10, atol=scale * 1e-13)
    assert img_a.sum() == pytest.approx(s_a["SUM_WEIGHTS"], rel=1e-11)
    # records: transmissionMagnet / cos(yaw) = conv_k pathCB^2 with pathCB = lengthB sqrt(1 + slope^2) for every survivor
    ok = rec["passedTillWindow"] != 0
    assert ok.sum() > 10_000
    tm = rec["transmissionMagnet"][ok] / np.cos(rec["yawAngles"][ok])
    assert tm.max() / tm.min() < 1.0 + 1e-4          # (1 + slope^2) varies by < 3e-5; a wall entry would shorten the path by per cents


@pytest.mark.parametrize("mode", ["f64", "fixed64"])
def test_rays_outside_a_small_image_are_counted_like_the_oracles_records_say(mode):
    """N_OUTSIDE_IMAGE for an image that covers a part of the focal spot only (prepareHeatmap drops such rays, :838-842): the
    count of passed rays whose position lies outside, from the oracle's records of the same ray ids.  (Round 5: the count used
    to be added as a wave-uniform number inside the region that runs under the passed lanes' mask, and was short by the passes
    in which lane 0's own ray had not passed - invisible while every test's image covered the chip.)"""
    from oracle.oracle import Oracle
    full = make_setup("babyiaxo_xmm")
    n, seed = 300_000, 12
    x_range, y_range = (6.2, 7.4), (6.9, 7.9)
    with sa.RayTracer(full) as rt:
        rt.set_accumulation_mode(mode)
        img, s = rt.trace_image(n, 24, 20, x_range=x_range, y_range=y_range, seed=seed)
        whole = rt.trace_histogram(n, seed=seed)[1]
    rec = Oracle(full, "q").trace_records(n, seed=seed)
    p = rec["passed"] == 1
    x, y = rec["pointdataX"][p], rec["pointdataY"][p]
    inside = (x >= x_range[0]) & (x < x_range[1]) & (y >= y_range[0]) & (y < y_range[1])
    assert s["N_PASSED"] == whole["N_PASSED"] == p.sum()
    assert 0.05 * p.sum() < (~inside).sum() < 0.95 * p.sum()
    assert abs(s["N_OUTSIDE_IMAGE"] - (~inside).sum()) <= 2          # (a ray within 1e-10 mm of the image's edge)
    assert img.sum() == pytest.approx(rec["weights"][p][inside].sum(), rel=1e-6)
    assert whole["N_OUTSIDE_IMAGE"] == 0


def test_bench_rank0_failing_behind_the_reduce_ends_the_self_launched_run():
    """only rank 0 finalizes and can raise behind the reduce (bench.py); the other rank walks into the next
    all_reduce.  `bench.py --gpus 2` as its own launcher, two gloo ranks on this GPU, rank 0 made to fail there: the run ends
    within seconds with a non-zero exit, no JSON line, and the launcher says which rank failed."""
    import os
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(SART_BENCH_BACKEND="gloo", SART_BENCH_DEVICE="0", SART_BENCH_FAIL_RANK0_AFTER_REDUCE="1")
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--rays-per-step", "2e7",
                          "--profile-run"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 1, (out.returncode, out.stderr[-2000:])
    assert time.time() - t0 < 300
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert "launcher: rank 0 exited with code 1; ending rank(s) 1" in out.stderr, out.stderr[-2000:]
    assert any(l.startswith("[rank 0]") and "SART_BENCH_FAIL_RANK0_AFTER_REDUCE" in l for l in out.stderr.splitlines())
    # the knob is a rehearsal knob: without the gloo backend it does nothing (one rank, RCCL never involved)
    env.pop("SART_BENCH_BACKEND")
    env.pop("SART_BENCH_DEVICE")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--rays-per-step", "2e7", "--profile-run"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
