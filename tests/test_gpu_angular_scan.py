"""Fused angular scan (include/sart.h: sart_trace_angular_scan; BASELINE configs[3]) on the MI355X box.

performAngularScan (raytracer.nim:2778-2802) re-runs the whole trace per telescope angle.  The angle enters a ray at the
transformation into the telescope's frame (:1878-1899) and nowhere before it, so the scan kernel samples every ray and takes it
through bore and pipes once and turns it through K angles.  Demanded here, per angle:
  * SART_ACCUM_FIXED64: the raw integers and all counters equal those of a single-angle launch (sart_set_telescope_angles +
    trace) on the same ray ids - bit for bit, for both kernel instantiations, for any split of the rays and of the angles;
  * SART_ACCUM_F64: the flux equals the single-angle launch to 1e-12 (summation order);
  * the flux equals the CPU oracle (80-bit build) with that angle to 1e-6, on small and on full-size tables;
and for the flux-only launches (image_nx = image_ny = 0) the host-loop scan uses: the scalars of an ordinary launch, bit for bit."""
import os

import numpy as np
import pytest

import solaraxionraytracing_amd as sa
from solaraxionraytracing_amd import _lib as L

from tests.conftest import make_setup

pytestmark = pytest.mark.gpu

N_IMG = 256 * 256


def angles(k, top=0.4):
    """k telescope angles in degrees, none of them zero (a single launch at angle 0 runs the unrotated kernel)."""
    return np.ascontiguousarray(np.linspace(0.02, top, k))


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


def raw_single(rt, torch, angle, n, seed, off=0, flags=None):
    """Scalars of the raw FIXED64 accumulator of one single-angle launch."""
    acc = torch.zeros(sa.accumulator_len(256), dtype=torch.int64, device="cuda")
    rt.set_telescope_angles(turned_y_deg=float(angle))
    p = rt.trace_params(n, seed=seed, ray_id_offset=off, flags=flags, accumulate=False)
    rt.trace_histogram_device(p, acc.data_ptr())
    rt.synchronize()
    return acc.cpu().numpy()[N_IMG:]


def raw_scan(rt, torch, an, pieces, seed, flags=None):
    acc = torch.zeros(sa.angular_scan_len(len(an)), dtype=torch.int64, device="cuda")
    for lo, hi in pieces:
        p = rt.trace_params(hi - lo, seed=seed, ray_id_offset=lo, flags=flags, accumulate=True)
        rt.trace_angular_scan_device(p, an, acc.data_ptr())
    rt.synchronize()
    return acc.cpu().numpy().reshape(len(an) + 1, L.ASCAN_ROW)


VARIANTS = {
    # specialised instantiation: solar source, vacuum, no hole loop (the single launches run histogram variant 4)
    "babyiaxo_xmm": ("babyiaxo_xmm", {}, None),
    # the same with a fixed tilt about x under the scanned angle
    "babyiaxo_xmm_turned_x": ("babyiaxo_xmm", {}, "turned_x"),
    # generic instantiation (switches read at run time; single launches: histogram variant 2)
    "generic": ("babyiaxo_xmm", {"SART_FORCE_GENERIC": "1"}, None),
    # gas stage: rotated + gas has no specialised kernel on either side
    "gas": ("babyiaxo_xmm_gas", {}, None),
    # X-ray test source: no stage A0 (no zones), one energy row
    "xray_test_source": ("babyiaxo_xmm_xray", {}, None),
    # cone optics, four coatings, a 43 mm bore that rays do enter through the wall
    "cast_llnl": ("cast_llnl", {}, None),
    # CAST + Abrixas (27 Wolter shells, six spokes)
    "cast_abrixas": ("cast_abrixas", {}, None),
    # stage A0 switched off: phase A runs on the rays as they come
    "no_early_reject": ("babyiaxo_xmm", {"SART_NO_EARLY_REJECT": "1"}, None),
}


def variant_setup(name):
    setup_name, knobs, tweak = VARIANTS[name]
    full = make_setup(setup_name)
    if tweak == "turned_x":
        full.setup.telescope_turned_x_deg = 0.03
    return full, knobs, (full.flags if setup_name.endswith("xray") else None)


def check_rows_equal_singles(scan, an, singles, n, label):
    shared = scan[len(an)]
    assert shared[L.ASCAN_SHARED["N_RAYS"]] == n
    for k, s in enumerate(singles):
        row = scan[k]
        for key in ("SUM_WEIGHTS", "SUM_WEIGHTS_SQ"):
            assert row[L.ASCAN[key]] == s[L.ACC[key]] and row[L.ASCAN_HI[key]] == s[L.ACC_HI[key]], (label, k, key, row, s)
        for key in ("N_PASSED", "N_SHELL_SELECTED", "N_HIT_NICKEL", "N_PASSED_TILL_WINDOW"):
            assert row[L.ASCAN[key]] == s[L.ACC[key]], (label, k, key, row[L.ASCAN[key]], s[L.ACC[key]])
        assert shared[L.ASCAN_SHARED["N_REACHED_TELESCOPE"]] == s[L.ACC["N_REACHED_TELESCOPE"]], (label, k)
        assert 0 <= row[L.ASCAN["SUM_WEIGHTS"]] < 2 ** L.FIXED_LIMB_BITS


@pytest.mark.parametrize("name", sorted(VARIANTS))
def test_fixed64_scan_equals_single_angle_launches_bit_for_bit(name):
    import torch
    full, knobs, flags = variant_setup(name)
    an = angles(35, top=1.2 if name.startswith("cast") else 0.4)   # two groups of angles: 18 + 17
    n, seed = 2_000_000, 17
    with env(**knobs):
        with sa.RayTracer(full) as rt:
            rt.set_accumulation_mode("fixed64")
            scan = raw_scan(rt, torch, an, [(0, n)], seed, flags)
            split = raw_scan(rt, torch, an, [(0, 700_001), (700_001, n)], seed, flags)   # rays in two accumulating calls
            y0 = rt.full.setup.telescope_turned_y_deg
            singles = [raw_single(rt, torch, a, n, seed, flags=flags) for a in an]
            rt.set_telescope_angles(turned_y_deg=y0)
    assert np.array_equal(scan, split)
    assert scan[:len(an), L.ASCAN["N_PASSED"]].max() > 1000
    check_rows_equal_singles(scan, an, singles, n, name)


def test_fixed64_scan_is_independent_of_the_grouping_of_the_angles():
    """32 angles per launch: 67 angles run as 23 + 22 + 22; every row equals the row of a scan of that angle alone."""
    import torch
    full = make_setup("babyiaxo_xmm")
    an = angles(67)
    n, seed = 1_000_000, 29
    with sa.RayTracer(full) as rt:
        rt.set_accumulation_mode("fixed64")
        scan = raw_scan(rt, torch, an, [(0, n)], seed)
        for k in (0, 22, 23, 44, 45, 66):
            one = raw_scan(rt, torch, an[k:k + 1], [(0, n)], seed)
            assert np.array_equal(one[0], scan[k]), k
            assert np.array_equal(one[1], scan[len(an)])


@pytest.mark.parametrize("flags", [L.CF_IGNORE_DET_WINDOW | L.CF_IGNORE_GAS_ABS, L.CF_IGNORE_REFLECTION,
                                   L.CF_IGNORE_DET_WINDOW | L.CF_IGNORE_REFLECTION | L.CF_IGNORE_GAS_ABS | L.CF_IGNORE_CONV_PROB])
def test_fixed64_scan_equals_single_angle_launches_under_the_ignore_flags(flags):
    import torch
    full = make_setup("babyiaxo_xmm")
    an = angles(5)
    n, seed = 1_000_000, 23
    with sa.RayTracer(full) as rt:
        rt.set_accumulation_mode("fixed64")
        scan = raw_scan(rt, torch, an, [(0, n)], seed, flags)
        singles = [raw_single(rt, torch, a, n, seed, flags=flags) for a in an]
    check_rows_equal_singles(scan, an, singles, n, flags)


def test_fixed64_scan_finalize_equals_finalized_single_launches():
    """The quanta are a function of (setup, tables, flags, headroom): finalize of the raw scan gives the very doubles the
    blocking single-angle call returns, and the blocking scan call returns them, too."""
    full = make_setup("babyiaxo_xmm")
    an = angles(7)
    n, seed = 2_000_000, 3
    with sa.RayTracer(full) as rt:
        rt.set_accumulation_mode("fixed64")
        per_angle, shared = rt.trace_angular_scan(an, n, seed=seed)
        for k, a in enumerate(an):
            rt.set_telescope_angles(turned_y_deg=float(a))
            s = rt.trace_histogram(n, seed=seed)[1]
            for key in ("SUM_WEIGHTS", "SUM_WEIGHTS_SQ", "N_PASSED", "N_SHELL_SELECTED", "N_HIT_NICKEL", "N_PASSED_TILL_WINDOW"):
                assert np.float64(per_angle[key][k]).view(np.uint64) == np.float64(s[key]).view(np.uint64), (k, key, per_angle[key][k], s[key])
        assert shared["N_RAYS"] == n
        assert np.all(np.isfinite(per_angle["SUM_WEIGHTS_SQ"]))


@pytest.mark.parametrize("name", ["babyiaxo_xmm", "generic", "gas", "cast_llnl"])
def test_f64_scan_equals_single_angle_launches(name):
    full, knobs, flags = variant_setup(name)
    an = np.concatenate([[0.0], angles(17, top=1.2 if name.startswith("cast") else 0.4)])   # with angle 0: the unrotated kernel on the other side
    n, seed = 2_000_000, 5
    with env(**knobs):
        with sa.RayTracer(full) as rt:
            per_angle, shared = rt.trace_angular_scan(an, n, seed=seed, flags=flags)
            for k, a in enumerate(an):
                rt.set_telescope_angles(turned_y_deg=float(a))
                s = rt.trace_histogram(n, seed=seed, flags=flags)[1]
                # (angle 0: the unrotated kernel's frame change is exact, the rotation by 0 rounds - a ray within 1e-13 mm of a cut may differ)
                slack = 2 if a == 0.0 else 0
                for key in ("N_PASSED", "N_SHELL_SELECTED", "N_HIT_NICKEL", "N_PASSED_TILL_WINDOW"):
                    assert abs(per_angle[key][k] - s[key]) <= slack, (name, k, key)
                assert shared["N_REACHED_TELESCOPE"] == s["N_REACHED_TELESCOPE"]
                if per_angle["N_PASSED"][k] == s["N_PASSED"]:
                    assert per_angle["SUM_WEIGHTS"][k] == pytest.approx(s["SUM_WEIGHTS"], rel=1e-12), (name, k)
                    assert per_angle["SUM_WEIGHTS_SQ"][k] == pytest.approx(s["SUM_WEIGHTS_SQ"], rel=1e-11), (name, k)


@pytest.mark.parametrize("name", ["babyiaxo_xmm", "xray_test_source", "cast_llnl", "gas"])
def test_scan_matches_the_oracle_per_angle(name):
    from oracle.oracle import Oracle
    full, knobs, flags = variant_setup(name)
    an = np.concatenate([[0.0], angles(6, top=1.0 if name.startswith("cast") else 0.3)])
    n, seed = 100_000, 4
    with sa.RayTracer(full) as rt:
        per_angle, shared = rt.trace_angular_scan(an, n, seed=seed, flags=flags)
    o = Oracle(full, "ld")
    for k, a in enumerate(an):
        s = full.setup.copy()
        s.telescope_turned_y_deg = float(a)
        want = o.trace_histogram(n, seed=seed, setup=s, flags=flags)[1]
        # (a ray within rounding of a cut may fall on the other side in the 80-bit oracle: then the flux differs by that ray)
        same_rays = per_angle["N_PASSED"][k] == want["N_PASSED"]
        assert abs(per_angle["N_PASSED"][k] - want["N_PASSED"]) <= 2
        assert per_angle["SUM_WEIGHTS"][k] == pytest.approx(want["SUM_WEIGHTS"], rel=1e-6 if same_rays else 1e-3), (name, k, a)
        for key in ("N_SHELL_SELECTED", "N_HIT_NICKEL", "N_PASSED_TILL_WINDOW"):
            assert abs(per_angle[key][k] - want[key]) <= 2, (name, k, key)
        assert abs(shared["N_REACHED_TELESCOPE"] - want["N_REACHED_TELESCOPE"]) <= 2


def test_scan_on_full_size_tables_matches_the_oracle_and_the_host_loop():
    """BASELINE configs[3] at its table sizes (1968 x 1500 CDFs, 1000 x 1000 reflectivity): 50 angles of the XMM off-axis curve
    through the C++ host driver (two launches of 25 angles), three of them against the 80-bit oracle and the
    single-angle launches."""
    from oracle.oracle import Oracle
    full = sa.initFullSetup()
    full.setup.chip_x_max = full.setup.chip_y_max = 100.0          # ChipXMax = 100 mm (raytracer.nim:262-264): the spot walks 65 mm at 0.5 deg
    flags = L.CF_IGNORE_DET_WINDOW | L.CF_IGNORE_GAS_ABS | L.CF_IGNORE_CONV_PROB   # the effective-area flags (cf. :2315)
    an = np.linspace(0.0, 0.5, 50)
    n, seed = 4_000_000, 11
    with sa.RayTracer(full) as rt:
        a_out, flux, rel, err, n_pass = sa.performAngularScan(rt, 0.0, 0.5, 50, n_rays_per_angle=n, seed=seed, flags=flags, fused=True, errors=True)
        assert np.array_equal(a_out, an) and rel.max() == 1.0 and int(np.argmax(flux)) <= 2
        assert np.all(err > 0) and np.all(err[:25] < 0.01 * flux[:25]) and np.all(err < 0.2 * flux)   # (few rays pass at 0.5 deg)
        for k in (7, 24, 49):
            rt.set_telescope_angles(turned_y_deg=float(an[k]))
            s = rt.trace_histogram(n, seed=seed, flags=flags)[1]
            assert flux[k] == pytest.approx(s["SUM_WEIGHTS"], rel=1e-12) and n_pass[k] == s["N_PASSED"], k
        rt.set_telescope_angles(turned_y_deg=0.0)
        n_o = 400_000
        per_angle, _ = rt.trace_angular_scan(an[[7, 24, 49]], n_o, seed=seed, flags=flags)
    o = Oracle(full, "ld")
    for j, k in enumerate((7, 24, 49)):
        s = full.setup.copy()
        s.telescope_turned_y_deg = float(an[k])
        want = o.trace_histogram(n_o, seed=seed, setup=s, flags=flags)[1]
        assert abs(per_angle["N_PASSED"][j] - want["N_PASSED"]) <= 2
        assert per_angle["SUM_WEIGHTS"][j] == pytest.approx(want["SUM_WEIGHTS"], rel=1e-6 if per_angle["N_PASSED"][j] == want["N_PASSED"] else 1e-4), k
    # the curve falls off with the angle (XMM's vignetting): what the reference plots against the McXtrace / XMM curves (:2803-2815)
    assert rel[49] < 0.6 * rel[0]


def test_scan_accumulation_errors_and_zero_rays():
    full = make_setup("babyiaxo_xmm")
    an = angles(5)
    with sa.RayTracer(full) as rt:
        pa, sh = rt.trace_angular_scan(an, 0, seed=2)
        assert not pa["SUM_WEIGHTS"].any() and sh["N_RAYS"] == 0
        for bad in (np.array([0.1, np.nan]), np.array([90.0]), np.array([])):
            with pytest.raises(L.SartError) as e:
                rt.trace_angular_scan(bad, 1000)
            assert e.value.code == L.SART_ERR_INVALID_ARGUMENT
        # the context's own angle is untouched by a scan, and a histogram launch afterwards runs the unrotated kernel as before
        before = rt.trace_histogram(300_000, seed=2)[1]
        rt.trace_angular_scan(an, 300_000, seed=2)
        after = rt.trace_histogram(300_000, seed=2)[1]
        assert before == after


@pytest.mark.parametrize("name", ["babyiaxo_xmm", "babyiaxo_xmm_rot", "cast_llnl", "babyiaxo_xmm_gas", "babyiaxo_xmm_xray"])
@pytest.mark.parametrize("mode", ["f64", "fixed64"])
def test_flux_only_launch_equals_the_scalars_of_an_image_launch(name, mode):
    """image_nx = image_ny = 0: no image, no LDS tile, no pilot launch - and the very same scalars (every passed ray counts as
    outside the image: N_OUTSIDE_IMAGE = N_PASSED).  (The X-ray test source asks for 64 image replicas: a flux-only launch must
    not follow it into a pilot launch on an image that does not exist - it did until round 5's tools/scan.py --xrayTest.)"""
    full = make_setup(name)
    n, seed = 1_500_000, 8
    with sa.RayTracer(full) as rt:
        rt.set_accumulation_mode(mode)
        img, want = rt.trace_histogram(n, seed=seed)
        got = rt.trace_flux(n, seed=seed)
    assert got["N_OUTSIDE_IMAGE"] == got["N_PASSED"] == want["N_PASSED"] > 1000
    for key in ("N_RAYS", "N_REACHED_TELESCOPE", "N_SHELL_SELECTED", "N_HIT_NICKEL", "N_PASSED_TILL_WINDOW"):
        assert got[key] == want[key], key
    for key in ("SUM_WEIGHTS", "SUM_WEIGHTS_SQ", "SUM_X", "SUM_Y", "SUM_R"):
        if mode == "fixed64":
            assert np.float64(got[key]).view(np.uint64) == np.float64(want[key]).view(np.uint64), key
        else:
            assert got[key] == pytest.approx(want[key], rel=1e-12), key


def test_host_loop_scan_uses_fresh_rays_per_angle_and_restores_the_setup():
    """sart_host_perform_angular_scan keeps the reference's shape (raytracer.nim:2791-2800): angle i on ray ids
    [offset + i n, offset + (i + 1) n), flux-only launches, the setup's own angle back afterwards."""
    full = make_setup("babyiaxo_xmm")
    an = angles(4)
    n, seed = 500_000, 6
    with sa.RayTracer(full) as rt:
        a_out, flux, rel = sa.performAngularScan(rt, 0, 0, angles=an, n_rays_per_angle=n, seed=seed, ray_id_offset=1000)
        assert rt.trace_flux(n, seed=seed) == rt.trace_flux(n, seed=seed)
        for i, a in enumerate(an):
            rt.set_telescope_angles(turned_y_deg=float(a))
            want = rt.trace_histogram(n, seed=seed, ray_id_offset=1000 + i * n)[1]["SUM_WEIGHTS"]
            assert flux[i] == pytest.approx(want, rel=1e-12)
    assert rel.max() == 1.0


def test_scan_cli_fused_writes_the_curve(tmp_path):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / "angle.csv")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "scan.py"), "angular", "--numAngularScanPoints", "9", "--rays", "1e6", "--angularScanMin", "0",
                        "--angularScanMax", "0.4", "--fused", "--out", out], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    rows = [l.split(",") for l in open(out).read().splitlines()]
    assert len(rows) == 10
    rel = np.array([float(x[2]) for x in rows[1:]])
    assert rel.max() == 1.0 and rel[-1] < rel[0]


@pytest.mark.parametrize("shard", ["bins", "rays"])
def test_fused_scan_driver_sharded_over_two_ranks_equals_one_process(tmp_path, shard):
    """tools/scan.py angular --fused (BASELINE configs[3]: angle bins over the GPUs): two gloo ranks on this one GPU - the
    angles dealt out to the ranks (--shard bins: every rank turns all ray ids through its group of angles), or the ray ids
    (--shard rays: one reduce of the scan accumulator) - give the curve of the single process."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    a, b = str(tmp_path / "one.csv"), str(tmp_path / "two.csv")
    common = ["angular", "--fused", "--numAngularScanPoints", "7", "--rays", "500000", "--angularScanMax", "0.3"]
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "scan.py")] + common + ["--out", a], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    env2 = dict(os.environ, SART_BENCH_BACKEND="gloo", SART_BENCH_DEVICE="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29573", os.path.join(root, "tools", "scan.py")] + common + ["--shard", shard, "--out", b],
                       env=env2, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    ca, cb = np.loadtxt(a, delimiter=",", skiprows=1, usecols=(0, 1, 2)), np.loadtxt(b, delimiter=",", skiprows=1, usecols=(0, 1, 2))
    np.testing.assert_allclose(cb[:, 1], ca[:, 1], rtol=1e-12)
    assert ca[:, 1].min() > 0 and ca[0, 1] > ca[-1, 1]


@pytest.mark.parametrize("mode", ["f64", "fixed64"])
def test_flux_only_launch_carries_the_spectra(mode):
    """image_nx = image_ny = 0 with params.spectra = 1: the accumulator is the 24 scalars followed by the radial and per-energy
    histograms - the post-processing of generateResultPlots without an image (and, in the integer mode, a conservation check whose
    pixel sum is all in SUM_WEIGHTS_OUTSIDE)."""
    full = make_setup("babyiaxo_xmm")
    n, seed = 1_000_000, 14
    with sa.RayTracer(full) as rt:
        rt.set_accumulation_mode(mode)
        img, s, spec = rt.trace_spectra(n, seed=seed, n_radial_bins=500, radial_max=10.0)
        img0, s0, spec0 = rt.trace_spectra(n, seed=seed, image_n=0, n_radial_bins=500, radial_max=10.0)
    assert img0.size == 0 and s0["N_OUTSIDE_IMAGE"] == s0["N_PASSED"] == s["N_PASSED"]
    for key in ("radial_counts", "energy_counts"):
        assert np.array_equal(spec0[key], spec[key]), key
    for key in ("radial_weights", "energy_weights", "energy_reflect"):
        if mode == "fixed64":
            assert np.array_equal(spec0[key], spec[key]), key
        else:
            np.testing.assert_allclose(spec0[key], spec[key], rtol=1e-11, atol=1e-30)
    assert spec0["radial_weights"].sum() == pytest.approx(s0["SUM_WEIGHTS"], rel=1e-12)


def test_an_unresolved_angle_reads_nan_in_its_own_row_only():
    """the FIXED64 scan has one quantum for all its angles (from the on-axis weight bound).  An angle far off axis can
    pass a few rays of tiny R1 R2; when the quantum does not resolve them (average below 2^12 quanta per passed ray) THAT row's
    SUM_WEIGHTS reads NaN - and nothing else: no SART_ERR_ACCUMULATOR for the scan, the other rows converted as ever, the
    counters of the unresolved row intact.  No setup at hand gets there by itself (24 angles across the edge of BabyIAXO's field of
    view at the coarsest quanta the library accepts stay resolved: the last rays that pass still average 2^-7 of the bound), so
    the finalize kernel is handed a raw accumulator with such a row; the MASS scan - a quantum per mass, from that mass's own
    bound - keeps reporting the same row as an error."""
    import torch
    full = make_setup("babyiaxo_xmm_gas")
    K = 3
    with sa.RayTracer(full) as rt:
        rt.set_accumulation_mode("fixed64")
        p = rt.trace_params(1000, seed=1)
        raw = np.zeros((K + 1, L.ASCAN_ROW), dtype=np.int64)
        raw[0, L.ASCAN["SUM_WEIGHTS"]], raw[0, L.ASCAN["SUM_WEIGHTS_SQ"]], raw[0, L.ASCAN["N_PASSED"]] = 10 ** 6 << 20, 10 ** 6 << 10, 10 ** 6
        raw[1, L.ASCAN["SUM_WEIGHTS"]], raw[1, L.ASCAN["SUM_WEIGHTS_SQ"]], raw[1, L.ASCAN["N_PASSED"]] = 1000 * 100, 1000, 1000    # 100 quanta per ray
        raw[1, L.ASCAN["N_SHELL_SELECTED"]], raw[1, L.ASCAN["N_HIT_NICKEL"]], raw[1, L.ASCAN["N_PASSED_TILL_WINDOW"]] = 5000, 7, 1200
        raw[2, L.ASCAN["N_SHELL_SELECTED"]] = 40                                                                                    # nothing passed: resolved, 0
        raw[K, L.ASCAN_SHARED["N_RAYS"]], raw[K, L.ASCAN_SHARED["N_REACHED_TELESCOPE"]] = 10 ** 7, 5 * 10 ** 6
        dev = torch.from_numpy(raw.ravel().copy()).cuda()
        out = torch.zeros(dev.numel(), dtype=torch.float64, device="cuda")
        rt.finalize_angular_scan_device(p, K, dev.data_ptr(), out.data_ptr())
        rt.synchronize()                                   # no SART_ERR_ACCUMULATOR: the unresolved row is not the scan's failure
        per, shared = sa.split_angular_scan(out.cpu().numpy(), K)
        q = per["SUM_WEIGHTS"][0] / float(10 ** 6 << 20)                        # the scan's quantum: a power of two
        assert q > 0 and np.frexp(q)[0] == 0.5 and np.isfinite(per["SUM_WEIGHTS_SQ"][0]) and per["N_PASSED"][0] == 10 ** 6
        assert np.isnan(per["SUM_WEIGHTS"][1]) and np.isnan(per["SUM_WEIGHTS_SQ"][1])
        assert (per["N_PASSED"][1], per["N_SHELL_SELECTED"][1], per["N_HIT_NICKEL"][1], per["N_PASSED_TILL_WINDOW"][1]) == (1000, 5000, 7, 1200)
        assert per["SUM_WEIGHTS"][2] == 0.0 and per["N_SHELL_SELECTED"][2] == 40
        assert shared["N_RAYS"] == 10 ** 7 and shared["N_REACHED_TELESCOPE"] == 5 * 10 ** 6
        # the same row in a mass scan (same row layout) is an error of the whole scan, as before
        mraw = np.zeros((K + 1, L.SCAN_ROW), dtype=np.int64)
        mraw[:K, :] = raw[:K, :]
        mraw[:K, L.ASCAN["N_SHELL_SELECTED"]] = mraw[:K, L.ASCAN["N_HIT_NICKEL"]] = mraw[:K, L.ASCAN["N_PASSED_TILL_WINDOW"]] = 0
        mdev = torch.from_numpy(mraw.ravel().copy()).cuda()
        rt.finalize_mass_scan_device(p, np.array([0.0, 0.008, 0.02]), mdev.data_ptr(), out.data_ptr())
        with pytest.raises(L.SartError) as e:
            rt.synchronize()
        assert e.value.code == L.SART_ERR_ACCUMULATOR


def test_scan_across_the_edge_of_the_field_of_view_stays_resolved_at_the_coarsest_quanta():
    """The natural version of the case above: on axis, nearly on axis and 24 angles from 0.4 to 0.7 deg (nothing passes beyond),
    headroom 44 (quantum = 2^-19 of the weight bound).  Every row that passes rays is a number close to the f64 scan's, every
    counter exact, no error."""
    full = make_setup("babyiaxo_xmm_rot")
    full.setup.telescope_turned_x_deg = 0.0
    an = np.concatenate([[0.0, 0.05], np.linspace(0.4, 0.7, 24)])
    n = 4_000_000
    with sa.RayTracer(full) as rt:
        ref, _ = rt.trace_angular_scan(an, n, seed=9)
        rt.set_accumulation_mode("fixed64", 44)
        per, shared = rt.trace_angular_scan(an, n, seed=9)
    assert shared["N_RAYS"] == n and not np.isnan(per["SUM_WEIGHTS"]).any()
    for k in ("N_PASSED", "N_SHELL_SELECTED", "N_HIT_NICKEL", "N_PASSED_TILL_WINDOW"):
        np.testing.assert_array_equal(per[k], ref[k])
    assert ref["N_PASSED"][-1] == 0 and ref["N_PASSED"][2] > 0
    np.testing.assert_allclose(per["SUM_WEIGHTS"], ref["SUM_WEIGHTS"], rtol=2e-3)


def test_angles_beyond_45_degrees_are_accepted_and_90_is_refused():
    """the fused scan used to refuse |angle| > 45 deg, which the host loop traces.  Now any angle inside (-90, 90) runs
    (nothing passes at such tilts; the counters equal the single launch's), 90 deg and non-finite angles are invalid arguments."""
    import torch
    full = make_setup("babyiaxo_xmm_rot")
    an = np.array([0.1, 50.0, -80.0])
    with sa.RayTracer(full) as rt:
        rt.set_accumulation_mode("fixed64")
        scan = raw_scan(rt, torch, an, [(0, 300_000)], seed=3)
        for k, a in enumerate(an):
            one = raw_single(rt, torch, a, 300_000, seed=3)
            assert scan[k, L.ASCAN["N_PASSED"]] == one[L.ACC["N_PASSED"]] and scan[k, L.ASCAN["N_SHELL_SELECTED"]] == one[L.ACC["N_SHELL_SELECTED"]]
            assert scan[k, L.ASCAN["SUM_WEIGHTS"]] == one[L.ACC["SUM_WEIGHTS"]]
        assert scan[0, L.ASCAN["N_PASSED"]] > 0 and scan[1, L.ASCAN["N_PASSED"]] == 0
        for bad in (90.0, -90.0, float("nan"), float("inf")):
            with pytest.raises(L.SartError) as e:
                rt.trace_angular_scan(np.array([0.1, bad]), 1000, seed=3)
            assert e.value.code == L.SART_ERR_INVALID_ARGUMENT
