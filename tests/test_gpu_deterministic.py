"""SART_ACCUM_FIXED64 — the deterministic accumulation mode (include/sart.h "accumulation mode"; run on the MI355X box).

The reference adds f64 weights into its heat map one after the other on the CPU (prepareHeatmap, raytracer.nim:838-842; flux
sum :2800).  On the GPU the order in which f64 atomics retire depends on the number of GPUs, on the replica / LDS-tile
placement and on how the rays are split over launches, so f64 images agree only to ~1e-13.  In FIXED64 mode every ray adds
rint(weight / quantum) as an integer: the tests below demand BITWISE equal images and scalars under all of those changes,
and agreement with the f64 mode to 1e-12 of the largest pixel."""
import os

import numpy as np
import pytest

import solaraxionraytracing_amd as sa
from solaraxionraytracing_amd import _lib as L

pytestmark = pytest.mark.gpu

N = 20_000_000
COUNTERS = ("N_RAYS", "N_PASSED", "N_PASSED_TILL_WINDOW", "N_HIT_NICKEL", "N_REACHED_TELESCOPE", "N_SHELL_SELECTED", "N_OUTSIDE_IMAGE")
SUMS = ("SUM_WEIGHTS", "SUM_X", "SUM_Y", "SUM_R", "SUM_WEIGHTS_SQ")

_cache = {}


def full_setup(name):
    if name not in _cache:
        if name == "babyiaxo_xmm":        # constant-path variant: stage A0 + LDS tile in ring 1's path column, 8 replicas
            _cache[name] = sa.initFullSetup()
        elif name == "cast_llnl_gold":    # no stage A0: LDS tile in ring 0, 64 replicas
            _cache[name] = sa.initFullSetup(L.ES_CAST, L.DK_INGRID2018, L.SK_VACUUM, L.TK_LLNL, reflectivity="gold")
        elif name == "babyiaxo_xmm_gas":  # gas-stage specialisation
            _cache[name] = sa.initFullSetup(stage=L.SK_GAS)
        elif name == "babyiaxo_xmm_rot":  # rotated-telescope specialisation, effective-area flags, 100 mm chip
            full = sa.initFullSetup()
            full.setup.telescope_turned_x_deg, full.setup.telescope_turned_y_deg = 0.02, 0.1
            full.setup.chip_x_max = full.setup.chip_y_max = 100.0
            full.flags = L.CF_IGNORE_DET_WINDOW | L.CF_IGNORE_GAS_ABS | L.CF_IGNORE_CONV_PROB
            _cache[name] = full
        elif name == "babyiaxo_xmm_xray":  # X-ray test source: the generic instantiation, no exposure factor, weights of order 1
            _cache[name] = sa.initFullSetup(flags=L.CF_XRAY_TEST)
        elif name == "cast_abrixas_gas":  # generic instantiation with the gas stage read at run time
            _cache[name] = sa.initFullSetup(L.ES_CAST, L.DK_INGRID2017, L.SK_GAS, L.TK_ABRIXAS)
        else:
            raise KeyError(name)
    return _cache[name]


def run(name, mode, splits=(N,), env=None, seed=5, spectra=False):
    """One image of N rays in `mode`, traced as len(splits) accumulating launches, in a fresh context (the SART_* knobs are
    read when a context is created)."""
    env = env or {}
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        with sa.RayTracer(full_setup(name)) as rt:
            rt.set_accumulation_mode(mode)
            off = 0
            for k, n in enumerate(splits):
                if spectra:
                    img, s, spec = rt.trace_spectra(n, seed=seed, ray_id_offset=off, accumulate=(k > 0), n_radial_bins=2000, radial_max=10.0)
                else:
                    img, s = rt.trace_histogram(n, seed=seed, ray_id_offset=off, accumulate=(k > 0))
                    spec = None
                off += n
            quanta = rt.fixed_quanta() if mode == "fixed64" else None
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v
    assert off == N
    return img, s, spec, quanta


def assert_bitwise(a, b, what):
    img_a, s_a = a[0], a[1]
    img_b, s_b = b[0], b[1]
    assert np.array_equal(img_a.view(np.uint64), img_b.view(np.uint64)), what + ": image differs"
    for k in COUNTERS + SUMS:
        assert np.float64(s_a[k]).view(np.uint64) == np.float64(s_b[k]).view(np.uint64), (what, k, s_a[k], s_b[k])


@pytest.mark.parametrize("name", ["babyiaxo_xmm", "cast_llnl_gold", "babyiaxo_xmm_gas", "babyiaxo_xmm_rot", "babyiaxo_xmm_xray", "cast_abrixas_gas"])
def test_fixed64_is_bitwise_independent_of_placement_and_splitting(name):
    base = run(name, "fixed64")
    assert base[1]["N_RAYS"] == N and base[1]["N_PASSED"] > 1e5
    # replica count (8 <-> 64: another wave -> image mapping and another fold)
    assert_bitwise(base, run(name, "fixed64", env={"SART_IMAGE_REPLICAS": "64" if name != "cast_llnl_gold" else "8"}), "replicas")
    assert_bitwise(base, run(name, "fixed64", env={"SART_IMAGE_REPLICAS": "1"}), "one image")
    # LDS tile off: every hit goes to a global atomic
    assert_bitwise(base, run(name, "fixed64", env={"SART_NO_IMAGE_TILE": "1"}), "tile off")
    # two other splittings of the same ray ids over accumulating launches (not multiples of the 256-ray chunks)
    assert_bitwise(base, run(name, "fixed64", splits=(7_000_001, 12_999_999)), "two launches")
    assert_bitwise(base, run(name, "fixed64", splits=(1, 999, 9_999_000, 10_000_000)), "four launches")
    # stage A0 off: other ring traffic, another wave -> ray assignment, other per-workgroup partial sums
    assert_bitwise(base, run(name, "fixed64", env={"SART_NO_EARLY_REJECT": "1"}), "stage A0 off")
    # rays that provably miss the innermost shell's first mirror go through phase B instead of ending in phase A
    assert_bitwise(base, run(name, "fixed64", env={"SART_NO_SURE_MISS": "1"}), "innermost-shell shortcut off")


def _shell0_miss_radius(full, env=None):
    import ctypes as C
    env = env or {}
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        with sa.RayTracer(full) as rt:
            out = C.c_double()
            fn = rt.lib.sart_internal_shell0_miss_radius
            fn.argtypes, fn.restype = [C.c_void_p, C.POINTER(C.c_double)], C.c_int
            L.check(fn(rt.handle, C.byref(out)))
            return out.value
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v


def test_innermost_shell_shortcut_where_it_is_proved_and_only_there():
    """Phase A ends rays that select the innermost shell from so far inside that neither root of findPosParabolic (:677-682)
    can lie in the mirror: r3 of that shell minus the largest drift over the mirror's length.  On for Wolter telescopes with
    the solar source (tilted or not), off (-1) everywhere else; exactness: the parametrised test above (bitwise) and the
    record path, which does not take the shortcut, against the histogram."""
    full = full_setup("babyiaxo_xmm")
    rho = _shell0_miss_radius(full)
    s = full.setup
    beta = np.deg2rad(s.all_angles_deg[0])
    tl = np.tan(beta) * s.l_mirror
    r3 = -tl + np.sqrt(tl * tl + s.all_r1[0] ** 2)
    assert 148.0 < rho < r3 - 1.0 and r3 < s.all_r1[0]                 # 151.6 - 4.6e-3 x 300 mm - margin
    assert _shell0_miss_radius(full_setup("babyiaxo_xmm_gas")) == rho  # the stage does not enter
    assert _shell0_miss_radius(full, env={"SART_NO_SURE_MISS": "1"}) == -1.0
    assert _shell0_miss_radius(full, env={"SART_NO_EARLY_REJECT": "1"}) == -1.0   # the slope bound is built with the zones
    for name in ("cast_llnl_gold", "babyiaxo_xmm_xray"):                          # cones (nothing to gain); slopes unbounded
        assert _shell0_miss_radius(full_setup(name)) == -1.0, name
    rot = _shell0_miss_radius(full_setup("babyiaxo_xmm_rot"))                     # tilted by acos(cos 0.02 deg cos 0.1 deg): 0.53 mm more drift
    assert rho - 0.6 < rot < rho - 0.5
    assert 30.0 < _shell0_miss_radius(full_setup("cast_abrixas_gas")) < full_setup("cast_abrixas_gas").setup.all_r1[0]   # the other Wolter telescope
    # what it is worth and that it is exact where it counts most: the rays it ends are counted as "shell selected" and
    # nothing else; records (no shortcut) binned on the host give the histogram's counters
    with sa.RayTracer(full) as rt:
        n = 2_000_000
        rec = rt.traceAxionWrapper(n, seed=11)
        img, summ = rt.trace_histogram(n, seed=11)
    assert summ["N_PASSED"] == int(rec["passed"].sum()) and summ["N_HIT_NICKEL"] == int(rec["hitNickel"].sum())
    assert summ["N_PASSED_TILL_WINDOW"] == int(rec["passedTillWindow"].sum())
    assert summ["SUM_WEIGHTS"] == pytest.approx(float(rec["weights"][rec["passed"] != 0].sum()), rel=1e-12)


@pytest.mark.parametrize("name", ["babyiaxo_xmm", "cast_llnl_gold", "babyiaxo_xmm_rot", "babyiaxo_xmm_xray"])
def test_fixed64_agrees_with_f64_accumulation(name):
    fx = run(name, "fixed64")
    fl = run(name, "f64")
    for k in COUNTERS:
        assert fx[1][k] == fl[1][k], k
    peak = fl[0].max()
    assert np.abs(fx[0] - fl[0]).max() <= 1e-12 * peak, np.abs(fx[0] - fl[0]).max() / peak
    for k in ("SUM_WEIGHTS", "SUM_X", "SUM_Y", "SUM_R"):
        assert fx[1][k] == pytest.approx(fl[1][k], rel=1e-12), k
    assert fx[1]["SUM_WEIGHTS_SQ"] == pytest.approx(fl[1]["SUM_WEIGHTS_SQ"], rel=1e-6)   # coarse quantum by design (44-bit ray count): 2e-8 measured
    # integers all the way: the image sums to SUM_WEIGHTS exactly (no ray fell outside the image)
    q = fx[3]["weight"]
    pix = np.rint(fx[0] / q).astype(np.int64)
    assert np.array_equal(pix * q, fx[0])                       # every pixel is a whole number of quanta
    total = int(pix.astype(object).sum())                       # exact (Python integers); may exceed 2^53
    if fx[1]["N_OUTSIDE_IMAGE"] == 0:
        assert float(total) * q == fx[1]["SUM_WEIGHTS"]         # two limbs -> one correctly rounded double
    # the quanta are powers of two below the weight of any ray
    assert np.log2(q) == np.floor(np.log2(q)) and q < fl[1]["SUM_WEIGHTS"] / fl[1]["N_PASSED"] * 1e-8


def test_fixed64_spectra():
    fx = run("babyiaxo_xmm", "fixed64", spectra=True)
    fx2 = run("babyiaxo_xmm", "fixed64", spectra=True, splits=(3_333_333, 16_666_667), env={"SART_IMAGE_REPLICAS": "16"})
    fl = run("babyiaxo_xmm", "f64", spectra=True)
    assert_bitwise(fx, fx2, "spectra run")
    for k in ("radial_counts", "radial_weights", "energy_counts", "energy_weights", "energy_reflect"):
        assert np.array_equal(fx[2][k].view(np.uint64), fx2[2][k].view(np.uint64)), k
        scale = np.abs(fl[2][k]).max()
        assert np.abs(fx[2][k] - fl[2][k]).max() <= 1e-11 * scale, k
    assert fx[2]["radial_counts"].sum() == fx[1]["N_PASSED"] == fx[2]["energy_counts"].sum()


def test_fixed64_raw_accumulators_sum_like_integers():
    """The multi-GPU form on one card: two contexts trace the two halves of the ray ids into raw int64 accumulators, the
    buffers are added as integers (what an int64 reduce does) and finalized: bit for bit the single-context result."""
    import torch
    full = full_setup("babyiaxo_xmm")
    n_acc = sa.accumulator_len(256)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(stream):
        def traced(ranges):
            acc = torch.zeros(n_acc, dtype=torch.int64, device=dev)
            with sa.RayTracer(full) as rt:
                rt.set_stream(stream.cuda_stream)
                rt.set_accumulation_mode("fixed64")
                for lo, hi in ranges:
                    p = rt.trace_params(hi - lo, seed=9, ray_id_offset=lo, accumulate=True)
                    rt.trace_histogram_device(p, acc.data_ptr())
                rt.synchronize()
                return acc, rt.fixed_quanta()
        whole, q0 = traced([(0, N)])
        a, q1 = traced([(0, N // 2 + 12345)])
        b, q2 = traced([(N // 2 + 12345, N)])
        assert q0 == q1 == q2                                   # equal inputs -> equal quanta on every rank
        both = a + b                                            # limb-wise integer sum
        with sa.RayTracer(full) as rt:
            rt.set_stream(stream.cuda_stream)
            rt.set_accumulation_mode("fixed64")
            p = rt.trace_params(1000, seed=9, accumulate=False)
            scratch = torch.zeros(n_acc, dtype=torch.int64, device=dev)
            rt.trace_histogram_device(p, scratch.data_ptr())    # freezes this context's quanta
            assert rt.fixed_quanta() == q0
            out_w = torch.empty(n_acc, dtype=torch.float64, device=dev)
            out_s = torch.empty(n_acc, dtype=torch.float64, device=dev)
            rt.finalize_accumulator_device(p, whole.data_ptr(), out_w.data_ptr())
            rt.finalize_accumulator_device(p, both.data_ptr(), out_s.data_ptr())
            rt.synchronize()
        stream.synchronize()
    assert torch.equal(out_w.view(torch.int64), out_s.view(torch.int64))
    n_img = 256 * 256
    assert out_w[n_img + L.ACC["N_RAYS"]].item() == N
    for k, i in L.ACC_HI.items():                               # high limbs read 0 in the f64 layout
        assert out_w[n_img + i].item() == 0.0
    lo = whole[n_img + L.ACC["SUM_WEIGHTS"]].item()
    assert 0 <= lo < 2 ** L.FIXED_LIMB_BITS                      # normalised low limb


def test_fixed64_errors():
    full = full_setup("babyiaxo_xmm")
    with sa.RayTracer(full) as rt:
        assert rt.accumulation_mode() == L.ACCUM_F64
        with pytest.raises(L.SartError):
            rt.set_accumulation_mode(7)
        with pytest.raises(L.SartError):
            rt.set_accumulation_mode("fixed64", headroom_bits=60)
        rt.set_accumulation_mode("fixed64")
        with pytest.raises(L.SartError) as e:
            rt.fixed_quanta()
        assert e.value.code == -3                                # SART_ERR_NOT_READY
        rt.trace_histogram(100_000, seed=1)
        q = rt.fixed_quanta()["weight"]
        # the next accumulating launch drops the conversion probability (weights ~1e22 times larger): does not fit the quantum
        with pytest.raises(L.SartError) as e:
            rt.trace_histogram(100_000, seed=1, ray_id_offset=100_000, accumulate=True, flags=L.CF_IGNORE_CONV_PROB)
        assert e.value.code == -1 and "quantum" in str(e.value)
        # a launch that zeroes the accumulator re-freezes
        rt.trace_histogram(100_000, seed=1, flags=L.CF_IGNORE_CONV_PROB)
        assert rt.fixed_quanta()["weight"] > q * 1e15
        # back to f64: the scratch accumulator of the blocking call starts from zero again
        rt.set_accumulation_mode("f64")
        img, s = rt.trace_histogram(100_000, seed=1, accumulate=True)
        assert s["N_RAYS"] == 100_000 and img.sum() == pytest.approx(s["SUM_WEIGHTS"], rel=1e-12)
        # larger headroom = coarser quantum
        rt.set_accumulation_mode("fixed64", headroom_bits=40)
        rt.trace_histogram(100_000, seed=1)
        assert rt.fixed_quanta()["weight"] == q * 2.0 ** 13     # default headroom: 27 bits


def test_bench_fixed64_flux_is_identical_for_one_and_two_ranks():
    """End to end through bench.py: the same 6e7 rays as one rank and as two ranks (strong scaling, gloo rehearsal on this
    card, int64 reduce of the raw accumulators, finalize on rank 0) - the flux and the image sum come out bit for bit equal;
    in f64 mode they agree to 1e-12 only."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}

    def bench(gpus, accumulation):
        env = dict(base, SART_BENCH_BACKEND="gloo", SART_BENCH_DEVICE="0") if gpus > 1 else base
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(gpus), "--steps", "3", "--warmup", "1",
                              "--rays-per-step", "2e7", "--profile-run", "--scaling", "strong", "--accumulation", accumulation],
                             env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
        assert d["n_gpus"] == gpus and d["accumulation"] == accumulation and d["config"]["total_rays"] == 6e7
        return d["results"]

    one, two = bench(1, "fixed64"), bench(2, "fixed64")
    assert one["flux"] == two["flux"] and one["image_sum"] == two["image_sum"]          # JSON round-trips doubles exactly
    assert one["passed_fraction"] == two["passed_fraction"]
    f1 = bench(1, "f64")
    assert f1["flux"] == pytest.approx(one["flux"], rel=1e-12) and f1["passed_fraction"] == one["passed_fraction"]


def test_scan_fixed64_curve_is_identical_for_one_and_two_ranks(tmp_path):
    """tools/scan.py mass --shard rays (BASELINE configs[4]: rays of every mass point sharded over the ranks, one reduce of the
    accumulator per point) with --accumulation fixed64: the CSV written by one rank and by two ranks is the same file."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    outs = []
    for gpus in (1, 2):
        env = dict(base, SART_BENCH_BACKEND="gloo", SART_BENCH_DEVICE="0") if gpus > 1 else base
        out = str(tmp_path / ("scan%d.csv" % gpus))
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "scan.py"), "mass", "--gpus", str(gpus), "--shard", "rays",
                            "--accumulation", "fixed64", "--points", "3", "--rays", "3e6", "--massMin", "0.004", "--massMax", "0.012",
                            "--out", out], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(open(out).read())
    assert outs[0] == outs[1] and outs[0].count("\n") == 4
    flux = [float(l.split(",")[1]) for l in outs[0].splitlines()[1:]]
    assert min(flux) > 0 and flux[1] > 1.1 * max(flux[0], flux[2])   # the middle point sits on the resonance at m_gamma = 0.008235 eV


def test_fixed64_says_when_the_quantum_does_not_resolve_the_weights():
    """The quantum comes from a BOUND of the weights (table maxima).  A table with an outlier the rays never meet - here a
    reflectivity of 1e4 in an angle cell below every grazing angle of the optic - inflates the bound by 1e8: the accumulated
    weights then average ~250 quanta per ray, and the blocking call reports that instead of returning an image of rounding
    noise; with more fractional bits (smaller headroom) the same setup accumulates fine."""
    import copy
    base = full_setup("babyiaxo_xmm")
    full = copy.copy(base)
    refl = copy.copy(base.reflectivity)
    refl.data = base.reflectivity.data.copy()
    refl.data[0, 0, :] = 1e4
    full.reflectivity = refl
    with sa.RayTracer(full) as rt:
        img_f, s_f = rt.trace_histogram(2_000_000, seed=3)
        rt.set_accumulation_mode("fixed64")
        with pytest.raises(L.SartError) as e:
            rt.trace_histogram(2_000_000, seed=3)
        assert e.value.code == L.SART_ERR_ACCUMULATOR and "quanta" in str(e.value)   # (not an argument error: round 5)
        rt.set_accumulation_mode("fixed64", headroom_bits=16)
        img_x, s_x = rt.trace_histogram(2_000_000, seed=3)
    assert s_x["N_PASSED"] == s_f["N_PASSED"]
    # 47 fractional bits under a bound that is 1e8 too high: ~7e-7 of a real weight per ray, 6e-8 of the peak measured
    assert np.abs(img_x - img_f).max() <= 1e-6 * img_f.max() and s_x["SUM_WEIGHTS"] == pytest.approx(s_f["SUM_WEIGHTS"], rel=1e-8)


# ---- round 4: the checks live on the device path, the bound follows the axion mass, slots that wrap are reported -----------------

def _device_fixed_run(rt, torch, n, seed, image_n=256, flags=None, launches=1):
    """trace_histogram_device (raw int64) + finalize in place + synchronize, as bench.py / tools/scan.py drive the library."""
    acc = torch.zeros(sa.accumulator_len(image_n), dtype=torch.float64, device="cuda")
    for k in range(launches):
        p = rt.trace_params(n, seed=seed, ray_id_offset=k * n, image_n=image_n, flags=flags, accumulate=True)
        rt.trace_histogram_device(p, acc.data_ptr())
    rt.finalize_accumulator_device(p, acc.data_ptr())
    rt.synchronize()
    return acc.cpu().numpy()


def test_fixed64_device_path_resolves_a_mass_far_off_resonance():
    """off resonance the gas-stage conversion probability falls like 4 / (q L)^2 - at m_a = 50 m_gamma it is ~1e-7
    of the coherent maximum the quantum used to be derived from, at the edge of what the integers resolve, and the device path
    (trace_histogram_device + finalize, what the multi-GPU drivers use) had no check.  Now the weight bound follows the mass
    (re-frozen when the mass changes): the far point resolves as well as the resonance does."""
    import torch
    full = full_setup("babyiaxo_xmm_gas")
    m_gamma = 0.008235
    n = 5_000_000
    k = 256 * 256
    with sa.RayTracer(full) as rt:
        out = {}
        for m in (m_gamma, 50.0 * m_gamma):
            rt.set_axion_mass(m)
            rt.set_accumulation_mode("f64")
            img_f, s_f = rt.trace_histogram(n, seed=13)
            rt.set_accumulation_mode("fixed64")
            host = _device_fixed_run(rt, torch, n, seed=13)
            q = rt.fixed_quanta()["weight"]
            s_x = {key: host[k + i] for key, i in L.ACC.items()}
            assert s_x["N_PASSED"] == s_f["N_PASSED"] and s_x["N_RAYS"] == n
            assert s_x["SUM_WEIGHTS"] == pytest.approx(s_f["SUM_WEIGHTS"], rel=1e-9)
            assert s_x["SUM_WEIGHTS_SQ"] == pytest.approx(s_f["SUM_WEIGHTS_SQ"], rel=1e-9)
            assert np.abs(host[:k].reshape(256, 256) - img_f).max() <= 1e-9 * img_f.max()
            out[m] = (s_x["SUM_WEIGHTS"], q)
    (w_res, q_res), (w_far, q_far) = out[m_gamma], out[50.0 * m_gamma]
    assert w_far < 1e-4 * w_res and q_far < 1e-3 * q_res       # the quantum went down with the weights


def test_fixed64_unresolved_weights_fail_on_the_device_path_too():
    """The outlier-in-a-table case of test_fixed64_says_when_the_quantum_does_not_resolve_the_weights, through
    trace_histogram_device + finalize: the finalize kernel sets the status word, the next synchronize raises (once)."""
    import copy
    import torch
    base = full_setup("babyiaxo_xmm")
    full = copy.copy(base)
    refl = copy.copy(base.reflectivity)
    refl.data = base.reflectivity.data.copy()
    refl.data[0, 0, :] = 1e4
    full.reflectivity = refl
    with sa.RayTracer(full) as rt:
        rt.set_accumulation_mode("fixed64")
        with pytest.raises(L.SartError) as e:
            _device_fixed_run(rt, torch, 2_000_000, seed=3)
        assert e.value.code == L.SART_ERR_ACCUMULATOR and "quanta" in str(e.value)   # (not an argument error: round 5)
        rt.synchronize()                                         # reported once
        rt.set_accumulation_mode("fixed64", headroom_bits=16)
        host = _device_fixed_run(rt, torch, 2_000_000, seed=3)   # resolves with 47 fractional bits ...
        assert host[256 * 256 + L.ACC["SUM_WEIGHTS"]] > 0
        assert np.isnan(host[256 * 256 + L.ACC["SUM_WEIGHTS_SQ"]])   # ... but not the squares (1e-16 of their bound): NaN, not a made-up number


def test_fixed64_reports_a_slot_that_wrapped():
    """pixels, spectra bins and sums are int64 slots; at headroom_bits = 16 a pixel holds 2^16 bound-weight
    rays.  A one-pixel image takes every passed ray and wraps - several times over - within 2e8 rays.  The finalize kernel
    checks every slot's sign / size AND that the pixels add up to SUM_WEIGHTS exactly (each wrap takes 2^64 out of that sum, so
    the check does not depend on where the wrapped value lands): the next synchronising call fails instead of handing out a
    wrapped number."""
    import torch
    full = full_setup("babyiaxo_xmm")
    n = 40_000_000
    with sa.RayTracer(full) as rt:
        rt.set_accumulation_mode("fixed64", headroom_bits=16)
        host = _device_fixed_run(rt, torch, 50_000, seed=2, image_n=1)           # ~1e4 passed rays: fits
        assert host[0] == pytest.approx(host[1 + L.ACC["SUM_WEIGHTS"]], rel=1e-12) and host[0] > 0
        with pytest.raises(L.SartError) as e:
            _device_fixed_run(rt, torch, n, seed=2, image_n=1, launches=5)
        assert e.value.code == L.SART_ERR_ACCUMULATOR and "wrapped" in str(e.value)
        with pytest.raises(L.SartError) as e:                     # the blocking call reports it, too
            for k in range(5):
                rt.trace_histogram(n, seed=2, ray_id_offset=k * n, image_n=1, accumulate=(k > 0))
        assert "wrapped" in str(e.value)
        # spectra bins: one radial bin for everything wraps the same way, the image (256 x 256) does not
        with pytest.raises(L.SartError) as e:
            for k in range(5):
                rt.trace_spectra(n, seed=2, ray_id_offset=k * n, n_radial_bins=1, radial_max=10.0, accumulate=(k > 0))
        assert "wrapped" in str(e.value)
        rt.set_accumulation_mode("fixed64", headroom_bits=40)
        host = _device_fixed_run(rt, torch, n, seed=2, image_n=1, launches=5)
        assert host[0] == pytest.approx(host[1 + L.ACC["SUM_WEIGHTS"]], rel=1e-9)


def test_setting_the_same_accumulation_mode_again_keeps_the_frozen_quanta():
    """sart_set_accumulation_mode used to clear the frozen quanta even when nothing changed; an accumulate = 1
    launch behind it could then re-freeze other quanta into an accumulator that already held data."""
    full = full_setup("babyiaxo_xmm")
    with sa.RayTracer(full) as rt:
        rt.set_accumulation_mode("fixed64")
        _, s1 = rt.trace_histogram(1_000_000, seed=4)
        q = rt.fixed_quanta()
        rt.set_accumulation_mode("fixed64")                      # no change
        assert rt.fixed_quanta() == q
        # an accumulating launch whose weights would ask for another quantum still meets the frozen one: refused, not re-frozen
        with pytest.raises(L.SartError):
            rt.trace_histogram(1_000_000, seed=4, ray_id_offset=1_000_000, accumulate=True, flags=L.CF_IGNORE_CONV_PROB)
        _, s2 = rt.trace_histogram(1_000_000, seed=4, ray_id_offset=1_000_000, accumulate=True)
        assert s2["N_RAYS"] == 2_000_000 and rt.fixed_quanta() == q


@pytest.mark.parametrize("name", ["babyiaxo_xmm", "cast_llnl_gold", "babyiaxo_xmm_gas", "babyiaxo_xmm_xray"])
def test_fixed64_sum_of_squared_weights_has_a_high_limb(name):
    """SUM_WEIGHTS_SQ was one int64 with a quantum of 2^-19 of the squared bound.  It now has a high limb and a
    quantum of 2^-39: it agrees with the f64 mode to 1e-9 on every workload."""
    f = run(name, "f64")
    x = run(name, "fixed64")
    assert x[1]["SUM_WEIGHTS_SQ"] == pytest.approx(f[1]["SUM_WEIGHTS_SQ"], rel=1e-9)
    assert x[3]["weight_sq"] == x[3]["weight"] ** 2 * 2.0 ** (2 * (63 - 27) - 39)     # 2^(2e - 39) beside 2^(e - 63 + 27)


def test_fixed64_rollover_limbs_carry_an_accumulation_past_the_headroom():
    """the only remedy for an accumulation that outlives its headroom used to be "pick 31 bits yourself".  With
    sart_rollover_accumulator_device every slot has a second limb; folded between launches, the accumulation runs on at the fine
    quantum.  Scaled down as test_fixed64_reports_a_slot_that_wrapped does it: at headroom_bits = 16 a pixel holds 2^16
    bound-weight rays - what 2^27 hold in 3e12 BabyIAXO rays the brightest pixel of this image meets in ~1.5e9.
      * without the roll-over the run ends in SART_ERR_ACCUMULATOR (wrapped);
      * with it: no error, and every slot's (hi 2^40 + lo) is EXACTLY the sum of the integers of the fifteen launches traced one
        by one into fresh accumulators - bit for bit;
      * finalized, it equals the same rays accumulated with 40 bits of headroom (a 2^24 times coarser quantum) to 1e-9."""
    import torch
    full = full_setup("babyiaxo_xmm")
    n, launches, seed = 100_000_000, 15, 6
    with sa.RayTracer(full) as rt:
        rt.set_accumulation_mode("fixed64", headroom_bits=16)
        with pytest.raises(L.SartError) as e:
            _device_fixed_run(rt, torch, n, seed=seed, launches=launches)
        assert e.value.code == L.SART_ERR_ACCUMULATOR and "wrapped" in str(e.value)
        rt.synchronize()
        # the same launches, a fold behind every one
        acc = torch.zeros(sa.accumulator_len(256), dtype=torch.int64, device="cuda")
        hi = torch.zeros_like(acc)
        exact = np.zeros(acc.numel(), dtype=object)
        one = torch.zeros_like(acc)
        for k in range(launches):
            p = rt.trace_params(n, seed=seed, ray_id_offset=k * n, accumulate=True)
            rt.trace_histogram_device(p, acc.data_ptr())
            rt.rollover_accumulator_device(p, acc.data_ptr(), hi.data_ptr())
            q = rt.trace_params(n, seed=seed, ray_id_offset=k * n, accumulate=(k > 0))   # (k = 0 zeroes `one` and keeps the quanta)
            one.zero_()
            q.accumulate = 1
            rt.trace_histogram_device(q, one.data_ptr())
            rt.synchronize()
            exact += one.cpu().numpy().astype(object)
        lo_h, hi_h = acc.cpu().numpy(), hi.cpu().numpy()
        assert lo_h.min() >= 0 and lo_h.max() < 2 ** 40 and hi_h.min() >= 0 and hi_h[:256 * 256].max() > 2 ** 22   # (> 2^62 in one limb)
        got = hi_h.astype(object) * 2 ** 40 + lo_h.astype(object)
        # the five two-limb sums of the accumulator keep their own high slot: compare slot pairs as one number
        k0 = 256 * 256
        pairs = dict(L.ACC_HI)
        for name, kh in list(pairs.items()) + [("SUM_WEIGHTS_OUTSIDE", 18)]:
            kl = 17 if name == "SUM_WEIGHTS_OUTSIDE" else L.ACC[name]
            a = got[k0 + kh] * 2 ** 40 + got[k0 + kl]
            b = exact[k0 + kh] * 2 ** 40 + exact[k0 + kl]
            assert a == b, name
            got[k0 + kh] = got[k0 + kl] = exact[k0 + kh] = exact[k0 + kl] = 0
        assert (got == exact).all()
        out = torch.zeros(acc.numel(), dtype=torch.float64, device="cuda")
        rt.finalize_accumulator_limbs_device(p, acc.data_ptr(), hi.data_ptr(), out.data_ptr())
        rt.synchronize()                                   # no error: nothing wrapped, the conservation sums include the limbs
        fine = out.cpu().numpy()
        rt.set_accumulation_mode("fixed64", headroom_bits=40)
        coarse = _device_fixed_run(rt, torch, n, seed=seed, launches=launches)
        # the fold is an integer operation: a context whose accumulators hold doubles refuses it
        rt.set_accumulation_mode("f64")
        with pytest.raises(L.SartError) as e:
            rt.rollover_accumulator_device(p, acc.data_ptr(), hi.data_ptr())
        assert e.value.code == L.SART_ERR_INVALID_ARGUMENT and "F64" in str(e.value)
    assert fine[k0 + L.ACC["N_RAYS"]] == coarse[k0 + L.ACC["N_RAYS"]] == n * launches
    for name in ("SUM_WEIGHTS", "SUM_X", "SUM_Y", "SUM_R", "N_PASSED", "SUM_WEIGHTS_SQ"):
        assert fine[k0 + L.ACC[name]] == pytest.approx(coarse[k0 + L.ACC[name]], rel=1e-9), name
    lit = coarse[:k0] > 1e-3 * coarse[:k0].max()
    assert np.abs(fine[:k0][lit] / coarse[:k0][lit] - 1.0).max() < 1e-6      # (the coarse run's own rounding: 2^-23 of the bound per ray)
    assert fine[:k0].sum() == pytest.approx(fine[k0 + L.ACC["SUM_WEIGHTS"]], rel=1e-12)


def test_a_handful_of_faint_rays_is_not_an_unresolved_accumulator():
    """Found by the round-6 stream: a launch of ONE ray into a fresh FIXED64 accumulator (the first piece of the four-launch split
    above) whose ray happens to pass with a weight of 1e-8 of the bound averaged below 2^12 quanta per passed ray, and the blocking
    call failed with SART_ERR_ACCUMULATOR.  The mean is judged once 256 rays have passed; below that every ray is still exact to
    half a quantum.  Every single-ray launch of the first 300 CAST ray ids runs, and equals the f64 launch to half a quantum."""
    full = full_setup("cast_llnl_gold")
    with sa.RayTracer(full) as rt:
        rt.set_accumulation_mode("f64")
        ref = [rt.trace_flux(1, seed=5, ray_id_offset=i) for i in range(300)]
        rt.set_accumulation_mode("fixed64")
        got = [rt.trace_flux(1, seed=5, ray_id_offset=i) for i in range(300)]
        q = rt.fixed_quanta()["weight"]
    passed = [i for i in range(300) if ref[i]["N_PASSED"] == 1]
    assert len(passed) > 200
    for i in range(300):
        assert got[i]["N_PASSED"] == ref[i]["N_PASSED"]
        assert abs(got[i]["SUM_WEIGHTS"] - ref[i]["SUM_WEIGHTS"]) <= 0.5 * q * (1 + 1e-9)
    faint = [i for i in passed if ref[i]["SUM_WEIGHTS"] < 4096 * q]
    assert faint, "no single ray below 2^12 quanta among these ids: pick other ids"
