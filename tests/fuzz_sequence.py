#!/usr/bin/env python3
"""A randomised campaign over SEQUENCES of C-ABI calls (GPU box): one long-lived context is taken through a random walk of state
changes - sart_set_setup with perturbed geometry / stage / rotation / chip size, sart_set_axion_mass, sart_set_telescope_angles,
accumulation mode, image binning - with a trace through a random door after each; every result must be what a FRESH context,
configured directly to the walk's current state, gives for the same call (counters exactly; sums to 1e-12 in f64, bit for bit in
fixed64).  Aimed at what the per-feature tests cannot see: stale caches (LDS tile position, hoisted tables, stage-A0 zones,
kernel variant, fixed-point quanta) surviving a state change.

  python tests/fuzz_sequence.py [--walks 40] [--steps 12] [--out gpurun_out/fuzz_sequence.txt]"""
import argparse
import ctypes as C
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

COUNTERS = ("N_RAYS", "N_REACHED_TELESCOPE", "N_SHELL_SELECTED", "N_HIT_NICKEL", "N_PASSED_TILL_WINDOW", "N_PASSED", "N_OUTSIDE_IMAGE")
SUMS = ("SUM_WEIGHTS", "SUM_WEIGHTS_SQ", "SUM_X", "SUM_Y", "SUM_R")


def mutate(rng, s, L, base):
    """One state change on the setup struct `s` (in place); returns a label."""
    kind = int(rng.integers(0, 9))
    if kind == 0:
        s.telescope_turned_y_deg = float(rng.choice([0.0, 0.0, rng.uniform(-0.1, 0.1)]))
        return "turn y %.4f" % s.telescope_turned_y_deg
    if kind == 1:
        s.telescope_turned_x_deg = float(rng.choice([0.0, 0.0, rng.uniform(-0.05, 0.05)]))
        return "turn x %.4f" % s.telescope_turned_x_deg
    if kind == 2:
        s.stage = L.SK_GAS if s.stage == L.SK_VACUUM else L.SK_VACUUM
        return "stage %d" % s.stage
    if kind == 3:
        s.chip_x_max = s.chip_y_max = float(rng.choice([14.0, 14.0, rng.uniform(8.0, 100.0)]))
        return "chip %.1f" % s.chip_x_max
    if kind == 4:
        s.magnet_radiusCB = base.magnet_radiusCB * rng.uniform(0.6, 1.2)
        return "bore %.1f" % s.magnet_radiusCB
    if kind == 5:
        s.pipe_cb_vt3_radius = base.pipe_cb_vt3_radius * rng.uniform(0.7, 1.3)
        s.pipe_vt3_xrt_radius = base.pipe_vt3_xrt_radius * rng.uniform(0.7, 1.3)
        return "pipes"
    if kind == 6:
        s.lateral_shift, s.transversal_shift = rng.uniform(-2.0, 2.0), rng.uniform(-2.0, 2.0)
        return "shift"
    if kind == 7:
        s.m_axion = float(rng.uniform(0.0, 0.03))
        return "m_a (setup) %.5f" % s.m_axion
    s.distance_detector_xrt = base.distance_detector_xrt * rng.uniform(0.98, 1.02)
    return "focal"


def trace(rt, rng_state, L, s):
    """One trace through a door chosen by rng_state (a tuple, so that both contexts make the same call)."""
    door, n, seed, off, img_n, flags = rng_state
    if door == 0:
        return ("hist", rt.trace_histogram(n, seed=seed, ray_id_offset=off, image_n=img_n, flags=flags))
    if door == 1:
        return ("flux", (None, rt.trace_flux(n, seed=seed, ray_id_offset=off, flags=flags)))
    if door == 2:
        img, summ, spec = rt.trace_spectra(n, seed=seed, ray_id_offset=off, flags=flags, image_n=img_n, n_radial_bins=300)
        return ("spectra", (img, summ))
    if door == 3:
        rec, cnt = rt.traceAxionWrapperPassed(n, seed=seed, ray_id_offset=off, flags=flags)
        return ("passed", (rec.tobytes(), cnt))
    if door == 4:
        if s.stage == L.SK_GAS:
            per, shared = rt.trace_mass_scan([0.3 * s.m_axion + 1e-4, s.m_axion, 0.02], n, seed=seed, ray_id_offset=off, flags=flags)
        else:
            per, shared = rt.trace_angular_scan([s.telescope_turned_y_deg, 0.03, -0.02], n, seed=seed, ray_id_offset=off, flags=flags)
        return ("scan", ({k: np.asarray(v).tolist() for k, v in per.items()}, shared))
    a = rt.trace_histogram(n // 2, seed=seed, ray_id_offset=off, image_n=img_n, flags=flags)
    return ("split", rt.trace_histogram(n - n // 2, seed=seed, ray_id_offset=off + n // 2, image_n=img_n, flags=flags, accumulate=True))


def same(kind, a, b, fixed):
    if kind == "passed":
        assert a[1] == b[1], (a[1], b[1])
        assert a[0] == b[0], "records differ"
        return
    if kind == "scan":
        for k in a[0]:
            x, y = np.asarray(a[0][k]), np.asarray(b[0][k])
            if k.startswith("N_") or fixed:
                assert np.array_equal(x, y), (k, x, y)
            else:
                np.testing.assert_allclose(x, y, rtol=1e-12, err_msg=k)
        assert a[1] == b[1] or all(a[1][k] == b[1][k] for k in a[1] if k.startswith("N_")), (a[1], b[1])
        return
    (img_a, sa_), (img_b, sb_) = a, b
    for k in COUNTERS:
        assert sa_[k] == sb_[k], (k, sa_[k], sb_[k])
    for k in SUMS:
        if fixed:
            assert np.float64(sa_[k]).view(np.uint64) == np.float64(sb_[k]).view(np.uint64), (k, sa_[k], sb_[k])
        else:
            assert sa_[k] == sb_[k] or abs(sa_[k] - sb_[k]) <= 1e-12 * abs(sb_[k]), (k, sa_[k], sb_[k])
    if img_a is not None:
        if fixed:
            assert np.array_equal(img_a, img_b), "images differ"
        else:
            np.testing.assert_allclose(img_a, img_b, rtol=1e-10, atol=1e-13 * max(float(img_b.max()), 1e-300))


def run_walk(walk, steps, sa, L):
    from tests.conftest import make_setup
    rng = np.random.default_rng(70_000 + walk)
    name = ["babyiaxo_xmm", "cast_llnl", "cast_abrixas", "babyiaxo_xmm_gas"][walk % 4]
    full = make_setup(name)
    base = full.setup.copy()
    s = full.setup          # (the very object: trace_params takes the image range from the chip size in it)
    mode = "f64"
    log = []
    with sa.RayTracer(full) as rt:
        for step in range(steps):
            what = int(rng.integers(0, 6))
            if what <= 2:
                log.append(mutate(rng, s, L, base))
                L.check(rt.lib.sart_set_setup(rt.handle, C.byref(s)))
            elif what == 3:
                s.m_axion = float(rng.uniform(0.0, 0.03))
                rt.set_axion_mass(s.m_axion)
                log.append("set_axion_mass %.5f" % s.m_axion)
            elif what == 4:
                s.telescope_turned_y_deg = float(rng.uniform(-0.1, 0.1))
                rt.set_telescope_angles(float("nan"), s.telescope_turned_y_deg)
                log.append("set_telescope_angles %.4f" % s.telescope_turned_y_deg)
            else:
                mode = "fixed64" if mode == "f64" else "f64"
                rt.set_accumulation_mode(mode)
                log.append("mode " + mode)
            flags = int(rng.choice([0, 0, L.CF_IGNORE_DET_WINDOW | L.CF_IGNORE_GAS_ABS | L.CF_IGNORE_CONV_PROB, L.CF_IGNORE_REFLECTION]))
            call = (int(rng.integers(0, 6)), int(rng.integers(20_000, 400_000)), int(rng.integers(1, 1 << 30)), int(rng.integers(0, 1 << 34)),
                    int(rng.choice([256, 256, 64, 31])), flags)
            kind, got = trace(rt, call, L, s)
            log.append("%s n=%d" % (kind, call[1]))
            fresh = make_setup(name)
            fresh.setup = s.copy()
            with sa.RayTracer(fresh) as rt2:
                rt2.set_accumulation_mode(mode)
                _, want = trace(rt2, call, L, s)
            try:
                same(kind, got, want, mode == "fixed64")
            except AssertionError as e:
                raise AssertionError("step %d of [%s]: %s" % (step, "; ".join(log), str(e)[:400])) from None
    return name, log


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--walks", type=int, default=40)
    ap.add_argument("--first", type=int, default=0)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--out", default="gpurun_out/fuzz_sequence.txt")
    args = ap.parse_args()
    import solaraxionraytracing_amd as sa
    from solaraxionraytracing_amd import _lib as L
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    failed, t0 = [], time.time()
    with open(args.out, "w") as out:
        def emit(line):
            print(line, flush=True)
            out.write(line + "\n")
            out.flush()
        emit("# tests/fuzz_sequence.py: walks %d .. %d, %d steps each, build %s" % (args.first, args.first + args.walks - 1, args.steps, L.build_id()))
        for walk in range(args.first, args.first + args.walks):
            try:
                name, log = run_walk(walk, args.steps, sa, L)
                emit("walk %4d ok    %-18s %s" % (walk, name, "; ".join(log)))
            except Exception as e:   # noqa: BLE001 - the campaign goes on
                failed.append(walk)
                emit("walk %4d FAIL  %s" % (walk, "".join(traceback.format_exception_only(type(e), e)).strip()[:1500]))
        emit("# %d walks, %d failed %s, %.0f s" % (args.walks, len(failed), failed, time.time() - t0))
    return 1 if failed else 0


if __name__ == "__main__":
    raise SystemExit(main())
