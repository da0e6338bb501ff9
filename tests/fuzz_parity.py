#!/usr/bin/env python3
"""A randomised parity campaign (GPU box): setups nobody wrote by hand - magnet, pipes, telescope attitude and position, detector
side, gas stage, hole types, X-ray source variants, every flag combination, all drawn together - through every door of the C-ABI
against the binary128 build of the oracle (tests/test_gpu_parity.py's randomised tests, many more of them and all perturbations
at once).  A case fails on the first difference; the campaign goes on and reports.

  python tests/fuzz_parity.py [--cases 200] [--first 0] [--rays 30000] [--out gpurun_out/fuzz_parity.txt]"""
import argparse
import ctypes as C
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def build_case(case, sa, L, small):
    rng = np.random.default_rng(50_000 + case)
    tel, exp = [(L.TK_XMM, L.ES_BABYIAXO), (L.TK_LLNL, L.ES_CAST), (L.TK_ABRIXAS, L.ES_CAST), (L.TK_XMM, L.ES_BABYIAXO)][case % 4]
    det = [L.DK_INGRID2017, L.DK_INGRID2018, L.DK_INGRIDIAXO][int(rng.integers(0, 3))]
    gas = rng.random() < 0.35
    xray = rng.random() < 0.2
    flags = 0
    for bit in (L.CF_IGNORE_DET_WINDOW, L.CF_IGNORE_GAS_ABS, L.CF_IGNORE_CONV_PROB, L.CF_IGNORE_REFLECTION):
        if rng.random() < 0.25:
            flags |= bit
    kw = dict(small)
    if xray:
        flags |= L.CF_XRAY_TEST
        src = L.TestSourceConfig()
        src.active, src.activity = 1, 0.125
        src.parallel = int(rng.random() < 0.6)
        src.energy = float(rng.uniform(0.5, 8.0))
        if exp == L.ES_CAST:
            src.distance, src.radius, src.lengthCol = float(rng.uniform(50.0, 500.0)), float(rng.uniform(5.0, 25.0)), float(rng.uniform(0.0, 40.0))
        else:
            src.distance, src.radius, src.lengthCol = float(rng.uniform(1000.0, 1e6)), float(rng.uniform(100.0, 450.0)), 0.0
        if not src.parallel:
            src.distance = float(rng.uniform(2e5, 2e6))
        kw["source_cfg"] = src
    full = sa.initFullSetup(exp, det, L.SK_GAS if gas else L.SK_VACUUM, tel, flags=flags, **kw)
    full.flags = flags
    s = full.setup
    s.magnet_radiusCB *= rng.uniform(0.6, 1.3)
    s.magnet_lengthB *= rng.uniform(0.8, 1.1)
    s.magnet_lengthColdbore = s.magnet_lengthB * rng.uniform(1.01, 1.08)
    s.pipe_cb_vt3_radius *= rng.uniform(0.7, 1.4)
    s.pipe_vt3_xrt_radius *= rng.uniform(0.7, 1.4)
    s.pipe_cb_vt3_length *= rng.uniform(0.5, 2.0)
    s.pipe_vt3_xrt_length *= rng.uniform(0.5, 2.0)
    if rng.random() < 0.5:
        s.telescope_turned_x_deg = rng.uniform(-0.08, 0.08)
        s.telescope_turned_y_deg = rng.uniform(-0.08, 0.08)
    if rng.random() < 0.4:
        s.optics_entrance[0] += rng.uniform(-5.0, 5.0)
        s.optics_entrance[1] += rng.uniform(-5.0, 5.0)
    s.lateral_shift = rng.uniform(-1.0, 1.0)
    s.transversal_shift = rng.uniform(-1.0, 1.0)
    s.distance_detector_xrt *= rng.uniform(0.97, 1.03)
    s.chip_x_max = s.chip_y_max = rng.uniform(10.0, 60.0)
    if tel == L.TK_XMM and rng.random() < 0.3:
        s.hole_type = [L.HT_CIRCLE, L.HT_CROSS, L.HT_STAR, L.HT_SQUARE, L.HT_DIAMOND][int(rng.integers(0, 5))]
        s.number_of_holes = int(rng.integers(1, 6))
        s.hole_in_optics = rng.uniform(2.0, 25.0)
    if rng.random() < 0.5:   # detector side
        s.radius_window = rng.uniform(4.0, 9.0)
        s.number_of_strips = int(rng.integers(2, 9))
        s.open_aperture_ratio = rng.uniform(0.7, 0.95)
        w, d = C.c_double(), C.c_double()
        assert L.load_host().sart_host_calc_window_vals(s.radius_window, s.number_of_strips, s.open_aperture_ratio, C.byref(w), C.byref(d)) == 0
        s.strip_width_window, s.strip_dist_window = w.value, d.value
        s.distance_window_focal_plane = rng.uniform(-5.0, 5.0)
        s.depth_det = rng.uniform(10.0, 40.0)
    if tel == L.TK_LLNL and rng.random() < 0.5:
        s.pipes_turned_deg = rng.uniform(2.0, 3.5)
    if gas:
        s.magnet_pGasRoom = rng.uniform(0.2, 3.0)
        s.magnet_tGas = rng.uniform(1.7, 293.0)
        s.magnet_B = rng.uniform(1.0, 9.0)
        s.m_axion = rng.uniform(0.0, 0.03)
        s.g_agamma = 10.0 ** rng.uniform(-13.0, -10.0)
    label = "%s/%s%s%s flags %#x" % ({L.TK_XMM: "xmm", L.TK_LLNL: "llnl", L.TK_ABRIXAS: "abrixas"}[tel], "gas" if gas else "vac",
                                     " xray" + ("-par" if xray and kw["source_cfg"].parallel else "-div" if xray else "") if xray else "",
                                     " rot" if s.telescope_turned_y_deg or s.telescope_turned_x_deg else "", flags)
    return full, flags, gas, xray, label, int(rng.integers(0, 1 << 40)), int(rng.integers(1, 1 << 31))


def run_case(case, n, sa, L, Oracle, compare_records, full_size=False):
    from tests.conftest import SMALL
    # full_size: the tables of BASELINE's configurations (1968 x 1500 emission CDFs, 1000 x 1000 reflectivity grid) instead of the
    # shrunken ones - the guide tables, the shell look-up table and the hoisted per-energy tables at the sizes the bench runs on
    full, flags, gas, xray, label, off, seed = build_case(case, sa, L, {} if full_size else SMALL)
    ref = Oracle(full, "q").trace_records(n, seed=seed, ray_id_offset=off, flags=flags)
    g = {"rec_" + k: ref[k] for k in ref.dtype.names}
    want = {k: int((ref[f] != 0).sum()) for k, f in (("N_PASSED", "passed"), ("N_PASSED_TILL_WINDOW", "passedTillWindow"), ("N_HIT_NICKEL", "hitNickel"))}
    with sa.RayTracer(full) as rt:
        rec = rt.traceAxionWrapper(n, seed=seed, ray_id_offset=off, flags=flags)
        for f in ("passed", "passedTillWindow", "hitNickel"):
            np.testing.assert_array_equal(rec[f], ref[f], err_msg=f)
        both = ref["passed"] != 0
        if want["N_PASSED"] > 0.15 * n or want["N_PASSED"] == 0:
            compare_records(rec, g, 1e-9, 1.0 if gas else 2e-8, 0.0)      # (gas stage: the weights are compared below)
        else:   # few rays pass (compare_records wants a populated sample): the fields of those that do
            for f in ("pointdataX", "pointdataY"):
                assert np.abs(rec[f][both] - ref[f][both]).max(initial=0.0) < 1e-9, f
            np.testing.assert_array_equal(rec["energiesPre"], ref["energiesPre"])
            if not gas:
                np.testing.assert_allclose(rec["weights"][both], ref["weights"][both], rtol=2e-8)
        if gas and both.any():
            # Gas stage: the probability carries 1 + e^(-GL) - 2 e^(-GL/2) cos(qL) and the per-ray gas column (axionMassforMagnet.nim:
            # 75-113, raytracer.nim:1603-1614); on a few rays in 1e4 - entries through the bore wall with a short way left in the
            # field, nulls of the oscillating factor - the reference's own f64 formulation is ill-conditioned (its f64 build is off
            # from its binary128 build by 1e-6 ... 1e-1 there).  Rule: within 2e-8 of the binary128 result, or - on at most 1e-3 of
            # the rays - within 2e-7 or no further from it than EIGHT times what the f64 build of the same formulas is on that very ray.
            # (Errors on such rays are two draws from one wide distribution - the ray's condition number times a few ulp, with other
            # roundings on either side: 2.02e-8 against the f64 build's 7e-9 happens, case 1202; for two such draws P(|a| > 2 |b|) is
            # 0.16 ... 0.3 per ray, and the 2000 cases of the round-6 stream met two rays at 3.8 x and 3.0 x - cases 494 and 1673, the
            # f64 build itself off by 9e-7 and 1.8e-4 there - where round 5's rays had met none above 2 x.  A wrong formula shows on
            # every ray, not on one in 1e4 where the reference's own arithmetic has lost its digits.)
            f64 = Oracle(full, "f64").trace_records(n, seed=seed, ray_id_offset=off, flags=flags)
            ok = both & (f64["passed"] != 0)
            for f in ("weights", "transmissionMagnet"):
                err = np.abs(rec[f][ok] / ref[f][ok] - 1.0)
                env = np.abs(f64[f][ok] / ref[f][ok] - 1.0)
                over = err > 2e-8
                # ... on at most 1e-3 of the rays, or a tenth as many as the f64 build of the reference's own formulas loses: near the
                # resonance (m_a within a few per cent of m_gamma: q L small, and Gamma L small with it) the bracket 1 + e^(-GL) -
                # 2 e^(-GL/2) cos(qL) cancels to (GL/2)^2 + (qL)^2 on EVERY ray, and that build is beyond 2e-8 on 30 - 80 % of all
                # rays (median error 3e-8 ... 9e-8: cases 2762, 2846, 3545 of the extended campaign, and 494, 1673, 1202 as well),
                # where the HIP path - the same formula with fused multiply-adds and its own exp / cos - is beyond it on 1.2 ... 1.5e-3
                assert over.sum() <= max(3, 1e-3 * ok.sum(), 0.1 * (env > 2e-8).sum()), (f, "rays beyond 2e-8", int(over.sum()), int(ok.sum()),
                                                                                        "f64 build beyond 2e-8", int((env > 2e-8).sum()))
                bad = err[over] > np.maximum(8.0 * env[over], 2e-7)
                if bad.any() and (env > 2e-8).mean() > 0.05:
                    # the whole setup is ill-conditioned (near the resonance: see above), tens of rays are "over", and the ray-by-ray
                    # envelope compares two draws per ray: 30 rays x P(ratio > 8) ~ 0.08 expects a few beyond it (case 2762: two).
                    # Held instead to the f64 build's error DISTRIBUTION: the k-th worst ray of the HIP path no worse than twice
                    # the k-th worst ray of that build (HIP 3.4e-5 against 8.5e-5 at k = 1 there)
                    e_s, v_s = np.sort(err)[::-1][:int(over.sum())], np.sort(env)[::-1][:int(over.sum())]
                    assert np.all(e_s <= np.maximum(2.0 * v_s, 2e-7)), (f, "order statistics", e_s[:10].tolist(), v_s[:10].tolist())
                else:
                    assert not bad.any(), (f, err[over].tolist(), env[over].tolist())
        flux = float(ref["weights"][ref["passed"] != 0].sum())
        only, cnt = rt.traceAxionWrapperPassed(n, seed=seed, ray_id_offset=off, flags=flags)
        assert (cnt["n_passed"], cnt["n_passed_till_window"], cnt["n_hit_nickel"]) == tuple(want.values())
        assert only.tobytes() == rec.view(np.uint8).reshape(n, 208)[rec["passed"] != 0].tobytes()
        def doors_of(mode, headroom=0):
            """Every accumulating door in one accumulation mode: counters exactly, sums against the binary128 records."""
            rt.set_accumulation_mode(mode, headroom)
            doors = {"histogram": rt.trace_histogram(n, seed=seed, ray_id_offset=off, flags=flags)[1],
                     "flux only": rt.trace_flux(n, seed=seed, ray_id_offset=off, flags=flags),
                     "spectra": rt.trace_spectra(n, seed=seed, ray_id_offset=off, flags=flags, n_radial_bins=200)[1]}
            # f64: 1e-7 (summation order, the gas stage's conditioning).  FIXED64 rounds every weight to its quantum: 0.29 quanta rms
            # per ray, so the sum of N rays is within 4 sigma = 1.2 sqrt(N) quanta of the exact one - 1e-7 of it in all but the
            # marginal setups (a few dozen passed rays a few thousand quanta each: case 8051, 1.2e-5)
            tol = 1e-7
            if mode == "fixed64" and flux > 0:
                tol = max(tol, 1.2 * np.sqrt(max(want["N_PASSED"], 1)) * rt.fixed_quanta()["weight"] / flux)
            for door, s in doors.items():
                for k, v in want.items():
                    assert s[k] == v, (mode, door, k, s[k], v)
                assert s["N_RAYS"] == n
                if flux > 0:
                    assert abs(s["SUM_WEIGHTS"] / flux - 1.0) < tol, (mode, door, s["SUM_WEIGHTS"], flux, tol)
            if gas:
                m = full.setup.m_axion
                per, shared = rt.trace_mass_scan([0.5 * m + 1e-4, m, 1.7 * m + 1e-4], n, seed=seed, ray_id_offset=off, flags=flags)
                assert per["N_PASSED"][1] == want["N_PASSED"] and shared["N_RAYS"] == n
                if flux > 0:
                    assert abs(per["SUM_WEIGHTS"][1] / flux - 1.0) < tol, (mode, "mass scan", per["SUM_WEIGHTS"][1], flux, tol)
            else:
                a0 = full.setup.telescope_turned_y_deg
                per, shared = rt.trace_angular_scan([a0, a0 + 0.02], n, seed=seed, ray_id_offset=off, flags=flags)
                assert per["N_PASSED"][0] == want["N_PASSED"] and per["N_HIT_NICKEL"][0] == want["N_HIT_NICKEL"] and shared["N_RAYS"] == n, (mode, "angular scan")
                if flux > 0:
                    assert abs(per["SUM_WEIGHTS"][0] / flux - 1.0) < tol, (mode, "angular scan", per["SUM_WEIGHTS"][0], flux, tol)
            return per

        per = doors_of("f64")
        # FIXED64 says when its quantum does not resolve the weights - and what to do about it ("a smaller headroom or
        # SART_ACCUM_F64").  Cases 5830, 6658, 7722 of the extended campaign: the X-ray test source at an energy where the window
        # passes 8e-12 / 4e-10 while the strongback - which the weight bound has to allow for, and which none of the 30 000 rays
        # meets - passes orders of magnitude more: weights 1e-9 ... 1e-11 of the bound.  The advice is followed: the smallest headroom
        # (47 fractional bits); where even that does not resolve them (5830) the f64 doors above are the result
        for headroom in (0, 16, None):
            if headroom is None:
                label += " [fixed64: unresolvable, f64 only]"
                break
            try:
                doors_of("fixed64", headroom)
                if headroom:
                    label += " [fixed64: headroom %d]" % headroom
                break
            except L.SartError as e:
                if not (e.code == L.SART_ERR_ACCUMULATOR and "quanta per passed ray" in str(e)):
                    raise
                rt.set_accumulation_mode("f64")   # (takes the reported status with it)
        rt.set_accumulation_mode("f64")      # (the integer mode rounds every weight to its quantum: not a pixel-for-pixel comparison)
        # a random image window and binning (prepareHeatmap with any nx x ny over any rectangle, :818-842): the histogram of the
        # library against the same arithmetic on its own records - floor((x - x_min) * (1 / ((x_max - x_min) / nx))) - pixel for
        # pixel in counts, and the passed rays that fall outside the window counted as such
        irng = np.random.default_rng(90_000 + case)
        chip = full.setup.chip_x_max
        nx, ny = int(irng.integers(1, 400)), int(irng.integers(1, 400))
        x0, y0 = irng.uniform(-0.2, 0.6) * chip, irng.uniform(-0.2, 0.6) * chip
        x1, y1 = x0 + irng.uniform(0.05, 1.0) * chip, y0 + irng.uniform(0.05, 1.0) * chip
        img, si = rt.trace_image(n, nx, ny, x_range=(x0, x1), y_range=(y0, y1), seed=seed, ray_id_offset=off, flags=flags)
        pr = rec[rec["passed"] != 0]
        fx = (pr["pointdataX"] - x0) * (1.0 / ((x1 - x0) / nx))
        fy = (pr["pointdataY"] - y0) * (1.0 / ((y1 - y0) / ny))
        inside = (fx >= 0.0) & (fx < nx) & (fy >= 0.0) & (fy < ny)
        want_img = np.zeros((ny, nx))
        np.add.at(want_img, (fy[inside].astype(int), fx[inside].astype(int)), pr["weights"][inside])
        assert si["N_OUTSIDE_IMAGE"] == int((~inside).sum()), ("image window", si["N_OUTSIDE_IMAGE"], int((~inside).sum()))
        assert np.array_equal(img != 0, want_img != 0), "image window: lit pixels"
        np.testing.assert_allclose(img, want_img, rtol=1e-11, atol=0, err_msg="image window")
        if not gas:   # the scan's second angle against the records of a context turned to it
            s2 = full.setup.copy()
            s2.telescope_turned_y_deg = full.setup.telescope_turned_y_deg + 0.02
            ref2 = Oracle(full, "q").trace_records(n, seed=seed, ray_id_offset=off, flags=flags, setup=s2)
            assert per["N_PASSED"][1] == int((ref2["passed"] != 0).sum()), "angular scan, second angle"
    return label, want["N_PASSED"] / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=200)
    ap.add_argument("--first", type=int, default=0)
    ap.add_argument("--rays", type=int, default=30_000)
    ap.add_argument("--full-size", action="store_true", help="full-size tables (1968 x 1500, 1000 x 1000) instead of the shrunken ones")
    ap.add_argument("--out", default="gpurun_out/fuzz_parity.txt")
    args = ap.parse_args()
    import solaraxionraytracing_amd as sa
    from solaraxionraytracing_amd import _lib as L
    from oracle.oracle import Oracle
    from tests.test_golden import compare_records
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    failed, lines, t0 = [], [], time.time()
    with open(args.out, "w") as out:
        def emit(line):
            print(line, flush=True)
            out.write(line + "\n")
            out.flush()
        emit("# tests/fuzz_parity.py: cases %d .. %d, %d rays each, %s tables, build %s" % (args.first, args.first + args.cases - 1, args.rays,
                                                                                            "full-size" if args.full_size else "shrunken", L.build_id()))
        for case in range(args.first, args.first + args.cases):
            try:
                label, frac = run_case(case, args.rays, sa, L, Oracle, compare_records, args.full_size)
                emit("case %4d ok    %-38s passed %.3f" % (case, label, frac))
            except Exception as e:   # noqa: BLE001 - the campaign goes on
                failed.append(case)
                emit("case %4d FAIL  %s" % (case, "".join(traceback.format_exception_only(type(e), e)).strip()[:600]))
                emit("".join(traceback.format_tb(e.__traceback__)[-2:]))
        emit("# %d cases, %d failed %s, %.0f s" % (args.cases, len(failed), failed, time.time() - t0))
    return 1 if failed else 0


if __name__ == "__main__":
    raise SystemExit(main())
