"""Every entry point of include/sart.h's hot path on every setup the tests know, in both accumulation modes: the same rays must
give the same counters whichever door they come through.  The per-feature tests check each door against the oracle on the setups
the feature was built for; this matrix is for the combinations nobody thought of (round 5: a flux-only launch with the X-ray test
source walked into a pilot launch on an image that does not exist)."""
import numpy as np
import pytest

import solaraxionraytracing_amd as sa
from solaraxionraytracing_amd import _lib as L
from tests.conftest import SETUP_NAMES, make_setup

pytestmark = pytest.mark.gpu

N, SEED, OFF = 300_000, 21, 12_345
COUNTERS = ("N_RAYS", "N_PASSED", "N_PASSED_TILL_WINDOW", "N_HIT_NICKEL")


@pytest.mark.parametrize("name", SETUP_NAMES + ["babyiaxo_xmm_gas_xray"])
@pytest.mark.parametrize("mode", ["f64", "fixed64"])
def test_every_door_counts_the_same_rays(name, mode):
    if name == "babyiaxo_xmm_gas_xray":
        full = make_setup("babyiaxo_xmm_xray", stage=L.SK_GAS)
    else:
        full = make_setup(name)
    gas = full.setup.stage == L.SK_GAS
    with sa.RayTracer(full) as rt:
        rt.set_accumulation_mode(mode)
        # the record path = the reference's own shape: the counts everything else is held against
        rec = rt.traceAxionWrapper(N, seed=SEED, ray_id_offset=OFF)
        want = {"N_RAYS": N, "N_PASSED": int((rec["passed"] != 0).sum()), "N_PASSED_TILL_WINDOW": int((rec["passedTillWindow"] != 0).sum()),
                "N_HIT_NICKEL": int((rec["hitNickel"] != 0).sum())}
        flux = float(rec["weights"][rec["passed"] != 0].sum())
        assert want["N_PASSED"] > 0, "a setup nobody passes tests nothing"
        only, cnt = rt.traceAxionWrapperPassed(N, seed=SEED, ray_id_offset=OFF)
        assert (cnt["n_rays"], cnt["n_passed"], cnt["n_passed_till_window"], cnt["n_hit_nickel"]) == tuple(want[k] for k in COUNTERS)
        assert only.tobytes() == rec.view(np.uint8).reshape(N, 208)[rec["passed"] != 0].tobytes()
        doors = {
            "histogram": rt.trace_histogram(N, seed=SEED, ray_id_offset=OFF)[1],
            "histogram 64 x 64": rt.trace_histogram(N, seed=SEED, ray_id_offset=OFF, image_n=64)[1],
            "flux only": rt.trace_flux(N, seed=SEED, ray_id_offset=OFF),
            "spectra": rt.trace_spectra(N, seed=SEED, ray_id_offset=OFF, n_radial_bins=500)[1],
        }
        for door, s in doors.items():
            for k in COUNTERS:
                assert s[k] == want[k], (door, k)
            assert s["SUM_WEIGHTS"] == pytest.approx(flux, rel=1e-9), door
        # two launches into one accumulator = one launch (the split falls inside a 256-ray chunk of the shared stream)
        a = rt.trace_histogram(100_001, seed=SEED, ray_id_offset=OFF)
        b = rt.trace_histogram(N - 100_001, seed=SEED, ray_id_offset=OFF + 100_001, accumulate=True)[1]
        for k in COUNTERS:
            assert b[k] == want[k], ("split", k)
        if gas:
            m = full.setup.m_axion
            per, shared = rt.trace_mass_scan([0.5 * m, m, 2.0 * m], N, seed=SEED, ray_id_offset=OFF)
            assert shared["N_RAYS"] == N and per["N_PASSED"][1] == want["N_PASSED"]
            assert per["SUM_WEIGHTS"][1] == pytest.approx(flux, rel=1e-9)
        else:
            a0 = full.setup.telescope_turned_y_deg
            per, shared = rt.trace_angular_scan([a0, a0 + 0.01], N, seed=SEED, ray_id_offset=OFF)
            assert shared["N_RAYS"] == N
            # (the scan keeps the setup's x angle: its first angle is the setup itself)
            assert per["N_PASSED"][0] == want["N_PASSED"] and per["N_HIT_NICKEL"][0] == want["N_HIT_NICKEL"]
            assert per["SUM_WEIGHTS"][0] == pytest.approx(flux, rel=1e-9)


def test_contexts_in_concurrent_threads_are_independent():
    """include/sart.h: a context is not thread-safe, distinct contexts are independent.  Six host threads (ctypes releases the GIL
    inside a call), each with a context of its own on a different setup, trace through several doors at the same time - every
    result is what the same thread's work gives when it runs alone."""
    import threading
    names = ["babyiaxo_xmm", "cast_llnl", "babyiaxo_xmm_gas", "cast_abrixas", "babyiaxo_xmm_rot", "babyiaxo_xmm_xray"]

    def work(name, mode):
        full = make_setup(name)
        out = []
        with sa.RayTracer(full) as rt:
            rt.set_accumulation_mode(mode)
            for k in range(3):
                img, s = rt.trace_histogram(400_000 + 1000 * k, seed=4 + k, ray_id_offset=77 * k)
                out.append((img.tobytes() if mode == "fixed64" else None, {c: s[c] for c in ("N_RAYS", "N_PASSED", "N_HIT_NICKEL", "N_SHELL_SELECTED")},
                            s["SUM_WEIGHTS"]))
                rec, cnt = rt.traceAxionWrapperPassed(150_000, seed=9 + k)
                out.append((rec.tobytes(), cnt))
                if full.setup.stage == L.SK_GAS:
                    per, sh = rt.trace_mass_scan([0.001, 0.01], 200_000, seed=k)
                else:
                    per, sh = rt.trace_angular_scan([0.0, 0.02], 200_000, seed=k)
                out.append((per["N_PASSED"].tolist(), sh["N_RAYS"]))
        return out

    alone = {n: work(n, "fixed64") for n in names}
    together, errors = {}, []

    def run(n):
        try:
            together[n] = work(n, "fixed64")
        except Exception as e:   # noqa: BLE001
            errors.append((n, repr(e)))

    threads = [threading.Thread(target=run, args=(n,)) for n in names]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for n in names:
        assert len(together[n]) == len(alone[n])
        for a, b in zip(together[n], alone[n]):
            assert a == b, n


def test_concurrent_contexts_stress_rounds():
    """The case above 200 times over in a process of its own (tools/stress_threads.py).  Round 6: a fresh context zeroed its scratch
    images with hipMemset - a null-stream operation that nothing orders with the trace kernel on the context's non-blocking stream;
    with five other contexts keeping the device busy the zeroing could run after the first pixel atomics and wipe them: 5 of 2500
    rounds failed the FIXED64 conservation check before the memsets moved onto the context's stream, 0 of 2500 after
    (profiles/r06_stress_threads.txt)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "stress_threads.py"), "200"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "0 of 200 rounds failed" in out.stdout, (out.stdout[-3000:], out.stderr[-1500:])
