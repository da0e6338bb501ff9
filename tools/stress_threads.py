#!/usr/bin/env python3
"""Stress form of tests/test_gpu_matrix.py::test_contexts_in_concurrent_threads_are_independent: six host threads, a context each,
several doors at once, repeated ROUNDS times in one process; prints every failure with the library's message (the FIXED64
conservation check reports its sums).   python tools/stress_threads.py [rounds]"""
import os
import sys
import threading
import time
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import solaraxionraytracing_amd as sa
from solaraxionraytracing_amd import _lib as L
from tests.conftest import make_setup

NAMES = ["babyiaxo_xmm", "cast_llnl", "babyiaxo_xmm_gas", "cast_abrixas", "babyiaxo_xmm_rot", "babyiaxo_xmm_xray"]


def work(name, mode="fixed64"):
    full = make_setup(name)
    out = []
    with sa.RayTracer(full) as rt:
        rt.set_accumulation_mode(mode)
        for k in range(3):
            img, s = rt.trace_histogram(400_000 + 1000 * k, seed=4 + k, ray_id_offset=77 * k)
            out.append((img.tobytes(), {c: s[c] for c in ("N_RAYS", "N_PASSED", "N_HIT_NICKEL", "N_SHELL_SELECTED")}, s["SUM_WEIGHTS"]))
            rec, cnt = rt.traceAxionWrapperPassed(150_000, seed=9 + k)
            out.append((rec.tobytes(), cnt))
            if full.setup.stage == L.SK_GAS:
                per, sh = rt.trace_mass_scan([0.001, 0.01], 200_000, seed=k)
            else:
                per, sh = rt.trace_angular_scan([0.0, 0.02], 200_000, seed=k)
            out.append((per["N_PASSED"].tolist(), sh["N_RAYS"]))
    return out


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    alone = {n: work(n) for n in NAMES}
    bad = 0
    for r in range(rounds):
        together, errors = {}, []

        def run(n):
            try:
                together[n] = work(n)
            except Exception as e:   # noqa: BLE001
                errors.append((n, repr(e), traceback.format_exc()[-800:]))

        t0 = time.time()
        threads = [threading.Thread(target=run, args=(n,)) for n in NAMES]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        for n, e, tb in errors:
            print("round %d: %s FAILED: %s\n%s" % (r, n, e, tb), flush=True)
        for n in NAMES:
            if n in together:
                for i, (a, b) in enumerate(zip(together[n], alone[n])):
                    if a != b:
                        print("round %d: %s result %d differs from the single-threaded run" % (r, n, i), flush=True)
                        errors.append((n, "mismatch", ""))
        bad += bool(errors)
        print("round %d: %s in %.1f s" % (r, "FAILED" if errors else "ok", time.time() - t0), flush=True)
    print("%d of %d rounds failed" % (bad, rounds))
    raise SystemExit(1 if bad else 0)


if __name__ == "__main__":
    main()
