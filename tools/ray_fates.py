#!/usr/bin/env python3
"""Where the rays of a configuration end (CPU oracle, no GPU): the tally behind DESIGN.md 3.1 "Innermost-shell shortcut" and 8.1.
Fractions of all rays: reached the telescope (bore + pipes), selected a shell (opaque structures, glass fronts), hit nickel
(:2040-2046), left the second mirror (pointdataXBefore set, :2088), passed till the window, passed; what lies between
"selected" and "nickel + left the second mirror" ended at the no-hit test (:2055).

    python tools/ray_fates.py [--rays 2e6] [--out profiles/NAME.json]"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import solaraxionraytracing_amd as sa   # noqa: E402
from oracle import oracle as O   # noqa: E402
from solaraxionraytracing_amd import _lib as L   # noqa: E402


def tally(full, n, seed=3):
    orc = O.Oracle(full)
    threads = len(os.sched_getaffinity(0))
    rec = orc.trace_records(n, seed=seed, n_threads=threads)
    _, summ, _ = orc.trace_histogram(n, seed=seed, n_threads=threads)
    out = {"reached_telescope": summ["N_REACHED_TELESCOPE"] / n, "shell_selected": summ["N_SHELL_SELECTED"] / n,
           "hit_nickel": float(rec["hitNickel"].mean()), "left_second_mirror": float(np.mean(rec["pointdataXBefore"] != 0)),
           "passed_till_window": float(rec["passedTillWindow"].mean()), "passed": float(rec["passed"].mean())}
    out["ended_at_no_hit_test"] = out["shell_selected"] - out["hit_nickel"] - out["left_second_mirror"]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=float, default=2e6)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    n = int(args.rays)
    res = {"rays": n, "babyiaxo_xmm": tally(sa.initFullSetup(), n),
           "cast_llnl_gold": tally(sa.initFullSetup(L.ES_CAST, L.DK_INGRID2018, L.SK_VACUUM, L.TK_LLNL, reflectivity="gold"), n)}
    print(json.dumps(res, indent=1))
    if args.out:
        with open(args.out, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
