#!/usr/bin/env python3
"""How long the OPCD front end takes at full size (1968 zones x 1500 energies): stand-in files of the OPCD 3.3 layout for every
(temperature, density) pair of the AGSS09 model -> parse (C++ reader, threads) -> absorption-coefficient kernel -> emission
kernel -> sampling tables.  The numbers in the files are made up (opacity.write_stand_in_tree); sizes and formats are real.

    python tools/opcd_time.py [--out profiles/NAME.json] [--keep DIR]"""
import argparse
import ctypes as C
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from solaraxionraytracing_amd import _lib, emission as em, opacity as op, raytracer as rt   # noqa: E402
from solaraxionraytracing_amd.tables import solar_grid   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="")
    ap.add_argument("--keep", default="", help="write the stand-in tree here and keep it")
    ap.add_argument("--densities-per-file", type=int, default=6, help="extra density tables per file (skipped by the reader)")
    args = ap.parse_args()
    zones = em.solar_zones()
    n_z = op.number_densities()
    _, energies = solar_grid(len(zones), 1500)
    root = args.keep or tempfile.mkdtemp(prefix="opcd_")
    res = {"n_radii": len(zones), "n_energies": 1500, "build_id": _lib.build_id()}
    t0 = time.time()
    extra = tuple(range(60, 60 + 2 * args.densities_per_file, 2))
    d = op.write_stand_in_tree(root, zones, densities_extra=extra)
    res["write_stand_in_s"] = time.time() - t0
    size = int(subprocess.run(["du", "-sb", d], capture_output=True, text=True).stdout.split()[0])
    res["tree_bytes"] = size
    res["files"] = len(os.listdir(d))
    for threads in (1, 4, 16):
        t0 = time.time()
        s = op.OpcdSet(root, zones, n_threads=threads)
        res["load_s_%d_threads" % threads] = time.time() - t0
        if threads != 16:
            s.close()
    res["parse_mb_per_s_16_threads"] = size / res["load_s_16_threads"] / 1e6
    T = s.tables.contents
    res["slots"] = int(T.n_slots)
    res["opacity_values"] = int(T.n_table_y)
    lib = _lib.load_sart()
    full = rt.initFullSetup(n_radii=16, n_energies=32)
    with rt.RayTracer(full) as tracer:
        import torch
        d_abs = torch.empty(len(zones) * 1500, dtype=torch.float64, device="cuda")
        for rep in range(3):
            t0 = time.time()
            _lib.check(lib.sart_emission_abs_coefs_device(tracer.handle, zones, len(zones), _lib.as_dp(n_z), _lib.as_dp(energies), 1500,
                                                          s.tables, C.c_void_p(d_abs.data_ptr())))
            res["abs_coefs_call_s_incl_upload"] = time.time() - t0
        res["abs_coefs_kernel_ms"] = float(lib.sart_emission_abs_coefs_last_kernel_ms())
        params = em.default_params()
        for rep in range(3):
            t0 = time.time()
            _lib.check(lib.sart_emission_to_solar_tables_opcd(tracer.handle, zones, len(zones), _lib.as_dp(n_z), _lib.as_dp(energies), 1500,
                                                              s.tables, C.byref(params)))
            res["opcd_to_sampling_tables_s"] = time.time() - t0
        res["emission_kernel_ms"] = em.last_kernel_ms()
        for rep in range(3):
            t0 = time.time()
            _lib.check(lib.sart_emission_to_solar_tables(tracer.handle, zones, len(zones), _lib.as_dp(energies), 1500, None, C.byref(params)))
            res["without_opcd_to_sampling_tables_s"] = time.time() - t0
        absc = d_abs.cpu().numpy().reshape(len(zones), 1500)
    res["abs_coef_nonzero_fraction"] = float(np.mean(absc != 0.0))
    s.close()
    if not args.keep:
        shutil.rmtree(root)
    print(json.dumps(res, indent=1))
    if args.out:
        with open(args.out, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
