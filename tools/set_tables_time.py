#!/usr/bin/env python3
"""What it costs to hand a context its solar sampling tables (1968 radii x 1500 energies), three ways:
  host     the emission table on the host -> sart_host_build_cdfs -> sart_set_solar_tables (guides built on the host, 38 MB uploaded)
  device   the emission table already on the device -> sart_set_solar_tables_device (CDFs + guides built there, 8 bytes read back)
  one-shot sart_emission_to_solar_tables: AGSS09 zones -> emission kernel -> CDFs -> guides, nothing but 50 KB of zones uploaded
  python tools/set_tables_time.py [--out gpurun_out/set_tables_time.json]"""
import argparse
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/set_tables_time.json")
    args = ap.parse_args()
    import torch
    import solaraxionraytracing_amd as sa
    from solaraxionraytracing_amd import _lib as L, emission, tables
    radii, energies, em = emission.agss09_emission_table()
    full = sa.initFullSetup(emission=em)
    res = {"n_radii": int(em.shape[0]), "n_energies": int(em.shape[1])}
    with sa.RayTracer(full) as rt:
        lib, h = rt.lib, rt.handle
        d_em = torch.from_numpy(em).to("cuda:0")
        torch.cuda.synchronize()
        for rep in range(3):
            t0 = time.perf_counter()
            rcdf, ecdf = tables.build_cdfs(em, radii, energies)
            t1 = time.perf_counter()
            L.check(lib.sart_set_solar_tables(h, L.as_dp(rcdf), L.as_dp(ecdf), L.as_dp(np.ascontiguousarray(energies)), em.shape[0], em.shape[1]))
            t2 = time.perf_counter()
            rt.set_solar_tables_device(d_em.data_ptr(), radii, energies)
            t3 = time.perf_counter()
            zones = emission.solar_zones()
            p = emission.default_params()
            t4 = time.perf_counter()
            L.check(lib.sart_emission_to_solar_tables(h, zones, len(zones), L.as_dp(np.ascontiguousarray(energies)), energies.size, None, C.byref(p)))
            t5 = time.perf_counter()
        # if the table had to come back from the device first (what round 2's configs[4] pipeline did)
        t6 = time.perf_counter()
        back = d_em.cpu().numpy()
        t7 = time.perf_counter()
        res.update({"host_build_cdfs_ms": (t1 - t0) * 1e3, "host_set_solar_tables_ms": (t2 - t1) * 1e3, "device_to_host_copy_of_the_table_ms": (t7 - t6) * 1e3,
                    "device_set_solar_tables_device_ms": (t3 - t2) * 1e3, "one_shot_emission_to_solar_tables_ms": (t5 - t4) * 1e3,
                    "emission_kernel_ms": emission.last_kernel_ms()})
        assert back.shape == em.shape
    print(json.dumps(res, indent=1))
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    json.dump(res, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
