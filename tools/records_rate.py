#!/usr/bin/env python3
"""Rate of the literal drop-in sart_trace_records (208-byte Axion records into caller memory, raytracer.nim:2223-2244, :2760):
records per second into a FRESH pageable buffer (what a Nim `newSeq[Axion]` or numpy.empty hands over), into the same buffer
again (pages mapped), with the host-side pre-fault switched off, and on the device alone; and of sart_trace_records_passed
(only the records with `passed` set travel): rays per second into a fresh buffer sized for all rays / for the passed ones, into
mapped pages, and on the device alone.

  python tools/records_rate.py [--records 2e7] [--out gpurun_out/records_rate.json]"""
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
    ap.add_argument("--records", type=float, default=2e7)
    ap.add_argument("--out", default="gpurun_out/records_rate.json")
    args = ap.parse_args()
    n = int(args.records)
    import solaraxionraytracing_amd as sa
    from solaraxionraytracing_amd import _lib as L
    full = sa.initFullSetup()

    def call(rt, buf):
        p = rt.trace_params(n, seed=5)
        t0 = time.perf_counter()
        L.check(rt.lib.sart_trace_records(rt.handle, C.byref(p), buf.ctypes.data_as(C.c_void_p)))
        return time.perf_counter() - t0

    res = {"records": n, "bytes": n * 208, "pcie_bound_records_per_s": 57.1e9 / 208}
    with sa.RayTracer(full) as rt:
        warm = np.empty(200_000, dtype=L.AXION_DTYPE)
        p = rt.trace_params(warm.size)
        L.check(rt.lib.sart_trace_records(rt.handle, C.byref(p), warm.ctypes.data_as(C.c_void_p)))
        buf = np.empty(n, dtype=L.AXION_DTYPE)            # fresh mapping: no page of it exists yet
        dt = call(rt, buf)
        res["fresh_buffer"] = {"seconds": dt, "records_per_s": n / dt, "gb_per_s": n * 208 / dt / 1e9}
        first = buf[:100_000].tobytes()     # (a .copy() of a structured array does not carry the padding bytes)
        dt = call(rt, buf)
        res["same_buffer_again"] = {"seconds": dt, "records_per_s": n / dt, "gb_per_s": n * 208 / dt / 1e9}
        assert first == buf[:100_000].tobytes(), "the second call wrote other records"
        n_passed = int(buf["passed"].sum(dtype=np.int64))
        first_passed = buf[:400_000].view(np.uint8).reshape(-1, 208)[buf[:400_000]["passed"] != 0][:20_000].tobytes()
        assert n_passed > 0.2 * n * 0.9, (n_passed, n)
        del buf
        # the device side alone
        import torch
        d = torch.empty(n * 208, dtype=torch.uint8, device="cuda:0")
        rt.enable_kernel_timing(True)
        p = rt.trace_params(n, seed=5)
        L.check(rt.lib.sart_trace_records_device(rt.handle, C.byref(p), C.c_void_p(d.data_ptr())))
        ms, _ = rt.kernel_timing()
        res["device_only"] = {"seconds": ms / 1e3, "records_per_s": n / (ms / 1e3)}
    # the passed rays only (sart_trace_records_passed): rays per second through the record interface when only the records the
    # reference's consumers read cross PCIe; buffer sized for all rays (fresh pages), for the passed ones + 2 %, and reused
    os.environ.pop("SART_NO_HOST_PREFAULT", None)
    with sa.RayTracer(full) as rt:
        rt.traceAxionWrapperPassed(200_000, seed=5)
        for label, cap in (("passed_only_fresh_buffer_of_n", n), ("passed_only_fresh_buffer_fitted", int(n_passed * 1.02))):
            out = got = res_call = None       # (the buffer of the round before is unmapped here, not inside the timed call)
            out = np.empty(cap, dtype=L.AXION_DTYPE)
            t0 = time.perf_counter()
            res_call = rt.traceAxionWrapperPassed(n, seed=5, out=out)
            dt = time.perf_counter() - t0
            got, c = res_call
            assert c["n_passed"] == n_passed == len(got)
            res[label] = {"seconds": dt, "rays_per_s": n / dt, "passed_records_per_s": n_passed / dt, "gb_per_s": n_passed * 208 / dt / 1e9,
                          "passed_fraction": n_passed / n}
        t0 = time.perf_counter()
        got, c = rt.traceAxionWrapperPassed(n, seed=5, out=out)
        dt = time.perf_counter() - t0
        res["passed_only_same_buffer_again"] = {"seconds": dt, "rays_per_s": n / dt, "passed_records_per_s": n_passed / dt, "gb_per_s": n_passed * 208 / dt / 1e9}
        assert got[:20_000].tobytes() == first_passed
        import torch
        d = torch.empty(int(n_passed * 1.02) * 208, dtype=torch.uint8, device="cuda:0")
        cnt = torch.zeros(4, dtype=torch.int64, device="cuda:0")
        p = rt.trace_params(n, seed=5)
        for rep in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            rt.trace_records_passed_device(p, d.data_ptr(), int(n_passed * 1.02), cnt.data_ptr())
            rt.synchronize()
            dt = time.perf_counter() - t0
        assert cnt.tolist()[1] == n_passed
        res["passed_only_device_only"] = {"seconds": dt, "rays_per_s": n / dt}
    os.environ["SART_NO_HOST_PREFAULT"] = "1"
    with sa.RayTracer(full) as rt:
        buf = np.empty(n, dtype=L.AXION_DTYPE)
        dt = call(rt, buf)
        res["fresh_buffer_no_prefault"] = {"seconds": dt, "records_per_s": n / dt, "gb_per_s": n * 208 / dt / 1e9}
        assert first == buf[:100_000].tobytes()
    print(json.dumps(res, indent=1))
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    json.dump(res, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
