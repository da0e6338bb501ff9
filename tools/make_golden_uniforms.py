#!/usr/bin/env python3
"""Generates tests/golden/uniform_keyed_<setup>.npz: records of rays whose six uniforms are STORED IN THE FILE.

The seed-keyed goldens (tools/make_golden.py) had to be regenerated when round 2 re-mapped (seed, ray id) -> uniforms
(commit 906e687: two Philox blocks instead of three) - oracle, kernel and fixtures changed in one commit, so those files
could not have caught a physics regression made alongside.  These fixtures do not depend on the mapping: they hold the
uniforms themselves (draw order of SURVEY App. B: u0, u1 -> solar point :433-434, u2 -> radius CDF :436, u3 -> disc radius
:418, u4 -> disc angle :419, u5 -> energy CDF :464) and what traceAxion makes of them.

  python tools/make_golden_uniforms.py [--oracle-lib PATH/libsart_oracle.so]

--oracle-lib: another build of the oracle, driven through its per-ray entry sart_oracle_trace_axion(res, setup, tables,
flags, u[6]) - e.g. the source of the commit BEFORE the stream change built in /tmp.  The committed files were made that way
(provenance is stored in the file: key "made_with"), and tests/test_golden.py checks that the current oracle reproduces
them bit for bit: the stream change left the physics untouched."""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

from oracle.oracle import Oracle
from solaraxionraytracing_amd._lib import AXION_DTYPE, Setup
from tests.conftest import make_setup

SETUPS = ["babyiaxo_xmm", "babyiaxo_xmm_gas", "babyiaxo_xmm_rot", "cast_llnl", "cast_abrixas", "babyiaxo_xmm_xray"]
N = 3000
FIELDS = ["passed", "passedTillWindow", "hitNickel", "pointdataX", "pointdataY", "pointdataR", "weights", "transmissionMagnet",
          "yawAngles", "reflect", "energiesAx", "energiesPre", "emratesPre", "shellNumber", "kindsWindow", "deviationDet", "pointdataXBefore",
          "pointdataYBefore"]


def uniforms_for(name: str) -> np.ndarray:
    """N x 6 uniforms in [0, 1): a fixed pseudo-random set (PCG64, seed from the setup name) with the disc radius u3 of a third
    of the rays pushed towards the bore wall / the pipe edge, where most of the reference's cuts sit."""
    rng = np.random.Generator(np.random.PCG64(abs(hash_name(name))))
    u = rng.random((N, 6))
    u[N // 3: 2 * N // 3, 3] = 1.0 - u[N // 3: 2 * N // 3, 3] ** 3 * 0.6      # large bore-exit radii
    u[2 * N // 3:, 5] = 1.0 - u[2 * N // 3:, 5] ** 4                            # the steep end of the energy CDFs
    return np.minimum(u, np.nextafter(1.0, 0.0))


def hash_name(name: str) -> int:
    h = 1469598103934665603
    for ch in name.encode():
        h = ((h ^ ch) * 1099511628211) % (1 << 63)
    return h


def trace_with_library(lib_path: str, o: Oracle, u: np.ndarray, flags: int) -> np.ndarray:
    lib = C.CDLL(lib_path)
    lib.sart_oracle_trace_axion.restype = None
    lib.sart_oracle_trace_axion.argtypes = [C.c_void_p, C.POINTER(Setup), C.c_void_p, C.c_uint32, C.POINTER(C.c_double)]
    buf = np.zeros(u.shape[0], dtype=AXION_DTYPE)
    base = buf.ctypes.data
    for i in range(u.shape[0]):
        row = np.ascontiguousarray(u[i])
        lib.sart_oracle_trace_axion(C.c_void_p(base + i * AXION_DTYPE.itemsize), C.byref(o.full.setup), C.byref(o.tables), flags,
                                    row.ctypes.data_as(C.POINTER(C.c_double)))
    return buf


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--oracle-lib", default=None)
    ap.add_argument("--made-with", default=None, help="provenance string stored in the files")
    args = ap.parse_args()
    out_dir = os.path.join(ROOT, "tests", "golden")
    for name in SETUPS:
        full = make_setup(name)
        o = Oracle(full)
        u = uniforms_for(name)
        rec = trace_with_library(args.oracle_lib, o, u, full.flags) if args.oracle_lib else o.trace_records_uniforms(u, n_threads=1)
        data = {"uniforms": u, "flags": np.array([full.flags]),
                "made_with": np.array([args.made_with or ("oracle/sart_oracle.c of the working tree" if not args.oracle_lib else args.oracle_lib)]),
                "inputs_checksum": np.array([full.fluxRadiusCDF.sum(), full.diffFluxCDFs.sum(), full.reflectivity.data.sum(),
                                             full.detector_tables.window.sum()])}
        data.update({"rec_" + f: rec[f] for f in FIELDS})
        np.savez_compressed(os.path.join(out_dir, "uniform_keyed_%s.npz" % name), **data)
        print(name, "passed", int(rec["passed"].sum()), "/", N, "nickel", int(rec["hitNickel"].sum()))


if __name__ == "__main__":
    main()
