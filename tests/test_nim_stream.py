"""The oracle's second random-number mode: the reference's own stream (Nim std/random = xoroshiro128+, one global stream,
`randomize(299792458)`, raytracer.nim:276) instead of a Philox block per ray.

Purpose: nobody can build the reference in this image, so ray-for-ray parity of the oracle with the
reference is unpinned.  This mode makes the oracle reproduce what a single-threaded run of the reference computes, so that
fixtures produced by whoever has a Nim toolchain (tests/golden/nim_*.npz, format below) pin it the day they exist.

Fixture format (numpy .npz): `meta` = [n_rays, seed, flags, init_variant(0: Nim < 1.4, 1: Nim >= 1.4)], `setup` = the
conftest setup name the run used (string), one array `rec_<field>` per Axion field of raytracer.nim:192-221 (length n_rays),
produced by `WEAVE_NUM_THREADS=1 ./raytracer` with the tables of that setup.
"""
import ctypes as C
import glob
import os

import numpy as np
import pytest

from tests.conftest import make_setup

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MASK = (1 << 64) - 1


def _py_next(s):
    """Independent restatement of xoroshiro128+ (rotations 55, 14, 36) in Python integers."""
    rotl = lambda x, k: ((x << k) | (x >> (64 - k))) & MASK
    s0, s1 = s
    res = (s0 + s1) & MASK
    s1 ^= s0
    return res, [rotl(s0, 55) ^ s1 ^ ((s1 << 14) & MASK), rotl(s1, 36)]


def test_xoroshiro_known_state_sequence():
    from oracle.oracle import load
    lib = load()
    st = (C.c_uint64 * 2)(1, 2)
    assert lib.sart_oracle_nim_rand_next(C.byref(st)) == 3          # s0 + s1
    py = _py_next([1, 2])[1]
    assert [st[0], st[1]] == py
    for _ in range(1000):
        want, py = _py_next(py)
        assert lib.sart_oracle_nim_rand_next(C.byref(st)) == want
    # initRand(299792458), Nim < 1.4: a0 = seed >> 16, a1 = seed & 0xffff, one discarded draw
    lib.sart_oracle_nim_rand_init(C.byref(st), 299792458, 0)
    assert [st[0], st[1]] == _py_next([299792458 >> 16, 299792458 & 0xFFFF])[1]
    # rand(1.0): 52 mantissa bits under the exponent of 1.0, minus 1.0
    py = [st[0], st[1]]
    for _ in range(100):
        x, py = _py_next(py)
        want = np.frombuffer(np.uint64((0x3FF << 52) | (x >> 12)).tobytes(), dtype=np.float64)[0] - 1.0
        got = lib.sart_oracle_nim_rand_float(C.byref(st))
        assert got == want and 0.0 <= got < 1.0
    # the Nim >= 1.4 variant differs (2^64 jump before the discarded draw) and is deterministic
    a, b = (C.c_uint64 * 2)(), (C.c_uint64 * 2)()
    lib.sart_oracle_nim_rand_init(C.byref(a), 299792458, 1)
    lib.sart_oracle_nim_rand_init(C.byref(b), 299792458, 1)
    assert [a[0], a[1]] == [b[0], b[1]] != [st[0], st[1]]


def test_stream_mode_traces_the_same_physics_as_the_philox_mode():
    """Same algorithm, different uniforms: survival fractions and the flux agree within Monte-Carlo error, and the stream
    mode is sequential (ray i uses draws 6 i .. 6 i + 5: skipping rays = skipping draws)."""
    from oracle.oracle import Oracle
    full = make_setup("babyiaxo_xmm")
    o = Oracle(full)
    n = 60_000
    a = o.trace_records_nim_stream(n, init_variant=0)
    b = o.trace_records(n, seed=11)
    pa, pb = a["passed"].mean(), b["passed"].mean()
    assert abs(pa - pb) < 5.0 * np.sqrt(pa * (1 - pa) / n * 2)
    assert a["weights"].sum() == pytest.approx(b["weights"].sum(), rel=0.05)
    tail = o.trace_records_nim_stream(1000, ray_id_offset=n - 1000, init_variant=0)
    for f in ("passed", "pointdataX", "weights", "energiesPre"):
        np.testing.assert_array_equal(tail[f], a[f][-1000:])
    x = Oracle(make_setup("babyiaxo_xmm_xray"))
    t1 = x.trace_records_nim_stream(2000, init_variant=1)
    t2 = x.trace_records_nim_stream(500, ray_id_offset=1500, init_variant=1)      # four draws per ray of the test source
    np.testing.assert_array_equal(t1["pointdataX"][1500:], t2["pointdataX"])


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "nim_*.npz"))) or [None])
def test_oracle_reproduces_fixtures_from_a_nim_run_of_the_reference(path):
    if path is None:
        pytest.skip("no tests/golden/nim_*.npz: the reference cannot be built in this image (no Nim toolchain); "
                    "parity of the oracle with a Nim run stays unpinned until such a fixture is contributed")
    from oracle.oracle import Oracle
    g = np.load(path, allow_pickle=False)
    n, seed, flags, variant = [int(x) for x in g["meta"]]
    full = make_setup(str(g["setup"]))
    rec = Oracle(full).trace_records_nim_stream(n, seed=seed, flags=flags, init_variant=variant)
    for f in ("passed", "passedTillWindow", "hitNickel", "shellNumber"):
        np.testing.assert_array_equal(rec[f], g["rec_" + f])
    for f in ("pointdataX", "pointdataY", "weights", "energiesPre"):
        np.testing.assert_allclose(rec[f], g["rec_" + f], rtol=1e-9, atol=5e-3 if f.startswith("point") else 0)
