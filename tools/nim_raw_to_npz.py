#!/usr/bin/env python3
"""Raw `Axion` records dumped by a Nim build of the reference (integration/dump_axions.nim) -> the fixture format of
tests/test_nim_stream.py (tests/golden/nim_<setup>.npz).

  python tools/nim_raw_to_npz.py /tmp/axions_babyiaxo_xmm.raw --setup babyiaxo_xmm --rays 200000 --out tests/golden/nim_babyiaxo_xmm.npz

The raw file is `NumberOfPointsSun` (raytracer.nim:251: 1_000_000) records of 208 bytes - the C layout of `type Axion`
(raytracer.nim:192-221; include/sart.h: sart_axion_t); the first --rays of them are kept.  --init-variant: how `randomize(seed)`
seeds xoroshiro128+ (0: Nim < 1.4, 1: Nim >= 1.4); `auto` compares the first records with <kit>/<setup>/oracle_sample.npz
(tools/make_nim_parity_kit.py) under both and takes the one that matches, or fails if neither does.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

RECORD_BYTES = 208


def read_raw(path, n_rays=None):
    from solaraxionraytracing_amd._lib import AXION_DTYPE
    size = os.path.getsize(path)
    if size % RECORD_BYTES:
        raise ValueError("%s: %d bytes is not a whole number of %d-byte Axion records (was the reference built with another "
                         "field layout?)" % (path, size, RECORD_BYTES))
    n_file = size // RECORD_BYTES
    n = n_file if n_rays is None else min(int(n_rays), n_file)
    return np.fromfile(path, dtype=AXION_DTYPE, count=n)


def guess_variant(rec, sample_path):
    s = np.load(sample_path, allow_pickle=False)
    n = min(int(s["n_rays"]), rec.size)
    for v in (0, 1):
        same = all(np.array_equal(rec[f][:n], s["v%d_%s" % (v, f)][:n]) for f in ("passed", "shellNumber", "hitNickel"))
        if same and np.allclose(rec["pointdataX"][:n], s["v%d_pointdataX" % v][:n], rtol=1e-9, atol=5e-3):
            return v
    raise SystemExit("the records match the oracle's nim-stream sample under neither initRand variant: either the run did not "
                     "use one thread / the kit's tables / seed 299792458, or the oracle differs from the reference - compare "
                     "field by field (tests/test_nim_stream.py)")


def to_npz(rec, setup, seed, flags, variant, out):
    arrays = {"meta": np.array([rec.size, seed, flags, variant], dtype=np.int64), "setup": np.array(setup)}
    for name in rec.dtype.names:
        arrays["rec_" + name] = np.ascontiguousarray(rec[name])
    os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
    np.savez_compressed(out, **arrays)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("raw")
    ap.add_argument("--setup", required=True, help="conftest setup name the kit directory was made for")
    ap.add_argument("--rays", type=int, default=200_000)
    ap.add_argument("--seed", type=int, default=299792458, help="randomize(...) of raytracer.nim:276")
    ap.add_argument("--flags", type=int, default=0, help="SART_CF_* bitset of the run's command-line switches")
    ap.add_argument("--init-variant", default="auto", choices=["auto", "0", "1"])
    ap.add_argument("--kit", default="nim_parity_kit", help="directory tools/make_nim_parity_kit.py wrote (for --init-variant auto)")
    ap.add_argument("--out", required=True)
    args = ap.parse_args()
    rec = read_raw(args.raw, args.rays)
    variant = int(args.init_variant) if args.init_variant != "auto" else \
        guess_variant(rec, os.path.join(args.kit, args.setup, "oracle_sample.npz"))
    to_npz(rec, args.setup, args.seed, args.flags, variant, args.out)
    print("wrote %s: %d records, passed %.4f, init variant %d" % (args.out, rec.size, float(rec["passed"].mean()), variant))


if __name__ == "__main__":
    main()
