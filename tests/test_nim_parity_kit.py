"""The parity kit for a Nim owner (tools/make_nim_parity_kit.py, integration/dump_axions.nim, tools/nim_raw_to_npz.py): the
kit's files - the reference's OWN input formats (solar_model_dataframe.csv raytracer.nim:2647-2668, reflectivity H5
:1174-1209, the TSVs :1499-1506, config.toml) - must carry exactly the tables the oracle's fixtures use, so that a Nim run on
them and the oracle's nim-stream mode trace the same rays.  Checked here without Nim: the files round-trip through this
repository's own readers bit for bit, the oracle on the re-read tables reproduces a COMMITTED record sample
(tests/golden/oracle_nim_stream_sample.npz), and the raw -> npz converter produces what tests/test_nim_stream.py reads."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

from solaraxionraytracing_amd import _lib as L, tables
from tests.conftest import SMALL, make_setup

GOLD = os.path.join(ROOT, "tests", "golden", "oracle_nim_stream_sample.npz")
SETUPS = ["babyiaxo_xmm", "cast_llnl"]


@pytest.fixture(scope="module")
def kit(tmp_path_factory):
    import make_nim_parity_kit as K
    try:
        L.load_host().sart_host_h5_write_reflectivity
    except Exception:
        pytest.skip("libsart_host not built")
    out = str(tmp_path_factory.mktemp("kit"))
    try:
        for name in SETUPS:
            K.write_setup(name, out)
    except L.SartError as e:
        if e.code == -4:
            pytest.skip("libhdf5 not available: " + str(e))
        raise
    return K, out


@pytest.mark.parametrize("name", SETUPS)
def test_kit_files_round_trip_bit_for_bit(kit, name):
    K, out = kit
    full = make_setup(name)
    res = os.path.join(out, name, "resources")
    # solar model CSV -> the same emission table -> the same CDFs
    radii, energies, em = tables.read_solar_model_csv(os.path.join(res, "solar_model_dataframe.csv"))
    r0, e0, em0 = K.emission_of(full, SMALL["n_radii"], SMALL["n_energies"])
    assert np.array_equal(radii, r0) and np.array_equal(energies, e0) and np.array_equal(em, em0)
    rcdf, ecdf = tables.build_cdfs(em, radii, energies)
    assert np.array_equal(rcdf, full.fluxRadiusCDF) and np.array_equal(ecdf, full.diffFluxCDFs)
    # reflectivity H5 (schema of raytracer.nim:1174-1209): same grid, same axis ends, same number of coatings
    h5 = "llnl_layer_reflectivities.h5" if name == "cast_llnl" else "gold_0.25microns_reflectivities.h5"
    g = tables.read_reflectivity_h5(os.path.join(res, h5))
    r = full.reflectivity
    assert g.data.shape == r.data.shape and np.array_equal(g.data, r.data)
    assert (g.angle_min, g.angle_max, g.energy_min, g.energy_max) == (r.angle_min, r.angle_max, r.energy_min, r.energy_max)
    # the whole directory through config.toml, as the reference would read it
    again = K.load_kit_setup(name, out)
    assert bytes(again.setup) == bytes(full.setup)
    assert np.array_equal(again.energies, full.energies) and np.array_equal(again.diffFluxCDFs, full.diffFluxCDFs)
    for f in ("x_kev", "strongback", "window", "gas_x_kev", "gas_absorption"):
        assert np.array_equal(getattr(again.detector_tables, f), getattr(full.detector_tables, f)), f
    assert not again.meta["notes"], again.meta["notes"]           # nothing fell back to a synthetic stand-in
    # the enum spellings of [Setup] are the reference's (raytracer.nim:16-41, :164-167)
    cfg = open(os.path.join(out, name, "config.toml")).read()
    for word in K.SETUP_BLOCKS[name]:
        assert '"%s"' % word in cfg


@pytest.mark.parametrize("name", SETUPS)
def test_oracle_on_the_kit_tables_reproduces_the_committed_sample(kit, name):
    from oracle.oracle import Oracle
    K, out = kit
    gold = np.load(GOLD, allow_pickle=False)
    n = int(gold["n_rays"])
    o = Oracle(K.load_kit_setup(name, out))
    for variant in (0, 1):
        rec = o.trace_records_nim_stream(n, init_variant=variant)
        for f in K.SAMPLE_FIELDS:
            want = gold["%s_v%d_%s" % (name, variant, f)]
            if want.dtype.kind == "f":
                np.testing.assert_array_equal(rec[f], want, err_msg="%s %s v%d" % (name, f, variant))
            else:
                np.testing.assert_array_equal(rec[f].astype(want.dtype), want)
    # and the kit's own sample file says the same
    s = np.load(os.path.join(out, name, "oracle_sample.npz"), allow_pickle=False)
    assert str(s["setup"]) == name and np.array_equal(s["v1_passed"][:n], gold["%s_v1_passed" % name])


def test_raw_dump_converter_makes_the_fixture_test_nim_stream_reads(kit, tmp_path):
    """A stand-in for the Nim run: the oracle's nim-stream records written as the raw dump integration/dump_axions.nim writes
    (208-byte records), converted, and read back the way tests/test_nim_stream.py reads tests/golden/nim_*.npz."""
    import nim_raw_to_npz as C
    from oracle.oracle import Oracle
    K, out = kit
    name = "babyiaxo_xmm"
    rec = Oracle(make_setup(name)).trace_records_nim_stream(30_000, init_variant=1)
    raw = tmp_path / "axions.raw"
    rec.tofile(str(raw))
    assert os.path.getsize(raw) == 30_000 * 208
    back = C.read_raw(str(raw), 25_000)
    assert back.size == 25_000 and back.tobytes() == rec[:25_000].tobytes()
    assert C.guess_variant(back, os.path.join(out, name, "oracle_sample.npz")) == 1
    npz = tmp_path / "nim_babyiaxo_xmm.npz"
    C.to_npz(back, name, 299792458, 0, 1, str(npz))
    g = np.load(str(npz), allow_pickle=False)
    assert [int(x) for x in g["meta"]] == [25_000, 299792458, 0, 1] and str(g["setup"]) == name
    for f in ("passed", "passedTillWindow", "hitNickel", "shellNumber", "pointdataX", "pointdataY", "weights", "energiesPre"):
        np.testing.assert_array_equal(g["rec_" + f], rec[f][:25_000])
    with pytest.raises(ValueError):
        (tmp_path / "bad.raw").write_bytes(b"x" * 100)
        C.read_raw(str(tmp_path / "bad.raw"))
    # records of another stream are recognised as such
    other = Oracle(make_setup(name)).trace_records(25_000, seed=1)
    with pytest.raises(SystemExit):
        C.guess_variant(other, os.path.join(out, name, "oracle_sample.npz"))


def test_nim_dump_source_documents_its_hook():
    src = open(os.path.join(ROOT, "integration", "dump_axions.nim")).read()
    for needle in ("proc dumpAxions*", "sizeof(Axion) == 208", "SART_DUMP_AXIONS", "WEAVE_NUM_THREADS=1", "exit(Weave)", "writeBuffer"):
        assert needle in src, needle
