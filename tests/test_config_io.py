"""the reference's config / file surface on the host side (TOML keys of config_default.toml,
HDF5 reflectivity schema, solar-model CSV)."""
import os
import subprocess

import numpy as np
import pytest

from solaraxionraytracing_amd import _lib as L, config, tables

# same sections and keys as config/config_default.toml:1-50 (values differ: CAST/LLNL with the magnet block enabled)
SAMPLE = """
[Resources]
resourcePath   = "res"
outputPath     = "out"
llnlEfficiency = "llnl_xray_telescope_cast_effective_area_parallel_light_DTU_thesis.csv"
goldFilePrefix = "henke_download/"
rawSolarModel  = "AGSS09_solar_model_stripped.dat"
solarModelFile = "solar_model_dataframe.csv"
llnlReflFile   = "llnl_layer_reflectivities.h5"
goldReflFile   = "gold_0.25microns_reflectivities.h5"

[ReadOpacityFile]
solarModelFile = "solar_model_dataframe.csv"
opcdPath       = "OPCD"

[Setup]
experimentSetup = "CAST"
detectorSetup   = "InGrid2018"
stageSetup      = "vacuum"
telescopeSetup  = "LLNL"

[Magnet]
useConfig = true
B = 8.5
radiusCB = 21.0
lengthColdbore = 9756.0
lengthB = 9260.0
pGasRoom = 1.0
tGas = 1.7

[TestXraySource]
useConfig = false
active = true
parallel = false
energy = 1.0
distance = 2000.0
radius = 350.0
offAxisUp = 0.0
offAxisLeft = 0.0
activity = 0.125
lengthCol = 0.021

[DetectorInstallation]
useConfig = false
distanceDetectorXRT = 1485.0
distanceWindowFocalPlane = 0.0
lateralShift = 0.0
transversalShift = 0.0
"""


def _reference_config():
    import json
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_constants.json")) as f:
        return json.load(f)["config_default_toml"]


def _toml_text(cfg: dict) -> str:
    def lit(v):
        return ("true" if v else "false") if isinstance(v, bool) else ('"%s"' % v if isinstance(v, str) else repr(float(v)))
    return "\n".join("[%s]\n%s\n" % (sec, "\n".join("%s = %s" % (k, lit(v)) for k, v in body.items())) for sec, body in cfg.items())


def test_config_key_set_is_the_reference_files():
    """config/config_default.toml:1-50, pinned like the constants (tools/make_reference_constants.py reads the file in the build
    container; sections, keys and default values as data): config.py reads exactly that key set - nothing the reference has is
    unknown here, nothing here is unknown to the reference - and the hand-written SAMPLE above has it too."""
    import tomli
    ref = _reference_config()
    assert {sec: set(body) for sec, body in ref.items()} == {sec: set(body) for sec, body in config.CONFIG_KEYS.items()}
    assert {sec: set(body) for sec, body in tomli.loads(SAMPLE).items()} == {sec: set(body) for sec, body in ref.items()}
    assert config.check_keys(ref) == []
    assert config.check_keys({"Setup": {"telescopSetup": "XMM"}, "Plots": {}}) == ["[Setup].telescopSetup", "[Plots]"]
    # the value types the parsers expect (raytracer.nim:1040-1096: getFloat / getBool / getStr)
    for sec in ("Magnet", "TestXraySource", "DetectorInstallation"):
        for k, v in ref[sec].items():
            assert isinstance(v, bool) if k in ("useConfig", "active", "parallel") else isinstance(v, float), (sec, k, v)


def test_reference_defaults_build_the_babyiaxo_setup_and_trace_on_the_cpu(tmp_path):
    """BASELINE configs[0]: config_default.toml, 1e5 rays on the CPU path (plumbing, no GPU).  The reference's default VALUES
    (fixture) written back as a config.toml drive init_full_setup_from_config to BabyIAXO / InGridIAXO / vacuum / XMM with every
    optional block off (useConfig = false: the magnet is the builder's BabyIAXO magnet, not the [Magnet] block's 350 mm bore), all
    paths resolve as the reference resolves them, and 1e5 rays go through the CPU restatement of traceAxion."""
    from oracle.oracle import Oracle
    ref = _reference_config()
    cfgdir = tmp_path / "config"
    cfgdir.mkdir()
    p = cfgdir / "config.toml"
    p.write_text(_toml_text(ref))
    full = config.init_full_setup_from_config(str(p), n_radii=400, n_energies=300, refl_n_angles=200, refl_n_energies=200)
    s = full.setup
    assert (s.experiment, s.detector_kind, s.stage, s.telescope_kind) == (L.ES_BABYIAXO, L.DK_INGRIDIAXO, L.SK_VACUUM, L.TK_XMM)
    assert s.test_active == 0 and s.n_shells == 58
    assert s.magnet_radiusCB == 500.0 != ref["Magnet"]["radiusCB"]       # initMagnet's BabyIAXO bore (raytracer.nim:1113): block not read
    assert not any("nothing reads" in n for n in full.meta["notes"])
    paths = config.resolve_resources(ref, str(cfgdir))
    assert paths["resourcePath"] == str(tmp_path / "resources") and full.outpath == str(tmp_path / "out")
    assert paths["goldReflFile"] == str(tmp_path / "resources" / "gold_0.25microns_reflectivities.h5")
    assert set(paths) == set(ref["Resources"])
    n = 100_000
    img, summ, _ = Oracle(full).trace_histogram(n, seed=299792458)
    assert summ["N_RAYS"] == n and 0.19 * n < summ["N_PASSED"] < 0.27 * n
    assert img.sum() == pytest.approx(summ["SUM_WEIGHTS"], rel=1e-12) and summ["SUM_WEIGHTS"] > 0
    # with --magnet (or useConfig = true) the block IS read: the reference's default [Magnet] is a 350 mm bore
    full_m = config.init_full_setup_from_config(str(p), config.flags_from_cli(magnet=True), n_radii=60, n_energies=50,
                                                refl_n_angles=30, refl_n_energies=30)
    assert (full_m.setup.magnet_radiusCB, full_m.setup.magnet_B, full_m.setup.magnet_tGas) == (350.0, 2.0, 100.0)


def _h5_available():
    try:
        tables.write_reflectivity_h5("/tmp/_sart_probe.h5", tables.analytic_reflectivity_grid(1, 4, 4))
        return True
    except L.SartError:
        return False


def test_config_toml_drives_init_full_setup(tmp_path):
    cfgdir = tmp_path / "config"
    cfgdir.mkdir()
    (tmp_path / "config" / "res").mkdir()
    p = cfgdir / "config.toml"
    p.write_text(SAMPLE)
    full = config.init_full_setup_from_config(str(p), n_radii=60, n_energies=50, refl_n_angles=30, refl_n_energies=30)
    s = full.setup
    assert (s.experiment, s.telescope_kind, s.detector_kind, s.stage) == (L.ES_CAST, L.TK_LLNL, L.DK_INGRID2018, L.SK_VACUUM)
    assert (s.magnet_B, s.magnet_radiusCB) == (8.5, 21.0)            # [Magnet] useConfig = true (raytracer.nim:1040)
    assert s.test_active == 0 and s.distance_detector_xrt == 1485.0
    assert len(full.meta["notes"]) == 2 and full.reflectivity.data.shape[0] == 4   # files absent -> synthetic stand-ins
    # --magnet / --detectorInstall / --xrayTest flags force the blocks (raytracer.nim:1040, 1061, 1086)
    fl = config.flags_from_cli(xrayTest=True, detectorInstall=True)
    assert fl == L.CF_XRAY_TEST | L.CF_READ_DET_INSTALL_CONFIG
    full2 = config.init_full_setup_from_config(str(p), fl, n_radii=60, n_energies=50, refl_n_angles=30, refl_n_energies=30)
    assert full2.setup.test_active == 1 and full2.setup.test_parallel == 0 and full2.setup.test_length_col == 0.021
    # bad enum -> ValueError like parseEnum (:1027)
    p.write_text(SAMPLE.replace('"LLNL"', '"Chandra"'))
    with pytest.raises(ValueError):
        config.init_full_setup_from_config(str(p))


@pytest.mark.skipif(not _h5_available(), reason="libhdf5 not loadable")
def test_h5_reflectivity_roundtrip_and_schema(tmp_path):
    g = tables.llnl_reflectivity_grids(37, 29)
    path = str(tmp_path / "llnl_layer_reflectivities.h5")
    tables.write_reflectivity_h5(path, g)
    r = tables.read_reflectivity_h5(path)
    np.testing.assert_array_equal(r.data, g.data)
    assert (r.angle_min, r.angle_max, r.energy_min, r.energy_max) == (0.0, 1.5, 0.03, 15.0)
    if os.path.exists("/opt/conda/bin/h5dump"):
        hdr = subprocess.run(["/opt/conda/bin/h5dump", "-H", path], capture_output=True, text=True).stdout
        # datasets of tools/llnl_layer_reflectivity.nim:62-80: (nE,1), (nA,1), Reflectivity0..3 declared (nE, nA)
        for name in ("Energy", "Angles", "Reflectivity0", "Reflectivity3"):
            assert 'DATASET "%s"' % name in hdr
        assert "( 29, 37 )" in hdr and "( 37, 1 )" in hdr and "( 29, 1 )" in hdr
    single = tables.gold_reflectivity_grid(11, 13)
    path2 = str(tmp_path / "gold_0.25microns_reflectivities.h5")
    tables.write_reflectivity_h5(path2, single)
    assert tables.read_reflectivity_h5(path2).data.shape == (1, 11, 13)
    with pytest.raises(L.SartError):
        tables.read_reflectivity_h5(str(tmp_path / "missing.h5"))


@pytest.mark.skipif(not _h5_available(), reason="libhdf5 not loadable")
def test_config_picks_up_real_files_when_present(tmp_path):
    cfgdir = tmp_path / "config"
    res = cfgdir / "res"
    res.mkdir(parents=True)
    (cfgdir / "config.toml").write_text(SAMPLE.replace("useConfig = true", "useConfig = false"))
    tables.write_reflectivity_h5(str(res / "llnl_layer_reflectivities.h5"), tables.analytic_reflectivity_grid(4, 20, 16))
    radii, energies = tables.solar_grid(12, 9)
    with open(res / "solar_model_dataframe.csv", "w") as f:
        f.write("Radius,Energy [keV],emRates\n")
        for r_ in radii:
            for e in energies:
                f.write("%s,%s,%s\n" % (repr(float(r_)), repr(float(e)), repr(float(1.0 + e))))
    full = config.init_full_setup_from_config(str(cfgdir / "config.toml"))
    assert full.meta["notes"] == [] and full.diffFluxCDFs.shape == (12, 9) and full.reflectivity.data.shape == (4, 20, 16)
    assert full.setup.magnet_B == 9.0


def test_solar_model_csv_round_trip(tmp_path):
    from solaraxionraytracing_amd import tables
    radii, energies = tables.solar_grid(12, 9)
    em = np.random.default_rng(2).random((12, 9)) * 1e-36
    path = str(tmp_path / "solar_model_dataframe.csv")
    tables.write_solar_model_csv(path, radii, energies, em)
    assert open(path).readline().strip() == "Radius,Energy [keV],emRates"
    r2, e2, em2 = tables.read_solar_model_csv(path)
    assert np.array_equal(r2, radii) and np.array_equal(e2, energies) and np.array_equal(em2, em)


def test_cli_parser_mirrors_the_reference_main():
    """`proc main` (raytracer.nim:2817-2826): same switches and defaults."""
    from solaraxionraytracing_amd import __main__ as cli
    a = cli.build_parser().parse_args([])
    assert (a.ignoreDetWindow, a.ignoreGasAbs, a.ignoreConvProb, a.ignoreReflection, a.xrayTest, a.detectorInstall, a.magnet,
            a.noPlots) == (False,) * 8
    assert (a.angularScanMin, a.angularScanMax, a.numAngularScanPoints, a.config, a.configPath) == (0.0, 0.0, 50, "", "")
    a = cli.build_parser().parse_args(["--ignoreDetWindow", "--xrayTest", "--angularScanMax", "0.3", "--numAngularScanPoints", "7"])
    full, flags = cli.setup_from_args(a)
    assert flags == (L.CF_IGNORE_DET_WINDOW | L.CF_XRAY_TEST) and full.setup.test_active == 1


def test_incomplete_opcd_directory_degrades_with_a_note_instead_of_failing(tmp_path):
    """a config.toml whose opcdPath merely EXISTS (fm01.mesh there, the rest missing or unreadable) used to switch
    the whole setup to the OPCD path, which then failed hard.  From a config file the OPCD term is an opportunistic upgrade:
    the setup comes up with the analytic emission terms and says so; asked for explicitly (opcd_path=...) it still raises.
    A setup that keeps its CDFs on the device tells a host-side consumer what to do instead of failing with a None."""
    from oracle.oracle import Oracle
    cfgdir = tmp_path / "config"
    (cfgdir / "res").mkdir(parents=True)
    mono = cfgdir / "OPCD" / "OPCD_3.3" / "mono"
    mono.mkdir(parents=True)
    (mono / "fm01.mesh").write_text("not a mesh file\n")
    p = cfgdir / "config.toml"
    p.write_text(SAMPLE)
    full = config.init_full_setup_from_config(str(p), n_radii=60, n_energies=50, refl_n_angles=30, refl_n_energies=30)
    notes = " | ".join(full.meta["notes"])
    assert "could not be loaded" in notes and "without the OPCD absorption term" in notes
    assert full.device_emission is not None and full.device_emission.get("opcd") is None
    with pytest.raises((L.SartError, OSError, ValueError)):
        sa_init = __import__("solaraxionraytracing_amd").initFullSetup
        sa_init(L.ES_CAST, L.DK_INGRID2018, L.SK_VACUUM, L.TK_LLNL, emission="agss09-device", opcd_path=str(cfgdir / "OPCD"),
                n_radii=60, n_energies=50, refl_n_angles=30, refl_n_energies=30)
    with pytest.raises(RuntimeError) as e:
        Oracle(full)
    assert "fetch_solar_tables" in str(e.value)


@pytest.mark.skipif(not _h5_available(), reason="libhdf5 not loadable")
def test_henke_directory_converts_to_the_gold_h5_file_the_config_names(tmp_path):
    """tools/convert_reflectivities_to_h5.nim:9-48: `henke_download/<angle>degGold0.25microns.csv`, one file per
    grazing angle, sorted by the angle in the name -> `gold_0.25microns_reflectivities.h5`.  Files synthesised from the Henke
    data the package ships (data/gold_henke.npz, resampled as tables.gold_reflectivity_grid does) in the downloader's naming and
    layout round-trip to that grid, through the CLI tool, and config.py then finds the result through `goldReflFile`."""
    import sys
    g = tables.gold_reflectivity_grid(n_angles=40, n_energies=120)
    cfgdir = tmp_path / "config"
    res = cfgdir / "res"
    henke = res / "henke_download"
    res.mkdir(parents=True)
    tables.write_henke_directory(str(henke), g)
    names = sorted(os.listdir(henke))
    assert len(names) == 40 and names[0] == "0.000000degGold0.25microns.csv" and names[-1] == "1.500000degGold0.25microns.csv"
    (henke / "notes.txt").write_text("not a scan")                      # walkFiles' pattern skips what is not a scan
    out = res / "gold_0.25microns_reflectivities.h5"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "convert_reflectivities_to_h5.py"), "--indir", str(henke), "--out", str(out)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    back = tables.read_reflectivity_h5(str(out))
    assert back.data.shape == (1, 40, 120) and np.array_equal(back.data, g.data)
    assert (back.angle_min, back.angle_max, back.energy_min, back.energy_max) == (0.0, 1.5, 0.03, 15.0)
    # the grid does not depend on the order the directory lists its files in, only on the angles in their names
    assert np.array_equal(tables.henke_directory_to_grid(str(henke)).data, g.data)
    # config.py: a BabyIAXO / XMM setup reads its reflectivity from that file (goldReflFile, raytracer.nim:1193-1194)
    toml = SAMPLE.replace('experimentSetup = "CAST"', 'experimentSetup = "BabyIAXO"').replace('telescopeSetup  = "LLNL"', 'telescopeSetup  = "XMM"')
    toml = toml.replace('detectorSetup   = "InGrid2018"', 'detectorSetup   = "InGridIAXO"').replace("useConfig = true", "useConfig = false")
    (cfgdir / "config.toml").write_text(toml)
    full = config.init_full_setup_from_config(str(cfgdir / "config.toml"))
    assert np.array_equal(full.reflectivity.data, g.data) and not [n for n in full.meta["notes"] if "goldReflFile" in n]
    # what the reference's converter would choke on, it says
    (henke / "abcdegGold0.25microns.csv").write_text("#PhotonEnergy(eV) Reflectivity\n# x\n1 1\n")
    with pytest.raises(IOError):
        tables.henke_directory_to_grid(str(henke))
