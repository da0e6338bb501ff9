"""Every number the host layer's setup builders hold (csrc/raytracer_host.cpp: sart_host_new_full_setup) against the numbers of
the reference's builders (src/raytracer.nim :248-272, :1098-1155, :1256-1346, :1350-1407, :1464-1490), read from the reference's
text by tools/make_reference_constants.py into tests/golden/reference_constants.json.

Why this test exists: the oracle and the HIP path are fed by the SAME product-built setup (tests/conftest.py -> initFullSetup ->
libsart_host.so), so a wrong digit in a shell table passes every GPU-vs-oracle test.  Here every field of sart_setup_t is
compared, for every experiment x telescope x detector combination the reference can build; a field nobody compares fails the test."""
import json
import math
import os

import numpy as np
import pytest

import solaraxionraytracing_amd as sa
from solaraxionraytracing_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_constants.json")))

EXPERIMENTS = {"esCAST": L.ES_CAST, "esBabyIAXO": L.ES_BABYIAXO}
TELESCOPES = {"tkLLNL": L.TK_LLNL, "tkXMM": L.TK_XMM, "tkAbrixas": L.TK_ABRIXAS}
DETECTORS = {"dkInGrid2017": L.DK_INGRID2017, "dkInGrid2018": L.DK_INGRID2018, "dkInGridIAXO": L.DK_INGRIDIAXO}


def test_the_enums_of_the_header_are_the_reference_ordinals():
    e = REF["enums"]
    assert e["ExperimentSetupKind"] == {"esCAST": L.ES_CAST, "esBabyIAXO": L.ES_BABYIAXO}
    assert e["StageKind"] == {"skVacuum": L.SK_VACUUM, "skGas": L.SK_GAS}
    assert e["TelescopeKind"] == {"tkLLNL": L.TK_LLNL, "tkXMM": L.TK_XMM, "tkCustomBabyIAXO": L.TK_CUSTOM_BABYIAXO, "tkAbrixas": L.TK_ABRIXAS,
                                  "tkOther": L.TK_OTHER}
    assert e["DetectorSetupKind"] == {"dkInGrid2017": L.DK_INGRID2017, "dkInGrid2018": L.DK_INGRID2018, "dkInGridIAXO": L.DK_INGRIDIAXO}
    assert e["HoleType"] == {"htNone": L.HT_NONE, "htCross": L.HT_CROSS, "htStar": L.HT_STAR, "htCircle": L.HT_CIRCLE, "htSquare": L.HT_SQUARE,
                             "htDiamond": L.HT_DIAMOND}
    assert e["ReflectivityKind"] == {"rkEffectiveArea": L.RK_EFFECTIVE_AREA, "rkSingleCoating": L.RK_SINGLE_COATING, "rkMultiCoating": L.RK_MULTI_COATING}


def expected_fields(es, tk, dk, stage, flags):
    """sart_setup_t field -> value, from the fixture alone."""
    mag, pipes, tel, refl = REF["magnet"][es], REF["pipes"][tk], REF["telescope"][tk], REF["reflectivity"][tk]
    src, inst, det, const = REF["testSource"][es], REF["detectorInstall"][tk], REF["detector"][dk], REF["constants"]
    n = len(tel["allR1"])
    assert n == len(tel["allThickness"]) == len(tel["allXsep"]) == len(tel["allAngles"])
    pad = lambda v: list(v) + [0.0] * (L.SART_MAX_SHELLS - len(v))
    wy = {v: k for k, v in REF["enums"]["WindowYearKind"].items()}[int(det["windowYear"])]
    win = [w for w in REF["calcWindowVals"] if (w["radiusWindow"], w["numberOfStrips"], w["openApertureRatio"]) ==
           (det["radiusWindow"], det["numberOfStrips"], det["openApertureRatio"])][0]["width_dist"]
    layers = [int(x) for x in refl.get("layers", [])]
    return {
        "experiment": EXPERIMENTS[es], "stage": stage, "telescope_kind": TELESCOPES[tk], "detector_kind": DETECTORS[dk],
        "magnet_B": mag["B"], "magnet_lengthB": mag["lengthB"], "magnet_lengthColdbore": mag["lengthColdbore"], "magnet_radiusCB": mag["radiusCB"],
        "magnet_pGasRoom": mag["pGasRoom"], "magnet_tGas": mag["tGas"],
        "pipe_cb_vt3_length": pipes["coldBoreToVT3"]["length"], "pipe_cb_vt3_radius": pipes["coldBoreToVT3"]["radius"],
        "pipe_vt3_xrt_length": pipes["vt3ToXRT"]["length"], "pipe_vt3_xrt_radius": pipes["vt3ToXRT"]["radius"],
        "pipes_turned_deg": pipes["pipesTurned"], "distance_cb_axis_xrt_axis": pipes["distanceCBAxisXRTAxis"],
        "optics_entrance": tel["optics_entrance"], "optics_exit": tel["optics_exit"],
        "telescope_turned_x_deg": tel["telescope_turned_x"], "telescope_turned_y_deg": tel["telescope_turned_y"],
        "n_shells": n, "hole_type": tel["holeType"], "number_of_holes": int(tel["numberOfHoles"]), "reflectivity_kind": refl["kind"],
        "all_r1": pad(tel["allR1"]), "all_thickness": pad(tel["allThickness"]), "all_xsep": pad(tel["allXsep"]), "all_angles_deg": pad(tel["allAngles"]),
        "l_mirror": tel["lMirror"], "hole_in_optics": tel["holeInOptics"],
        # (a single-coating telescope has no `layers` in the reference; the flattened setup holds its one grid as one layer boundary
        # behind the last shell)
        "n_coatings": len(layers) if layers else 1, "coating_layers": (layers or [n]) + [0] * (L.SART_MAX_COATINGS - len(layers or [n])),
        "distance_detector_xrt": inst["distanceDetectorXRT"], "distance_window_focal_plane": inst["distanceWindowFocalPlane"],
        "lateral_shift": inst["lateralShift"], "transversal_shift": inst["transversalShift"],
        "radius_window": det["radiusWindow"], "number_of_strips": int(det["numberOfStrips"]), "open_aperture_ratio": det["openApertureRatio"],
        "strip_dist_window": ("approx", win[1]), "strip_width_window": ("approx", win[0]),
        "theta_rad": ("approx", math.radians(REF["windowYearDeg"][wy])), "depth_det": det["depthDet"],
        "test_active": 1 if flags & L.CF_XRAY_TEST else 0, "test_parallel": int(src["parallel"]), "test_energy": src["energy"],
        "test_distance": src["distance"], "test_radius": src["radius"], "test_off_axis_up": src["offAxisUp"], "test_off_axis_left": src["offAxisLeft"],
        "test_activity": src["activity"], "test_length_col": src["lengthCol"],
        "distance_sun_earth": const["DistanceSunEarth"], "radius_sun": const["RadiusSun"], "room_temp": const["RoomTemp"], "m_axion": const["mAxion"],
        "g_agamma": const["g_agamma"], "chip_x_max": const["ChipXMax"], "chip_y_max": const["ChipYMax"],
    }


def compare(s, want, label):
    seen = set()
    for name, _ in L.Setup._fields_:
        if name.startswith("_pad"):
            continue
        got = getattr(s, name)
        got = list(got) if hasattr(got, "__len__") else got
        w = want[name]
        if isinstance(w, tuple):      # computed on both sides (calcWindowVals, degToRad): to the last few ulps
            assert got == pytest.approx(w[1], rel=1e-14), (label, name, got, w[1])
        else:
            assert got == w, (label, name, got, w)   # literals: exactly
        seen.add(name)
    assert seen == set(want), (label, set(want) ^ seen)


@pytest.mark.parametrize("es", sorted(EXPERIMENTS))
@pytest.mark.parametrize("tk", sorted(TELESCOPES))
@pytest.mark.parametrize("dk", sorted(DETECTORS))
def test_every_field_of_the_full_setup_equals_the_reference(es, tk, dk):
    for stage, flags in ((L.SK_VACUUM, 0), (L.SK_GAS, L.CF_XRAY_TEST)):
        s = sa.newFullSetup(EXPERIMENTS[es], DETECTORS[dk], stage, TELESCOPES[tk], flags)
        compare(s, expected_fields(es, tk, dk, stage, flags), (es, tk, dk, stage, flags))


def test_a_flipped_digit_is_seen():
    """The comparison is exact: one digit in one shell radius, angle or thickness turns it red."""
    s = sa.newFullSetup(L.ES_BABYIAXO, L.DK_INGRIDIAXO, L.SK_VACUUM, L.TK_XMM)
    want = expected_fields("esBabyIAXO", "tkXMM", "dkInGridIAXO", L.SK_VACUUM, 0)
    compare(s, want, "clean")
    for field, j in (("all_r1", 33), ("all_angles_deg", 17), ("all_thickness", 57), ("all_xsep", 8)):
        bad = s.copy()
        getattr(bad, field)[j] = np.nextafter(getattr(bad, field)[j], 1e9) if getattr(bad, field)[j] else 1e-3
        with pytest.raises(AssertionError):
            compare(bad, want, field)


def test_calc_window_vals_equals_the_restated_formula():
    """calcWindowVals (raytracer.nim:1431-1462) of the host library against the generator's restatement, for the reference's
    window and for others."""
    host = L.load_host()
    import ctypes as C
    for w in REF["calcWindowVals"]:
        width, dist = C.c_double(), C.c_double()
        L.check(host.sart_host_calc_window_vals(w["radiusWindow"], int(w["numberOfStrips"]), w["openApertureRatio"], C.byref(width), C.byref(dist)), host=True)
        assert width.value == pytest.approx(w["width_dist"][0], rel=1e-14) and dist.value == pytest.approx(w["width_dist"][1], rel=1e-14)
    # the reference's window: 7 mm radius, 4 strips, 83.8 % open: strips 0.5004 mm wide, 2.2996 mm apart
    assert REF["calcWindowVals"][0]["width_dist"] == pytest.approx([0.5004177831498976, 2.2995822168501023], rel=1e-15)


def test_fixture_is_what_the_generator_makes_when_the_reference_is_here(tmp_path):
    """In the build container (the reference's text present) the committed fixture is the generator's output, number for number."""
    if not os.path.exists("/root/reference/src/raytracer.nim"):
        pytest.skip("no reference here (GPU box)")
    import subprocess
    import sys
    out = tmp_path / "c.json"
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_reference_constants.py"), "--out", str(out)], check=True, capture_output=True)
    assert json.load(open(out)) == REF
