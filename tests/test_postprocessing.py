"""post-processing reductions of generateResultPlots (raytracer.nim:2378-2527, :887-921) —
radial / per-energy histograms accumulated on the device, containment radii, and the image CSV."""
import csv

import numpy as np
import pytest

import solaraxionraytracing_amd as sa
from solaraxionraytracing_amd import raytracer as R
from tests.conftest import make_setup


def _reference_radii(r, w):
    """generateResultPlots on explicit records (raytracer.nim:2459-2524), literally."""
    order = np.argsort(r, kind="stable")
    pointR, ws = r[order], w[order]
    n = len(pointR)
    sigma1, sigma2 = int(round(n * 0.68)), int(round(n * 0.955))
    rS1, rS2 = pointR[sigma1 - 1], pointR[sigma2 - 1]
    sumW = ws.sum()
    start = int(round(n * 0.63))
    weightSum = ws[:start + 1].sum()
    rS1W = rS2W = 0.0
    for i in range(start + 1, n):
        weightSum += ws[i]
        if weightSum < sumW * 0.68:
            rS1W = pointR[i]
        elif weightSum < sumW * 0.955:
            rS2W = pointR[i]
    return rS1, rS2, rS1W, rS2W


def test_containment_radii_from_histogram_match_sorted_records():
    from oracle.oracle import Oracle
    full = make_setup("babyiaxo_xmm")
    o = Oracle(full)
    n = 60_000
    rec = o.trace_records(n, seed=12)
    p = rec[rec["passed"] == 1]
    want = _reference_radii(p["pointdataR"], p["weights"])
    _, _, spec = o.trace_spectra(n, seed=12, n_radial_bins=10_000, radial_max=10.0)
    got = R.containment_radii(spec)
    assert spec["radial_counts"].sum() == len(p)
    # histogram resolution is one 0.001 mm bin (the reference's own plotting bin width, :2386)
    for g, w_ in zip(got, want):
        assert g == pytest.approx(w_, abs=2.5e-3), (got, want)
    assert got[0] < got[1] and got[2] < got[3]


def test_image_csv_schema(tmp_path):
    img = np.arange(16, dtype=float).reshape(4, 4)
    path = tmp_path / "axion_image_2018.csv"
    flux = R.write_image_csv(str(path), img, 14.0, 1.0, 2.0)
    assert flux == img.sum()
    rows = list(csv.reader(open(path)))
    # raytracer.nim:887-899
    assert rows[0] == ["x", "y", "photon flux", "yr0", "yr02", "x-position [mm]", "y-position [mm]", "xr", "xrneg", "yr",
                       "xr2", "xrneg2", "yr2"]
    assert len(rows) == 17
    r5 = [float(v) for v in rows[1 + 6]]             # pixel x = 2, y = 1
    assert r5[:3] == [2.0, 1.0, img[1, 2]]
    assert r5[5] == 2 * 14.0 / 4 and r5[6] == 1 * 14.0 / 4
    yr0 = -1.0 + 2.0 * 6 / 15
    assert r5[3] == pytest.approx(yr0) and r5[7] == pytest.approx(np.sqrt(1 - yr0 ** 2) + 7.0) and r5[9] == pytest.approx(yr0 + 7.0)
    first, last = [float(v) for v in rows[1]], [float(v) for v in rows[-1]]
    assert first[3] == -1.0 and last[3] == 1.0 and first[4] == -2.0 and last[4] == 2.0


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["babyiaxo_xmm", "cast_llnl"])
def test_device_spectra_match_oracle(name):
    from oracle.oracle import Oracle
    full = make_setup(name)
    n = 300_000
    with sa.RayTracer(full) as rt:
        img, summ, spec = rt.trace_spectra(n, seed=31, n_radial_bins=2000, radial_max=10.0)
        img0, summ0 = rt.trace_histogram(n, seed=31)
    oimg, osumm, ospec = Oracle(full).trace_spectra(n, seed=31, n_radial_bins=2000, radial_max=10.0)
    np.testing.assert_allclose(img, img0, rtol=1e-11, atol=1e-30)      # spectra do not disturb the image
    assert spec["radial_counts"].sum() == summ["N_PASSED"] == spec["energy_counts"].sum()
    assert spec["radial_weights"].sum() == pytest.approx(summ["SUM_WEIGHTS"], rel=1e-11)
    assert spec["energy_weights"].sum() == pytest.approx(summ["SUM_WEIGHTS"], rel=1e-11)
    # per-energy spectra: energy indices are exact, so only rays flipping `passed` within the oracle's noise differ
    assert np.abs(spec["energy_counts"] - ospec["energy_counts"]).sum() <= max(4, 3e-4 * n)
    np.testing.assert_allclose(spec["energy_weights"].sum(), ospec["energy_weights"].sum(), rtol=1e-3)
    m = ospec["energy_counts"] > 200
    np.testing.assert_allclose(spec["energy_reflect"][m] / spec["energy_counts"][m],
                               ospec["energy_reflect"][m] / ospec["energy_counts"][m], rtol=2e-3)
    # radial histogram: 5 um bins vs ~1 um noise of the literal oracle -> compare cumulative distributions
    c, oc = np.cumsum(spec["radial_counts"]), np.cumsum(ospec["radial_counts"])
    assert np.abs(c - oc).max() <= 2e-3 * oc[-1]
    got, want = R.containment_radii(spec), R.containment_radii(ospec)
    for g, w_ in zip(got, want):
        assert g == pytest.approx(w_, abs=0.011)
