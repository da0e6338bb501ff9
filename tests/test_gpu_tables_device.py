"""sart_set_solar_tables_device / sart_emission_to_solar_tables — the CDF construction of initFullSetup
(raytracer.nim:2670-2705) and the guide tables in front of it, built on the device from a device-resident emission table
(csrc/sart_tables.hip).  The host path (sart_host_build_cdfs, which keeps the reference's order of operations, + the guide
construction of sart_set_solar_tables) is the checker: every table must come out BIT-IDENTICAL.  Run on the MI355X box."""
import numpy as np
import pytest

import solaraxionraytracing_amd as sa
from solaraxionraytracing_amd import _lib as L, tables

pytestmark = pytest.mark.gpu


def _tables_from(kind):
    if kind == "primakoff":                   # E1, the bench's table: 1968 x 1500
        radii, energies = tables.solar_grid()
        return radii, energies, tables.primakoff_emission_table()
    if kind == "agss09":                      # all terms of readOpacityFile.nim from the emission kernel: 1968 x 1500
        from solaraxionraytracing_amd import emission
        return emission.agss09_emission_table()
    if kind == "legacy":                      # the reference's own emission_rates_Hz.txt: 397 x 233
        return tables.legacy_emission_table()
    if kind == "ragged":                      # sizes that are no multiple of anything, steep rows (wide guide buckets)
        rng = np.random.default_rng(3)
        radii, energies = tables.solar_grid(131, 77)
        em = rng.random((131, 77)) ** 8 * np.exp(-np.linspace(0, 12, 77))[None, :] + 1e-30
        return radii, energies, em
    raise KeyError(kind)


@pytest.mark.parametrize("kind", ["primakoff", "agss09", "legacy", "ragged"])
def test_device_built_tables_are_bit_identical_to_the_host_path(kind):
    import torch
    radii, energies, em = _tables_from(kind)
    em = np.ascontiguousarray(em, dtype=np.float64)
    n_r, n_e = em.shape
    rcdf, ecdf = tables.build_cdfs(em, radii, energies)          # host: reference order of operations
    full = sa.initFullSetup(emission=em, n_radii=n_r, n_energies=n_e) if kind not in ("legacy",) else sa.initFullSetup(emission="legacy")
    with sa.RayTracer(full) as rt_host:                          # host path: sart_set_solar_tables
        h_rcdf, h_ecdf, h_rg, h_eg = rt_host.solar_tables(guides=True)
        rec_host = rt_host.traceAxionWrapper(20_000, seed=3)
    assert np.array_equal(h_rcdf, rcdf) and np.array_equal(h_ecdf, ecdf)   # sart_get_solar_tables returns what was set
    with sa.RayTracer(full) as rt_dev:
        d_em = torch.from_numpy(em).to("cuda:0")
        rt_dev.set_solar_tables_device(d_em.data_ptr(), radii, energies)
        d_rcdf, d_ecdf, d_rg, d_eg = rt_dev.solar_tables(guides=True)
        rec_dev = rt_dev.traceAxionWrapper(20_000, seed=3)
        img_d, s_d = rt_dev.trace_histogram(2_000_000, seed=4)
    assert np.array_equal(d_rcdf.view(np.uint64), rcdf.view(np.uint64)), "fluxRadiusCDF"
    assert np.array_equal(d_ecdf.view(np.uint64), ecdf.view(np.uint64)), "diffFluxCDFs"
    assert np.array_equal(d_rg, h_rg), "radius guide"
    assert np.array_equal(d_eg, h_eg), "energy guide"
    assert d_rcdf[-1] == 1.0 and np.all(d_ecdf[:, -1] == 1.0)
    assert rec_dev.tobytes() == rec_host.tobytes()               # radius_span and everything downstream identical
    with sa.RayTracer(full) as rt_host:
        img_h, s_h = rt_host.trace_histogram(2_000_000, seed=4)
    for k in ("N_PASSED", "N_SHELL_SELECTED", "N_REACHED_TELESCOPE", "N_HIT_NICKEL"):
        assert s_d[k] == s_h[k], k
    assert s_d["SUM_WEIGHTS"] == pytest.approx(s_h["SUM_WEIGHTS"], rel=1e-12)


def test_agss09_pipeline_stays_on_the_device():
    """BASELINE configs[4]'s front end: emission="agss09-device" runs emission kernel -> CDFs -> guides inside the context;
    nothing but the zones and the energy grid is uploaded.  Same records as the path through the host tables."""
    full_h = sa.initFullSetup(stage=L.SK_GAS, emission="agss09")
    full_d = sa.initFullSetup(stage=L.SK_GAS, emission="agss09-device")
    assert full_d.diffFluxCDFs is None and full_d.device_emission is not None
    with sa.RayTracer(full_h) as rt:
        rec_h = rt.traceAxionWrapper(30_000, seed=11)
        _, s_h = rt.trace_histogram(3_000_000, seed=12)
    with sa.RayTracer(full_d) as rt:
        rec_d = rt.traceAxionWrapper(30_000, seed=11)
        _, s_d = rt.trace_histogram(3_000_000, seed=12)
        full_d.fetch_solar_tables(rt)
    assert np.array_equal(full_d.fluxRadiusCDF, full_h.fluxRadiusCDF) and np.array_equal(full_d.diffFluxCDFs, full_h.diffFluxCDFs)
    assert rec_d.tobytes() == rec_h.tobytes()
    for k in ("N_PASSED", "N_SHELL_SELECTED", "N_REACHED_TELESCOPE"):
        assert s_d[k] == s_h[k]
    # the fetched tables feed the CPU oracle like any others
    from oracle.oracle import Oracle
    o_rec = Oracle(full_d).trace_records(30_000, seed=11)
    assert (o_rec["passed"] == rec_d["passed"]).mean() > 0.9995
    both = (o_rec["passed"] == 1) & (rec_d["passed"] == 1)
    assert np.array_equal(o_rec["energiesAx"][both], rec_d["energiesAx"][both])


def test_device_tables_reject_what_is_not_a_cdf():
    import torch
    radii, energies = tables.solar_grid(50, 40)
    full = sa.initFullSetup(n_radii=50, n_energies=40, refl_n_angles=20, refl_n_energies=20)
    with sa.RayTracer(full) as rt:
        em = np.ones((50, 40))
        em[7, :] = 0.0                                            # a row that sums to zero: 0 / 0
        d = torch.from_numpy(em).to("cuda:0")
        with pytest.raises(L.SartError) as e:
            rt.set_solar_tables_device(d.data_ptr(), radii, energies)
        assert e.value.code == L.SART_ERR_INVALID_ARGUMENT and "CDF" in str(e.value)
        with pytest.raises(L.SartError):                          # the context has no usable tables now
            rt.trace_histogram(1000)
        em = np.ones((50, 40))
        em[3, 5] = -1.0                                           # a negative rate: the running sum decreases
        d = torch.from_numpy(em).to("cuda:0")
        with pytest.raises(L.SartError):
            rt.set_solar_tables_device(d.data_ptr(), radii, energies)
        em[3, 5] = np.nan
        d = torch.from_numpy(em).to("cuda:0")
        with pytest.raises(L.SartError):
            rt.set_solar_tables_device(d.data_ptr(), radii, energies)
        d = torch.ones((50, 40), dtype=torch.float64, device="cuda:0")
        rt.set_solar_tables_device(d.data_ptr(), radii, energies)
        _, s = rt.trace_histogram(100_000)
        assert s["N_RAYS"] == 100_000 and s["N_PASSED"] > 0
