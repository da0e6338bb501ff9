"""Every number the reference ships that bears on the hot path and is not yet used elsewhere.

None of these can make parity "green" — the reference has no runnable fixtures — but each shrinks the unpinned surface:

 * `emission_rates_Hz.txt` + `energies.txt` (input E2): the one non-synthetic solar table.  It is
   the legacy C++ code's axion-electron Compton emission rate; its spectral shape per radius pins the Compton plane of the
   emission producer (readOpacityFile.nim:360-362) and the temperature profile behind it; as a 397 x 233 input it is traced
   ray for ray against the binary128 oracle.
 * `McXtrace_angular_xmm.csv` / `xmm_newton_angular_effective_area.csv` (raytracer.nim:2805-2813): the angular scan of the
   HIP path lies between / next to the two curves the reference overlays, and on SURVEY App. C's independent numbers.
 * `llnl_xray_telescope_cast_effective_area_parallel_light_DTU_thesis.csv` (config_default.toml:16): CAST / LLNL effective
   area for parallel light from `--xrayTest` rays, with the coating caveat spelled out.
"""
import numpy as np
import pytest

import solaraxionraytracing_amd as sa
from solaraxionraytracing_amd import _lib as L, tables


# ---------------------------------------------------------------------------------------------------------------------
# E2: emission_rates_Hz.txt / energies.txt
# ---------------------------------------------------------------------------------------------------------------------
def test_legacy_emission_table_is_the_reference_file():
    radii, energies, em = tables.legacy_emission_table()
    assert em.shape == (397, 233) and energies.shape == (233,) and radii.shape == (397,)
    # first / last values of the two files as shipped (emission_rates_Hz.txt:1, :92501; energies.txt:1, :233)
    assert em[0, 0] == 2.0996e-11 and em[-1, -1] == 1.0979e-15
    assert energies[0] == 0.360496 and energies[-1] == 11.3847
    assert radii[0] == 0.0015 and radii[-1] == pytest.approx(0.1995)
    assert np.all(np.diff(energies) > 0) and np.all(em > 0)


def test_legacy_emission_as_tracer_input_through_the_oracle():
    """E2 through initFullSetup's CDF construction (raytracer.nim:2670-2705) and the CPU oracle: plumbing of the one real
    solar input (BASELINE configs[0] style: no GPU)."""
    from oracle.oracle import Oracle
    full = sa.initFullSetup(emission="legacy", refl_n_angles=200, refl_n_energies=200)
    assert full.diffFluxCDFs.shape == (397, 233) and full.fluxRadiusCDF[-1] == 1.0
    assert np.all(np.diff(full.fluxRadiusCDF) >= 0) and np.all(full.diffFluxCDFs[:, -1] == 1.0)
    _, s, _ = Oracle(full).trace_histogram(200_000, seed=3)
    # SURVEY App. C (scratch restatement on this very table): ~24 % of the rays reach the histogram, 33 % the mirrors
    assert s["N_PASSED"] / s["N_RAYS"] == pytest.approx(0.24, abs=0.02)
    assert s["N_SHELL_SELECTED"] / s["N_RAYS"] == pytest.approx(0.329, abs=0.01)


def _compton_plane(energies, use_gpu):
    import solaraxionraytracing_amd.emission as em
    zones = em.solar_zones()
    if use_gpu:
        return em.emission_table(zones, energies, params=em.default_params(L.EM_ALL), components=True)[1][0][:397]
    from oracle import oracle as O
    _, comp = O.emission_table(zones, energies, em.default_params(), components=True)
    return comp[0][:397]


def _check_compton_shape(compton, em_legacy):
    # per-radius spectral shape: E^2 / (exp(E / T(r)) - 1) in both codes -> identical up to the legacy file's six digits
    a = em_legacy / em_legacy.sum(axis=1, keepdims=True)
    b = compton / compton.sum(axis=1, keepdims=True)
    assert np.abs(a / b - 1.0).max() < 3e-3
    # radial profile: both are proportional to n_e(r); the legacy code builds n_e differently (fewer species), the ratio drifts by 16 % over the 397 radii
    k = (em_legacy / compton).mean(axis=1)
    k = k / k[0]
    assert np.all((k > 0.8) & (k < 1.001)) and 0.8 < k[-1] < 0.9


def test_legacy_compton_rates_pin_the_emission_oracles_compton_plane():
    """emission_rates_Hz.txt = K(r) x comptonEmrate(r, E) of readOpacityFile.nim:360-362 (checked here against the emission
    oracle; the GPU producer is checked against the oracle in tests/test_emission.py and against the file below)."""
    _, energies, em_legacy = tables.legacy_emission_table()
    _check_compton_shape(_compton_plane(energies, use_gpu=False), em_legacy)


@pytest.mark.gpu
def test_legacy_compton_rates_pin_the_emission_kernels_compton_plane():
    _, energies, em_legacy = tables.legacy_emission_table()
    assert L.EM_TERMS[0].lower().startswith("compton")
    _check_compton_shape(_compton_plane(energies, use_gpu=True), em_legacy)


@pytest.mark.gpu
def test_legacy_emission_traced_ray_for_ray_against_binary128_oracle():
    """E2 (397 x 233, non-uniform energy grid) as the solar input of the HIP path."""
    from oracle.oracle import Oracle
    full = sa.initFullSetup(emission="legacy")
    n = 30_000
    with sa.RayTracer(full) as rt:
        rec = rt.traceAxionWrapper(n, seed=77)
        img, s = rt.trace_histogram(2_000_000, seed=77)
    ref = Oracle(full, "q").trace_records(n, seed=77)
    for f in ("passed", "passedTillWindow", "hitNickel", "shellNumber", "kindsWindow"):
        np.testing.assert_array_equal(rec[f], ref[f])
    np.testing.assert_array_equal(rec["energiesPre"], ref["energiesPre"])
    assert set(np.unique(rec["energiesPre"])) <= set(np.maximum(full.energies, 0.03))
    m = rec["passedTillWindow"] == 1
    assert np.abs(rec["pointdataX"] - ref["pointdataX"]).max() < 1e-10
    np.testing.assert_allclose(rec["weights"][m], ref["weights"][m], rtol=2e-8)
    assert s["N_PASSED"] / s["N_RAYS"] == pytest.approx(0.24, abs=0.01)     # SURVEY App. C on this table


# ---------------------------------------------------------------------------------------------------------------------
# angular scan vs the curves the reference overlays
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_angular_scan_against_mcxtrace_and_xmm_curves():
    """performAngularScan (raytracer.nim:2778-2802) on McXtrace's own angle grid, XMM shells, chip 100 mm, effective-area
    flags.  The reference only overlays the two curves (:2805-2813) and makes no claim of its own, so
    the bound is: the HIP curve stays within 0.07 of the band spanned by the McXtrace simulation and the XMM-Newton
    vignetting curve (it runs up to 0.066 above both around 0.1 deg and between them from 0.2 deg on), and it reproduces the
    numbers SURVEY App. C got from an independent vectorised restatement (different emission table: E2) to 0.015."""
    ref = tables.reference_curves()
    ang = ref["mcxtrace_angle_deg"]
    assert ang[0] == 0.0 and ang[-1] == 0.3 and len(ang) == 14
    full = sa.initFullSetup()
    full.setup.chip_x_max = full.setup.chip_y_max = 100.0
    flags = L.CF_IGNORE_DET_WINDOW | L.CF_IGNORE_GAS_ABS | L.CF_IGNORE_CONV_PROB
    with sa.RayTracer(full) as rt:
        _, flux, rel = sa.performAngularScan(rt, 0, 0, 1, 2_000_000, flags=flags, angles=ang)
    ok = ref["xmm_effective_area"] > 0
    xmm = np.interp(ang, ref["xmm_angle_arcmin"][ok] / 60.0, ref["xmm_effective_area"][ok] / ref["xmm_effective_area"].max())
    mcx = ref["mcxtrace_rel_flux"]
    lo, hi = np.minimum(mcx, xmm), np.maximum(mcx, xmm)
    assert rel[0] == 1.0 and np.all(np.diff(rel) < 0)
    assert np.all(rel >= lo - 0.07) and np.all(rel <= hi + 0.07), np.c_[ang, rel, mcx, xmm]
    far = ang >= 0.2
    assert np.all(rel[far] >= lo[far] - 0.005) and np.all(rel[far] <= hi[far] + 0.01)      # between the two from 0.2 deg on
    assert np.sqrt(np.mean((rel - 0.5 * (mcx + xmm)) ** 2)) < 0.04
    survey_c = {0.05: 0.948, 0.1: 0.856, 0.2: 0.633, 0.3: 0.454}
    for a, want in survey_c.items():
        assert rel[np.argmin(np.abs(ang - a))] == pytest.approx(want, abs=0.015), a


# ---------------------------------------------------------------------------------------------------------------------
# CAST / LLNL effective area for parallel light
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_llnl_parallel_beam_effective_area_against_dtu_thesis_curve():
    """`--xrayTest` parallel beam (raytracer.nim:1765-1806) filling the CAST bore (r = 21.5 mm, 14.5 cm^2), window / gas /
    conversion factors off: pi r^2 sum(w) / N is the telescope's effective area for parallel light, the quantity of
    llnl_xray_telescope_cast_effective_area_parallel_light_DTU_thesis.csv.

    Coating caveat: the real optic carries Pt/C multilayers (the reference's llnl_layer_reflectivities.h5, not shipped);
    the only reflectivity data in the repository are the Henke scans of 0.25 um gold.  With gold on all 14 shells the curve is
    reproduced to 12 % up to 2 keV and at 4-5 keV — geometry (92 % geometric throughput of the bore) and the two-bounce
    reflectivity level —, sits a third lower at 3 keV (gold M edges, 2.2-3.4 keV) and falls off faster above 6 keV, where
    the multilayers are what carries the real telescope."""
    e_ref, a_ref = tables.llnl_effective_area()
    area_cm2 = np.pi * 2.15 ** 2
    flags = L.CF_XRAY_TEST | L.CF_IGNORE_DET_WINDOW | L.CF_IGNORE_GAS_ABS | L.CF_IGNORE_CONV_PROB
    got = {}
    for energy in (0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 5.0, 7.0, 9.0):
        src = L.TestSourceConfig()
        src.active, src.parallel = 1, 1
        src.energy, src.distance, src.radius, src.activity = energy, 100.0, 21.5, 1.0
        src.offAxisUp = src.offAxisLeft = src.lengthCol = 0.0
        full = sa.initFullSetup(L.ES_CAST, L.DK_INGRID2018, L.SK_VACUUM, L.TK_LLNL, flags=flags, source_cfg=src, reflectivity="gold")
        with sa.RayTracer(full) as rt:
            _, s = rt.trace_histogram(1_000_000, seed=4, flags=flags)
        got[energy] = area_cm2 * s["SUM_WEIGHTS"] / s["N_RAYS"]
        assert s["N_PASSED"] / s["N_RAYS"] == pytest.approx(0.921, abs=0.005)       # geometric throughput of the bore
    ratio = {e: got[e] / np.interp(e, e_ref, a_ref) for e in got}
    for e in (0.5, 1.0, 1.5, 2.0, 4.0, 5.0):
        assert abs(ratio[e] - 1.0) < 0.12, (e, got[e], ratio[e])
    assert 0.55 < ratio[3.0] < 0.8, ratio
    assert ratio[7.0] < 0.7 and ratio[9.0] < 0.4, ratio                         # gold alone: below the multilayer optic
    vals = [got[e] for e in (2.0, 5.0, 7.0, 9.0)]
    assert all(a > b for a, b in zip(vals, vals[1:]))                           # falls with energy like the thesis curve


# ---------------------------------------------------------------------------------------------------------------------
# axionMass/: the two numbers the reference's own notes assert for the effective photon mass, and the NIST table its
# helium attenuation fit was made from
# ---------------------------------------------------------------------------------------------------------------------
# axionMass.org:775-806: pressures equivalent to 1 bar / 3 bar at 293 K in a 4.2 K magnet (p * 4.2 / 293), and
# m_gamma = babyIaxoEffMass(p) (10 m x 0.3 m bore, 4.2 K); table "biljana_ref_values": 14.3345 mbar -> 0.26048 eV,
# 43.0034 mbar -> 0.45117 eV; the notes `doAssert` 0.26 and 0.4483 (eps 1e-2) from the IAXO gas-phase study.
M_GAMMA_TABLE = ((1000.0 * 4.2 / 293.0, 14.3345, 0.26048), (3000.0 * 4.2 / 293.0, 43.0034, 0.45117))
# axionMass/mass_attenuation_nist_data.txt rows 1-10 keV (NIST XCOM, helium): energy [keV], mu/rho [cm^2/g]
NIST_HELIUM = ((1.0, 6.084e+01), (1.5, 1.676e+01), (2.0, 6.863e+00), (3.0, 2.007e+00), (4.0, 9.329e-01), (5.0, 5.766e-01),
               (6.0, 4.195e-01), (8.0, 2.933e-01), (10.0, 2.476e-01))
NIST_FIT_RESIDUAL = 0.025   # |fit / table - 1| of logMassAttenuation (axionMassforMagnet.nim:70-73) over 1-10 keV: max 2.1 % at 1.5 keV


def test_oracle_effective_photon_mass_against_the_reference_notes():
    from oracle.oracle import load
    lib = load()
    for p, p_printed, m_gamma in M_GAMMA_TABLE:
        assert p == pytest.approx(p_printed, abs=5e-5)                      # the table's own pressure column
        got = lib.sart_oracle_eff_photon_mass2(p, 10.0, 0.3, 4.2)           # effPhotonMass2 = babyIaxoEffMass (:51-61 / org :189-202)
        assert got == pytest.approx(m_gamma, abs=5e-6), (p, got)            # the table prints five digits
        # the volume cancels (amountMol / vol): any bore gives the same mass
        assert lib.sart_oracle_eff_photon_mass2(p, 11.0, 0.5, 4.2) == pytest.approx(got, rel=1e-14)
    # the notes' own assertions (IAXO gas-phase study: 0.26 eV and 0.4483 eV within 1e-2)
    assert lib.sart_oracle_eff_photon_mass2(M_GAMMA_TABLE[0][0], 10.0, 0.3, 4.2) == pytest.approx(0.26, abs=1e-3)
    assert lib.sart_oracle_eff_photon_mass2(M_GAMMA_TABLE[1][0], 10.0, 0.3, 4.2) == pytest.approx(0.4483, abs=1e-2)


def test_oracle_helium_mass_attenuation_against_the_nist_table():
    from oracle.oracle import load
    lib = load()
    worst = 0.0
    for e, mu in NIST_HELIUM:
        fit = lib.sart_oracle_mass_attenuation(e)
        worst = max(worst, abs(fit / mu - 1.0))
        assert fit == pytest.approx(mu, rel=NIST_FIT_RESIDUAL), (e, fit, mu)
    assert 0.015 < worst < NIST_FIT_RESIDUAL      # it IS a fit: a residual of zero would mean the table was copied, not fitted
    # intensitySuppression2 (:100-113) is exp(-mu/rho * rho * d): one metre of helium at 1 mbar / 293.15 K at 4 keV
    rho = 1e2 * 4.002602 / (8.314 * 293.15 * 1000.0) / 1000.0               # density(), :4-15, g / cm^3
    want = np.exp(-lib.sart_oracle_mass_attenuation(4.0) * rho * 100.0)
    assert lib.sart_oracle_intensity_suppression2(4.0, 0.0, 1.0, 1.0, 293.15, 293.15) == pytest.approx(want, rel=1e-14)


@pytest.mark.gpu
def test_gas_stage_resonance_sits_at_the_reference_notes_photon_mass():
    """Product-level: the hoisted m_gamma of the HIP path (sart_api.hip: hoist_setup) through the physics it drives.  In the
    gas stage the conversion probability peaks where the momentum transfer vanishes, m_a = m_gamma; with the magnet at the
    notes' 14.3345 mbar / 4.2 K the mass scan must peak at 0.26048 eV."""
    full = sa.initFullSetup(stage=L.SK_GAS, n_radii=400, n_energies=300, refl_n_angles=200, refl_n_energies=200)
    s = full.setup
    s.room_temp, s.magnet_tGas = 293.0, 4.2
    s.magnet_pGasRoom = 1000.0          # pGas = pGasRoom / roomTemp * tGas = 14.3345, handed on as mbar (raytracer.nim:1601, sic)
    masses = np.linspace(0.2565, 0.2645, 161)
    with sa.RayTracer(full) as rt:
        flux = sa.performAxionMassScan(rt, masses, 200_000, flags=L.CF_IGNORE_DET_WINDOW)
    assert flux.max() > 10 * max(flux[0], flux[-1])                         # a resonance, not a slope
    top = flux > 0.5 * flux.max()
    peak = float((masses[top] * flux[top]).sum() / flux[top].sum())
    assert peak == pytest.approx(0.26048, abs=1.5e-4), peak


@pytest.mark.gpu
def test_gas_absorption_follows_the_nist_helium_table():
    """Product-level: mu(E) of the HIP path's per-energy table (sart_api.hip: hoist_energy_tables) read back from records.
    With the conversion probability ignored transmissionMagnet = cos(yaw) * exp(-mu/rho(E) * (rho_pipe d_pipe + rho_magnet L));
    within one shell the geometry is the same for every ray, so -ln(absorption) at two energies is in the ratio of mu/rho."""
    full = sa.initFullSetup(stage=L.SK_GAS, n_radii=400, n_energies=300, refl_n_angles=200, refl_n_energies=200)
    with sa.RayTracer(full) as rt:
        rec = rt.traceAxionWrapper(400_000, seed=21, flags=L.CF_IGNORE_CONV_PROB)
    ok = rec["passed"] == 1
    tau = -np.log(rec["transmissionMagnet"][ok] / np.cos(rec["yawAngles"][ok]))   # cos of a degree value taken as radians - sic (:1598)
    e, shell = rec["energiesAx"][ok], rec["shellNumber"][ok]
    assert tau.min() > 0 and ok.sum() > 50_000
    nist_e = np.array([x for x, _ in NIST_HELIUM])
    nist_mu = np.array([y for _, y in NIST_HELIUM])
    checked = 0
    for sh in np.unique(shell)[::7]:
        in_shell = shell == sh
        # per energy of the grid: the mean optical depth of this shell's rays (they differ by the slope factor, < 1e-5)
        energies = np.unique(e[in_shell])
        energies = energies[(energies >= 1.0) & (energies <= 10.0)]
        ref_e = energies[np.argmin(np.abs(energies - 4.0))]
        tau_ref = tau[in_shell & (e == ref_e)].mean()
        for en in energies[::12]:
            got = tau[in_shell & (e == en)].mean() / tau_ref
            want = np.exp(np.interp(np.log(en), np.log(nist_e), np.log(nist_mu)) - np.interp(np.log(ref_e), np.log(nist_e), np.log(nist_mu)))
            assert got == pytest.approx(want, rel=2.2 * NIST_FIT_RESIDUAL), (sh, en, got, want)   # two fit residuals + the log-log interpolation
            checked += 1
    assert checked > 30
