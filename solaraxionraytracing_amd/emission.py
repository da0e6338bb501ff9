"""Solar emission-table producer (include/sart_emission.h): the cell loop of ``calculateOpacities``
(readOpacityFile.nim:745-860) on the GPU, plus its host-side helpers.  ctypes plumbing only."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib
from .tables import DATA_DIR, solar_grid


def solar_zones(n_radii: int | None = None, profile: dict | None = None):
    """Per-radius plasma quantities (first loop of calculateOpacities, :655-705) from the AGSS09 columns shipped in
    data/solar_profile.npz (or ``profile`` = {"temp_K", "rho", "mass_fractions"[n][29]}).  Returns a ctypes array of
    ``SolarZone``."""
    host = _lib.load_host()
    if profile is None:
        profile = np.load(os.path.join(DATA_DIR, "solar_profile.npz"))
    temp = np.ascontiguousarray(profile["temp_K"], dtype=np.float64)
    rho = np.ascontiguousarray(profile["rho"], dtype=np.float64)
    frac = np.ascontiguousarray(profile["mass_fractions"], dtype=np.float64)
    if n_radii is None:
        n_radii = temp.shape[0]
    if n_radii > temp.shape[0] or frac.shape != (temp.shape[0], 29):
        raise ValueError("solar profile: need %d radii and 29 mass fractions per radius" % n_radii)
    zones = (_lib.SolarZone * n_radii)()
    _lib.check(host.sart_host_solar_zones(_lib.as_dp(temp), _lib.as_dp(rho), _lib.as_dp(frac), n_radii, zones), host=True)
    return zones


def default_params(terms: int = _lib.EM_ALL) -> _lib.EmissionParams:
    p = _lib.EmissionParams()
    _lib.load_sart().sart_emission_default_params(C.byref(p))
    p.terms = terms
    return p


def emission_table(zones, energies, abs_coefs=None, params: _lib.EmissionParams | None = None, components: bool = False,
                   device: int = 0):
    """emRates[n_radii][n_energies] (and, with ``components``, the eight single terms [8][nR][nE] in the order of
    ``_lib.EM_TERMS``) from the HIP kernel.  Raises without a GPU: there is no CPU path."""
    lib = _lib.load_sart()
    energies = np.ascontiguousarray(energies, dtype=np.float64)
    n_r, n_e = len(zones), energies.size
    if params is None:
        params = default_params()
    if abs_coefs is not None:
        abs_coefs = np.ascontiguousarray(abs_coefs, dtype=np.float64)
        if abs_coefs.shape != (n_r, n_e):
            raise ValueError("abs_coefs must be [n_radii][n_energies]")
    out = np.empty((n_r, n_e))
    comp = np.empty((len(_lib.EM_TERMS), n_r, n_e)) if components else None
    ctx = C.c_void_p()
    _lib.check(lib.sart_create(device, C.byref(ctx)))
    try:
        _lib.check(lib.sart_emission_table(ctx, zones, n_r, _lib.as_dp(energies), n_e,
                                           _lib.as_dp(abs_coefs) if abs_coefs is not None else None, C.byref(params),
                                           _lib.as_dp(out), _lib.as_dp(comp) if components else None))
    finally:
        lib.sart_destroy(ctx)
    return (out, comp) if components else out


def last_kernel_ms() -> float:
    return float(_lib.load_sart().sart_emission_last_kernel_ms())


def flux_spectrum(em_rates, energies) -> np.ndarray:
    """getFluxFractionR (readOpacityFile.nim:535-584): flux at Earth in 1/(keV y m^2) per energy."""
    host = _lib.load_host()
    em = np.ascontiguousarray(em_rates, dtype=np.float64)
    energies = np.ascontiguousarray(energies, dtype=np.float64)
    out = np.empty(energies.size)
    _lib.check(host.sart_host_flux_spectrum(_lib.as_dp(em), em.shape[0], _lib.as_dp(energies), energies.size, _lib.as_dp(out)),
               host=True)
    return out


def agss09_emission_table(n_radii: int | None = None, n_energies: int | None = None, terms: int = _lib.EM_ALL, device: int = 0):
    """The reference's emission table for the AGSS09 model on the reference's grid (energies linspace(1e-3, 15, 1500),
    :612-613) without the OPCD absorption coefficients (absCoef = 0): returns (radii, energies, emRates)."""
    zones = solar_zones(n_radii)
    radii, energies = solar_grid(len(zones), n_energies) if n_energies else solar_grid(len(zones))
    return radii, energies, emission_table(zones, energies, params=default_params(terms), device=device)
