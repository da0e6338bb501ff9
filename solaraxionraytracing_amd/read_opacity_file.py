"""Command line of the reference's `readOpacityFile` binary (`proc main`, readOpacityFile.nim:983-1005) on top of the GPU path:
the solar-model pre-processor that writes the emission table `raytracer` samples from.

    python -m solaraxionraytracing_amd.read_opacity_file [--config FILE | --configPath DIR] [--device N]

Reads `[Resources].resourcePath / rawSolarModel` (the AGSS09 table), `[ReadOpacityFile].opcdPath` (the OPCD 3.3 files) and
writes `[Resources].outputPath / [ReadOpacityFile].solarModelFile` = `solar_model_dataframe.csv` (columns Radius,
Energy [keV], emRates; :853-854, :1003) - the file `initFullSetup` reads (raytracer.nim:2647-2668).  `calculateOpacities`
(:598-880) runs as: zones and number densities on the host (first loop, :655-705), OPCD files parsed by the C++ reader,
absorption coefficients (:790-823) and the eight emission terms (:826-851) in two HIP kernels.  What differs from the
reference: no plots - the numbers behind `diffFlux.pdf` (getFluxFractionR of the total and of every term, :914-922) are
written as `diff_flux.csv`; without OPCD files the run continues with absCoef = 0 (the reference raises) and says so."""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np

from . import _lib, config as cfgmod, emission, opacity, tables

FLUX_COLUMNS = ("Total flux", "FB BB Flux", "Compton Flux", "EE Flux", "FF Flux", "Primakoff Flux", "LP Flux", "TP Flux",
                "57Fe Flux")                                          # the `type` labels of :914-922
_COMPONENT_OF = (None, 1, 0, 2, 3, 4, 5, 6, 7)                        # column -> plane of _lib.EM_TERMS (None = the sum)


def calculate_opacities(solar_model: str, opcd_path: str | None, device: int = 0, n_energies: int = tables.N_ENERGIES):
    """calculateOpacities (:598-880).  Returns (radii, energies, emRates[nR][nE], components[8][nR][nE], absCoefs or None, notes)."""
    try:
        profile = tables.read_solar_model(solar_model)
    except OSError as e:
        raise IOError("Failed to read solar model at path: %s" % solar_model) from e      # :601-602
    n_r = profile["rho"].size
    zones = emission.solar_zones(profile=profile)
    radii = np.array([z.radius_frac for z in zones])
    energies = np.linspace(1e-3, 15.0, n_energies)                                          # :608-609
    notes, absc = [], None
    if opcd_path and os.path.exists(os.path.join(opacity.mono_dir(opcd_path), "fm01.mesh")):
        n_z = opacity.number_densities(profile)
        with opacity.OpcdSet(opcd_path, zones) as s:
            absc = opacity.abs_coefs(zones, n_z, energies, s, device=device)
            notes.append("OPCD: %d (temperature, density) slots, %d opacity values" % (s.n_slots, s.tables.contents.n_table_y))
    else:
        notes.append("no OPCD files under %r: absorption coefficients set to 0 (term1 and the plasmon terms lose their opacity part)"
                     % (opcd_path,))
    total, comp = emission.emission_table(zones, energies, abs_coefs=absc, components=True, device=device)
    assert total.shape == (n_r, n_energies)
    return radii, energies, total, comp, absc, notes


def write_diff_flux_csv(path: str, energies, total, comp) -> None:
    """getFluxFractionR (:535-584) of the total and of the eight terms: what `diffFlux.pdf` shows (:914-922, :958-971)."""
    cols = [emission.flux_spectrum(total if k is None else comp[k], energies) for k in _COMPONENT_OF]
    with open(path, "w") as f:
        f.write("Energy," + ",".join(FLUX_COLUMNS) + "\n")
        for i, e in enumerate(energies):
            f.write(repr(float(e)) + "," + ",".join(repr(float(c[i])) for c in cols) + "\n")


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog="python -m solaraxionraytracing_amd.read_opacity_file", description=__doc__,
                                 formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--config", default="", help="path of a config.toml")
    ap.add_argument("--configPath", default="", help="directory that holds config.toml")
    ap.add_argument("--device", type=int, default=0)
    args = ap.parse_args(argv)
    path = args.config or os.path.join(args.configPath or "config", "config.toml")
    if not os.path.exists(path):
        default = os.path.join(os.path.dirname(path), "config_default.toml")
        if args.config or not os.path.exists(default):
            raise IOError("no config file %s" % path)
        with open(default) as src, open(path, "w") as dst:     # :993-996: config.toml is created from config_default.toml
            dst.write(src.read())
    cfg = cfgmod.load_config(path)
    base = os.path.dirname(os.path.abspath(path))
    res, rof = cfg["Resources"], cfg["ReadOpacityFile"]
    resources = os.path.normpath(os.path.join(base, res["resourcePath"]))
    outpath = os.path.normpath(os.path.join(base, res["outputPath"]))
    solar_model = os.path.join(resources, res["rawSolarModel"])
    opcd = cfgmod.resolve_opcd_path(cfg, base)
    os.makedirs(outpath, exist_ok=True)
    print("Walking all radii")
    radii, energies, total, comp, absc, notes = calculate_opacities(solar_model, opcd, device=args.device)
    for n in notes:
        print(n)
    out = os.path.join(outpath, rof["solarModelFile"])
    tables.write_solar_model_csv(out, radii, energies, total)
    write_diff_flux_csv(os.path.join(outpath, "diff_flux.csv"), energies, total, comp)
    iron = emission.flux_spectrum(comp[7], energies)
    g_an = emission.default_params().g_anuclei
    print("57Fe Flux ", float(iron.sum()) * 0.001 / g_an / g_an * 3.171e-12, " g_aN² cm⁻2 s⁻1")        # :927
    print("wrote", out)
    return 0


if __name__ == "__main__":
    sys.exit(main())
