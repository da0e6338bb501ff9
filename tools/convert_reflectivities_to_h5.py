#!/usr/bin/env python3
"""tools/convert_reflectivities_to_h5.nim of the reference: the gold reflectivity files downloaded from
henke.lbl.gov (`resources/henke_download/<angle>degGold0.25microns.csv`, one per grazing angle) -> ONE H5 file in the schema
initReflectivity reads (raytracer.nim:1196-1209): what `goldReflFile` of config.toml names.

  python tools/convert_reflectivities_to_h5.py [--indir ../resources/henke_download] [--out ../resources/gold_0.25microns_reflectivities.h5]

(The reference's script also draws a raster plot of the grid; plots are out of scope here.)"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--indir", default="../resources/henke_download/")
    ap.add_argument("--out", default="../resources/gold_0.25microns_reflectivities.h5")
    ap.add_argument("--pattern", default="*degGold0.25microns.csv")
    args = ap.parse_args()
    from solaraxionraytracing_amd import tables
    g = tables.convert_henke_directory_to_h5(args.indir, args.out, args.pattern)
    print("wrote %s: %d angles %.6g .. %.6g deg x %d energies %.3g .. %.3g keV" % (args.out, g.data.shape[1], g.angle_min, g.angle_max,
                                                                                  g.data.shape[2], g.energy_min, g.energy_max))


if __name__ == "__main__":
    main()
