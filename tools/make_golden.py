#!/usr/bin/env python3
"""Generates tests/golden/*.npz with the CPU oracle (oracle/libsart_oracle.so, f64).

The reference (Nim) cannot be run in the build image, so these vectors pin *HIP path vs oracle* and guard
the oracle against regressions; oracle vs reference is pinned by tests/test_oracle_known_answers.py.
Inputs are the deterministic synthetic tables of solaraxionraytracing_amd.tables at the SMALL sizes of
tests/conftest.py.  Run from the repo root:  python tools/make_golden.py
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from tests.conftest import make_setup, SETUP_NAMES
from oracle.oracle import Oracle

N_REC, N_HIST, SEED = 4000, 100_000, 2024
FIELDS = ["passed", "passedTillWindow", "hitNickel", "pointdataX", "pointdataY", "pointdataR", "weights",
          "transmissionMagnet", "yawAngles", "reflect", "deviationDet", "pointdataXBefore", "pointdataYBefore",
          "energiesAx", "energiesPre", "emratesPre", "transProbWindow", "transProbArgon", "shellNumber", "kindsWindow"]

out_dir = os.path.join(ROOT, "tests", "golden")
os.makedirs(out_dir, exist_ok=True)
import solaraxionraytracing_amd as sa
# "<name>_full": the same setup on the default (BASELINE-size) tables: 1968 x 1500 emission CDFs, 1000 x 1000 reflectivity
for name in SETUP_NAMES + ["babyiaxo_xmm_full"]:
    full = sa.initFullSetup() if name == "babyiaxo_xmm_full" else make_setup(name)
    o = Oracle(full)
    rec = o.trace_records(N_REC, seed=SEED)
    img, summ, _ = o.trace_histogram(N_HIST, seed=SEED)
    data = {"rec_" + f: rec[f] for f in FIELDS}
    data["summary_keys"] = np.array(sorted(summ))
    data["summary_vals"] = np.array([summ[k] for k in sorted(summ)])
    # coarse image (32x32 blocks of the 256x256 image) - robust against single-pixel edge flips
    data["image_coarse"] = img.reshape(32, 8, 32, 8).sum(axis=(1, 3))
    data["inputs_checksum"] = np.array([full.fluxRadiusCDF.sum(), full.diffFluxCDFs.sum(), full.reflectivity.data.sum(),
                                        full.detector_tables.window.sum()])
    data["meta"] = np.array([N_REC, N_HIST, SEED, full.flags])
    np.savez_compressed(os.path.join(out_dir, name + ".npz"), **data)
    print(name, "passed", int(rec["passed"].sum()), "/", N_REC, "flux", summ["SUM_WEIGHTS"])

# ---- emission-table producer (include/sart_emission.h): a 50 x 30 sub-grid of the eight terms from the C oracle ----
import solaraxionraytracing_amd.emission as em
from oracle import oracle as O
from solaraxionraytracing_amd import tables
zones = em.solar_zones()
_, energies = tables.solar_grid()
R_STRIDE, E_STRIDE = 40, 50
total, comp = O.emission_table(zones, energies, em.default_params(), components=True, r_stride=R_STRIDE, e_stride=E_STRIDE)
np.savez_compressed(os.path.join(out_dir, "emission_agss09.npz"), r_stride=R_STRIDE, e_stride=E_STRIDE,
                    total=total[::R_STRIDE, ::E_STRIDE], components=comp[:, ::R_STRIDE, ::E_STRIDE],
                    zone_n_e=np.array([z.n_e for z in zones])[::R_STRIDE], zone_temp_index=np.array([z.temp_index for z in zones])[::R_STRIDE])
print("emission_agss09", total[::R_STRIDE, ::E_STRIDE].shape, "max", np.nanmax(total))
