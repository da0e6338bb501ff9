import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # Keep the native pieces in step with their sources: `make` is a no-op when they are up to date
    # (hipcc cross-compiles without a GPU; the same toolchain exists on the GPU box).
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "solaraxionraytracing_amd", "csrc")], check=True)
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


# ---- shared setups (small tables so that the oracle finishes in seconds) ----------------------
SMALL = dict(n_radii=400, n_energies=300, refl_n_angles=200, refl_n_energies=200)


def make_setup(name, **kw):
    import solaraxionraytracing_amd as sa
    from solaraxionraytracing_amd import _lib as L
    args = dict(SMALL)
    args.update(kw)
    if name == "babyiaxo_xmm":
        return sa.initFullSetup(**args)
    if name == "babyiaxo_xmm_gas":
        return sa.initFullSetup(stage=L.SK_GAS, **args)
    if name == "cast_llnl":
        return sa.initFullSetup(L.ES_CAST, L.DK_INGRID2018, L.SK_VACUUM, L.TK_LLNL, **args)
    if name == "cast_llnl_gold":
        return sa.initFullSetup(L.ES_CAST, L.DK_INGRID2018, L.SK_VACUUM, L.TK_LLNL, reflectivity="gold", **args)
    if name == "cast_abrixas":
        return sa.initFullSetup(L.ES_CAST, L.DK_INGRID2017, L.SK_VACUUM, L.TK_ABRIXAS, **args)
    if name == "babyiaxo_xmm_xray":
        return sa.initFullSetup(flags=L.CF_XRAY_TEST, **args)
    if name == "babyiaxo_xmm_rot":
        full = sa.initFullSetup(**args)
        full.setup.telescope_turned_x_deg = 0.02
        full.setup.telescope_turned_y_deg = 0.05
        full.setup.chip_x_max = full.setup.chip_y_max = 100.0
        full.flags = L.CF_IGNORE_DET_WINDOW | L.CF_IGNORE_GAS_ABS | L.CF_IGNORE_CONV_PROB
        return full
    raise KeyError(name)


SETUP_NAMES = ["babyiaxo_xmm", "babyiaxo_xmm_gas", "cast_llnl", "cast_llnl_gold", "cast_abrixas", "babyiaxo_xmm_xray",
               "babyiaxo_xmm_rot"]
