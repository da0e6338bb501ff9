import sys, os, ctypes, numpy as np
sys.path.insert(0, os.getcwd())
import solaraxionraytracing_amd as sa
from solaraxionraytracing_amd import _lib as L
lib = ctypes.CDLL(os.environ["SART_LIBSART"])
names = {0: "B passes", 1: "asin fallback", 2: "atan fallback", 3: "cos fallback", 4: "yaw-slope fallback", 5: "normal facing the ray", 6: "energy wide/tie",
         7: "radius wide", 8: "normal_z_general (miss)", 9: "nickel lz<=0", 10: "ACC", 11: "A1 passes"}
def run(name, full):
    buf = (ctypes.c_ulonglong * 32)()
    with sa.RayTracer(full) as rt:
        rt.trace_histogram(1_000_000, seed=1)
        lib.sart_internal_rare_counts(None, 1)
        n = 100_000_000
        rt.trace_histogram(n, seed=1)
        lib.sart_internal_rare_counts(buf, 0)
    c = list(buf)
    print(name, "per 64 launched rays: A1 passes %.4f, B passes %.4f" % (c[11] / (n / 64), c[0] / (n / 64)))
    for k in sorted(names):
        if k in (0, 11): continue
        base = c[11] if k == 7 else c[0]
        print("   %-26s %10d  = %.3f per %s pass, %.2f lanes each" % (names[k], c[k], c[k] / max(base, 1), "A1" if k == 7 else "B", c[16 + k] / max(c[k], 1)))
run("BabyIAXO", sa.initFullSetup())
run("CAST", sa.initFullSetup(L.ES_CAST, L.DK_INGRID2018, L.SK_VACUUM, L.TK_LLNL, reflectivity="gold"))
run("gas", sa.initFullSetup(stage=L.SK_GAS))
