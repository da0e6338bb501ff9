import sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd())
import solaraxionraytracing_amd as sa
from solaraxionraytracing_amd import _lib as L
from solaraxionraytracing_amd.raytracer import accumulator_len
full = sa.initFullSetup()
n = 1_000_000_000
names = ["N_RAYS","N_REACHED","N_SHELL","N_NICKEL","N_TILL","N_PASSED","SUM_W","SUM_X","SUM_Y","SUM_R","SUM_W2","res11","W_HI","X_HI","Y_HI","R_HI","W2_HI","W_OUT","W_OUT_HI"]
with sa.RayTracer(full) as rt:
    rt.set_accumulation_mode("fixed64", int(os.environ.get("HEADROOM", "0")))
    acc = torch.zeros(accumulator_len(256), dtype=torch.int64, device="cuda:0")
    out = torch.zeros(accumulator_len(256), dtype=torch.float64, device="cuda:0")
    for k in range(int(os.environ.get("STEPS", "2600"))):
        p = rt.trace_params(n, seed=1, ray_id_offset=k * n, accumulate=(k > 0))
        rt.trace_histogram_device(p, acc.data_ptr())
        if (k + 1) % 200 == 0:
            rt.finalize_accumulator_device(p, acc.data_ptr(), out.data_ptr())
            try:
                rt.synchronize(); ok = "ok"
            except Exception as e:
                ok = "FAIL: " + str(e)[:160]
            a = acc.cpu().numpy(); img = a[:65536]; sc = a[65536:65536 + 24]
            print(k + 1, ok, "max pixel 2^%.2f" % np.log2(float(img.max())), "min pixel", int(img.min()), "scalars", {names[i]: int(sc[i]) for i in (0, 5, 6, 10, 12, 16, 17) if i < len(names)}, flush=True)
