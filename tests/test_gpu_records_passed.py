"""sart_trace_records_passed: the passed rays of traceAxionWrapper's buffer, compacted on the device in ray order, and the counts
generateResultPlots echoes (raytracer.nim:2252-2257; the scan sum :2800 filters the same way).  Checked against the full record
path: byte for byte the records a host-side `axions.filterIt(it.passed)` keeps."""
import ctypes as C
import os

import numpy as np
import pytest

import solaraxionraytracing_amd as sa
from solaraxionraytracing_amd import _lib as L
from tests.conftest import make_setup

pytestmark = pytest.mark.gpu


def reference_of(rt, n, **kw):
    full = rt.traceAxionWrapper(n, **kw)
    counts = {"n_rays": n, "n_passed": int(full["passed"].sum()), "n_passed_till_window": int(full["passedTillWindow"].sum()),
              "n_hit_nickel": int(full["hitNickel"].sum())}
    # (rows of the byte view: a fancy-indexed copy of a structured array does not carry the padding bytes)
    return full.view(np.uint8).reshape(n, 208)[full["passed"] != 0], counts


@pytest.mark.parametrize("name", ["babyiaxo_xmm", "cast_llnl", "babyiaxo_xmm_gas"])
@pytest.mark.parametrize("n", [1, 63, 1025, 200_003])
def test_passed_records_are_the_filtered_buffer(name, n):
    with sa.RayTracer(make_setup(name)) as rt:
        want, counts = reference_of(rt, n, seed=17, ray_id_offset=5)
        got, c = rt.traceAxionWrapperPassed(n, seed=17, ray_id_offset=5)
    assert c == counts
    assert got.tobytes() == want.tobytes()
    if n >= 1025:
        assert 0 < c["n_passed"] < n and c["n_passed_till_window"] >= c["n_passed"]


def test_chunked_pipeline_ragged_chunks_and_small_capacity():
    """Several chunks through the two half-buffers (SART_RECORDS_CHUNK), chunk sizes that are no multiple of the compaction's
    blocks; a capacity below the number of passed rays keeps the first `capacity` of them and still counts all."""
    n = 150_001
    with sa.RayTracer(make_setup("cast_llnl")) as rt:
        want, counts = reference_of(rt, n, seed=31, ray_id_offset=7)
    for chunk in (65_536, 40_001, 1_000, 149_999):
        os.environ["SART_RECORDS_CHUNK"] = str(chunk)
        try:
            with sa.RayTracer(make_setup("cast_llnl")) as rt:
                got, c = rt.traceAxionWrapperPassed(n, seed=31, ray_id_offset=7)
                assert c == counts and got.tobytes() == want.tobytes(), chunk
                cap = counts["n_passed"] // 3
                guard = np.full(cap + 5, 0, dtype=L.AXION_DTYPE)
                guard["weights"] = -7.0
                few, c2 = rt.traceAxionWrapperPassed(n, seed=31, ray_id_offset=7, capacity=cap, out=guard)
                assert c2 == counts and len(few) == cap and few.tobytes() == want[:cap].tobytes()
                assert (guard[cap:]["weights"] == -7.0).all()          # nothing behind the capacity is touched
                none, c3 = rt.traceAxionWrapperPassed(n, seed=31, ray_id_offset=7, capacity=0, out=guard)
                assert c3 == counts and len(none) == 0
        finally:
            os.environ.pop("SART_RECORDS_CHUNK", None)


def test_no_ray_passes_and_no_rays():
    full = make_setup("babyiaxo_xmm")
    full.setup.chip_x_max = full.setup.chip_y_max = 1e-6      # a chip nobody hits
    with sa.RayTracer(full) as rt:
        got, c = rt.traceAxionWrapperPassed(50_000, seed=3)
        assert len(got) == 0 and c["n_passed"] == 0 and c["n_rays"] == 50_000 and c["n_passed_till_window"] > 0
        got, c = rt.traceAxionWrapperPassed(0)
        assert len(got) == 0 and c == {"n_rays": 0, "n_passed": 0, "n_passed_till_window": 0, "n_hit_nickel": 0}


def test_device_entry_appends_over_launches():
    """The device form with accumulate: two launches into one buffer = one launch over both ranges; 3 Mi rays cross the 2^20
    chunk of the compaction inside one call."""
    import torch
    n1, n2 = 1_300_000, 1_845_729
    with sa.RayTracer(make_setup("babyiaxo_xmm")) as rt:
        want, counts = reference_of(rt, n1 + n2, seed=9, ray_id_offset=11)
        cap = counts["n_passed"] + 10
        buf = torch.zeros(cap * 208, dtype=torch.uint8, device="cuda")
        cnt = torch.full((4,), 99, dtype=torch.int64, device="cuda")
        p = rt.trace_params(n1, seed=9, ray_id_offset=11)
        rt.trace_records_passed_device(p, buf.data_ptr(), cap, cnt.data_ptr())
        p = rt.trace_params(n2, seed=9, ray_id_offset=11 + n1, accumulate=True)
        rt.trace_records_passed_device(p, buf.data_ptr(), cap, cnt.data_ptr())
        rt.synchronize()
        assert cnt.tolist() == [counts[k] for k in ("n_rays", "n_passed", "n_passed_till_window", "n_hit_nickel")]
        got = buf.cpu().numpy()[:counts["n_passed"] * 208].view(L.AXION_DTYPE)
        assert got.tobytes() == want.tobytes()
        assert not buf[counts["n_passed"] * 208:].any().item()
        # a device buffer with room for half of them: the first half arrives, nothing is written behind it, the counts are whole
        half = counts["n_passed"] // 2
        small = torch.full(((half + 3) * 208,), 0xAB, dtype=torch.uint8, device="cuda")
        p = rt.trace_params(n1 + n2, seed=9, ray_id_offset=11)
        rt.trace_records_passed_device(p, small.data_ptr(), half, cnt.data_ptr())
        rt.synchronize()
        assert cnt.tolist()[1] == counts["n_passed"] and cnt.tolist()[0] == n1 + n2
        assert small[:half * 208].cpu().numpy().tobytes() == want[:half].tobytes() and (small[half * 208:] == 0xAB).all().item()


def test_failure_inside_the_pipeline_leaves_through_the_synchronised_exit():
    os.environ["SART_RECORDS_CHUNK"] = "30000"
    os.environ["SART_RECORDS_FAIL_CHUNK"] = "3"
    try:
        with sa.RayTracer(make_setup("cast_llnl")) as rt:
            with pytest.raises(L.SartError, match="SART_RECORDS_FAIL_CHUNK"):
                rt.traceAxionWrapperPassed(200_000, seed=1)
            os.environ.pop("SART_RECORDS_FAIL_CHUNK")
        with sa.RayTracer(make_setup("cast_llnl")) as rt:
            got, c = rt.traceAxionWrapperPassed(200_000, seed=1)      # the context of a new call works
            assert c["n_passed"] == len(got) > 100_000
    finally:
        os.environ.pop("SART_RECORDS_CHUNK", None)
        os.environ.pop("SART_RECORDS_FAIL_CHUNK", None)


def test_null_arguments():
    with sa.RayTracer(make_setup("cast_llnl")) as rt:
        p = rt.trace_params(10)
        cnt = L.RecordCounts()
        assert rt.lib.sart_trace_records_passed(rt.handle, C.byref(p), None, 10, C.byref(cnt)) == L.SART_ERR_INVALID_ARGUMENT
        assert rt.lib.sart_trace_records_passed(rt.handle, C.byref(p), None, 0, None) == L.SART_ERR_INVALID_ARGUMENT
        assert rt.lib.sart_trace_records_passed(rt.handle, C.byref(p), None, 0, C.byref(cnt)) == 0 and cnt.n_rays == 10


def test_record_scratch_only_grows_and_can_be_released():
    """the record entries keep their device scratch (up to three buffers of 2^20 records) in the context.  It only
    grows - calls of different sizes do not free and reallocate (hipFree synchronises the device) - and sart_release_scratch
    (ABI 5) hands it back; the next call allocates again and returns the same bytes."""
    import torch
    with sa.RayTracer(make_setup("babyiaxo_xmm")) as rt:
        torch.cuda.synchronize()
        free0 = torch.cuda.mem_get_info()[0]
        big, cb = rt.traceAxionWrapperPassed(300_000, seed=4)
        after_big = torch.cuda.mem_get_info()[0]
        assert free0 - after_big >= 300_000 * 208                          # at least the trace buffer
        small, cs = rt.traceAxionWrapperPassed(40_000, seed=4)
        again, ca = rt.traceAxionWrapperPassed(300_000, seed=4)
        assert torch.cuda.mem_get_info()[0] == after_big                   # nothing freed, nothing allocated in between
        assert again.tobytes() == big.tobytes() and ca == cb and small.tobytes() == big[:cs["n_passed"]].tobytes()
        L.check(rt.lib.sart_release_scratch(rt.handle))
        assert torch.cuda.mem_get_info()[0] >= after_big + 300_000 * 208   # handed back
        once_more, cm = rt.traceAxionWrapperPassed(300_000, seed=4)
        assert once_more.tobytes() == big.tobytes() and cm == cb
        assert rt.lib.sart_release_scratch(None) == L.SART_ERR_INVALID_ARGUMENT
