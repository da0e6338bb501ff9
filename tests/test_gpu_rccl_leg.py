"""The RCCL leg of the C-ABI (`sart_reduce_across_devices`) executed line by line on ONE card.

The public entry returns for n == 1 before any RCCL call (one accumulator: nothing to add up), and a one-GPU box cannot
offer n > 1.  The test entry `sart_internal_reduce_across_devices` (sart_api.hip, not part of include/sart.h) is the same
body with two switches: take the RCCL route for n == 1 too, and hand ncclReduce an element type that does not exist.  So
what runs here is everything a C / Nim host would run on a node, with a group of one: dlopen + symbol lookup,
ncclCommInitAll, ncclGroupStart / ncclReduce(ncclSum) / ncclGroupEnd on the context's stream in both element types (f64 and
the raw int64 of SART_ACCUM_FIXED64), the communicator cache, and the failure branch that drops the communicators.
(The reference's parallelism for this path is `weave.parallelFor` over independent rays, raytracer.nim:2234: the reduce
of the output histograms is what replaces its shared `axBuf`.)"""
import ctypes as C

import numpy as np
import pytest

import solaraxionraytracing_amd as sa
from solaraxionraytracing_amd import _lib as L

pytestmark = pytest.mark.gpu

FORCE_RCCL, INJECT_FAILURE = 1, 2


def _entry():
    lib = L.load_sart()
    fn = lib.sart_internal_reduce_across_devices
    fn.restype = C.c_int
    fn.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int32, C.c_size_t, C.c_int32,
                   C.c_uint32, C.POINTER(C.c_int32)]
    return lib, fn


def _reduce(fn, rt, acc, recv=None, flags=FORCE_RCCL):
    ctxs = (C.c_void_p * 1)(rt.handle)
    accs = (C.c_void_p * 1)(acc.data_ptr())
    recvs = (C.c_void_p * 1)(recv.data_ptr()) if recv is not None else None
    info = (C.c_int32 * 2)(-1, -1)
    rc = fn(ctxs, accs, recvs, 1, acc.numel(), 0, flags, info)
    return rc, info[0], info[1]


def _setup():
    return sa.initFullSetup(n_radii=400, n_energies=300, refl_n_angles=200, refl_n_energies=200)


@pytest.mark.parametrize("mode", ["f64", "fixed64"])
def test_accumulator_comes_back_bit_for_bit_through_the_rccl_leg(mode):
    """The accumulator the kernel has just filled goes through ncclCommInitAll + grouped ncclReduce (in place, as the public
    entry does it, and out of place) and keeps every bit; the second call finds the communicators in the cache."""
    import torch
    lib, fn = _entry()
    n = 200_000
    with sa.RayTracer(_setup()) as rt:
        rt.set_accumulation_mode(mode)
        acc = torch.zeros(sa.accumulator_len(256), dtype=torch.float64, device="cuda")
        p = rt.trace_params(n, seed=11)
        rt.trace_histogram_device(p, acc.data_ptr())          # still queued: the entry must wait for the launch itself
        rc, hit0, alive0 = _reduce(fn, rt, acc)
        assert rc == 0, lib.sart_last_error()
        assert alive0 >= 1
        before = acc.view(torch.int64).clone()
        assert before.count_nonzero().item() > 1000           # the launch had finished: the image is there
        rc, hit, alive = _reduce(fn, rt, acc)                 # in place, second call
        assert rc == 0 and hit == 1 and alive == alive0, (rc, hit, alive, lib.sart_last_error())
        assert torch.equal(acc.view(torch.int64), before)
        recv = torch.zeros_like(acc)                          # out of place: the bytes really travel through the collective
        rc, hit, alive = _reduce(fn, rt, acc, recv)
        assert rc == 0 and hit == 1
        assert torch.equal(recv.view(torch.int64), before) and torch.equal(acc.view(torch.int64), before)
        # what the host does next with the reduced accumulator: finalize (fixed64) and read the scalars
        if mode == "fixed64":
            rt.finalize_accumulator_device(p, recv.data_ptr())
        rt.synchronize()
        host = recv.cpu().numpy()
        assert host[256 * 256 + L.ACC["N_RAYS"]] == n
        img, summ = rt.trace_histogram(n, seed=11)
        if mode == "fixed64":
            assert np.array_equal(host[:256 * 256].view(np.uint64), img.ravel().view(np.uint64))
            assert host[256 * 256 + L.ACC["SUM_WEIGHTS"]] == summ["SUM_WEIGHTS"]
        else:
            np.testing.assert_allclose(host[:256 * 256], img.ravel(), rtol=1e-9, atol=1e-12 * img.max())
            assert host[256 * 256 + L.ACC["N_PASSED"]] == summ["N_PASSED"]


def test_failed_collective_drops_the_communicators_and_the_next_call_builds_new_ones():
    """ncclReduce refuses an element type that does not exist: the call reports SART_ERR_INTERNAL with RCCL's own words, the
    accumulator is untouched, the communicator set is gone from the cache - and the next call creates a fresh one and works."""
    import torch
    lib, fn = _entry()
    with sa.RayTracer(_setup()) as rt:
        acc = torch.zeros(sa.accumulator_len(256), dtype=torch.float64, device="cuda")
        rt.trace_histogram_device(rt.trace_params(50_000, seed=3), acc.data_ptr())
        rc, _, alive = _reduce(fn, rt, acc)
        assert rc == 0 and alive >= 1
        before = acc.view(torch.int64).clone()
        rc, hit, alive_after = _reduce(fn, rt, acc, flags=FORCE_RCCL | INJECT_FAILURE)
        assert rc == L.SART_ERR_INTERNAL, rc
        assert b"ncclReduce" in lib.sart_last_error(), lib.sart_last_error()
        assert hit == 1 and alive_after == alive - 1
        assert torch.equal(acc.view(torch.int64), before)
        rc, hit, alive_again = _reduce(fn, rt, acc)
        assert rc == 0 and hit == 0 and alive_again == alive, (rc, hit, alive_again, lib.sart_last_error())
        assert torch.equal(acc.view(torch.int64), before)
        # the public entry still takes its n == 1 short cut (a stream synchronisation and nothing else)
        ctxs, accs = (C.c_void_p * 1)(rt.handle), (C.c_void_p * 1)(acc.data_ptr())
        assert lib.sart_reduce_across_devices(ctxs, accs, 1, acc.numel(), 0) == 0


def test_rccl_leg_checks_its_arguments_like_the_public_entry():
    import torch
    lib, fn = _entry()
    with sa.RayTracer(_setup()) as rt:
        acc = torch.zeros(64, dtype=torch.float64, device="cuda")
        info = (C.c_int32 * 2)()
        two = (C.c_void_p * 2)(rt.handle, rt.handle)
        accs2 = (C.c_void_p * 2)(acc.data_ptr(), acc.data_ptr())
        assert fn(two, accs2, None, 2, 64, 0, FORCE_RCCL, info) == L.SART_ERR_INVALID_ARGUMENT     # same device twice
        assert b"distinct devices" in lib.sart_last_error()
        one, accs1 = (C.c_void_p * 1)(rt.handle), (C.c_void_p * 1)(acc.data_ptr())
        assert fn(one, accs1, None, 1, 64, 1, FORCE_RCCL, info) == L.SART_ERR_INVALID_ARGUMENT     # root out of range
        null_recv = (C.c_void_p * 1)(None)
        assert fn(one, accs1, null_recv, 1, 64, 0, FORCE_RCCL, info) == L.SART_ERR_INVALID_ARGUMENT
