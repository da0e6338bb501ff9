"""Multi-GPU sharding of the hot path: one process per GPU, rays shard by global ray id, one reduce of
the fused output accumulator (image + scalars) per image — RCCL over xGMI on the GPU box
(``torch.distributed`` backend "nccl"), gloo in the CPU tests.

The reference has no distributed layer (weave threads over one ``parallelFor``, raytracer.nim:2234);
rays are independent, so the partition is a pure index split and the only exchange is the final sum.
Because the Philox counter is the *global* ray id, the union of the shards is the same set of rays for
any world size; results agree up to f64 summation order - or bit for bit in the SART_ACCUM_FIXED64 accumulation mode
(integer accumulators, reduced as int64).
"""
from __future__ import annotations

import os
import sys
from typing import Callable, Tuple

import numpy as np


def shard_range(n_total: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Global ray ids [lo, hi) of ``rank``: contiguous blocks, remainder spread over the first ranks."""
    base, rem = divmod(int(n_total), int(world_size))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def step_shard(scaling: str, rays_per_step: int, rank: int, world_size: int):
    """What one rank traces in one benchmark / production step: returns (rays of this rank, rays of the whole step, offset of
    this rank's first ray inside the step).  "weak": every rank traces ``rays_per_step`` rays (per-GPU work fixed);
    "strong": ``rays_per_step`` is the step's total, split by ``shard_range``.  Step k of the job then covers the global ray
    ids [k * step_total, (k + 1) * step_total) exactly once, whatever the world size."""
    n = int(rays_per_step)
    if scaling == "weak":
        return n, n * world_size, rank * n
    if scaling == "strong":
        lo, hi = shard_range(n, rank, world_size)
        return hi - lo, n, lo
    raise ValueError("scaling must be 'weak' or 'strong'")


def shard_angles(n_angles: int, rank: int, world_size: int):
    """Angle bins of an angular scan owned by ``rank`` (round-robin, performAngularScan :2791-2800 is a
    loop over independent full runs)."""
    return list(range(rank, n_angles, world_size))


MAX_RANKS_ON_ONE_DEVICE = 6   # rehearsal mode (all ranks on one card): the GPU boxes allow six processes per card


class LaunchRefused(SystemExit):
    """Raised (exit code 2) when the requested number of ranks cannot be honoured: nothing has touched a GPU yet."""

    def __init__(self, msg: str):
        print("refused: " + msg, file=sys.stderr)
        super().__init__(2)


def visible_devices() -> int:
    """Number of HIP devices this process could use, WITHOUT initialising one (torch.cuda.device_count() only asks the
    driver; torch.cuda.is_available() / any tensor on a device would create a context, after which spawning or re-executing
    is no longer safe on the GPU boxes)."""
    import torch
    try:
        return int(torch.cuda.device_count())
    except Exception:
        return 0


def check_world(n_ranks: int, backend: str | None, shared_device: int | None, n_devices: int | None = None) -> None:
    """The rules a launch of ``n_ranks`` ranks must meet, checked before anything touches a GPU: one device per rank — or,
    rehearsal mode, all ranks on ``shared_device`` with a backend that allows it (gloo; RCCL refuses two ranks on one
    card) and at most MAX_RANKS_ON_ONE_DEVICE of them."""
    n_dev = visible_devices() if n_devices is None else n_devices
    if n_ranks < 1:
        raise LaunchRefused("--gpus must be >= 1")
    if shared_device is not None:
        if shared_device < 0 or shared_device >= n_dev:
            raise LaunchRefused("SART_BENCH_DEVICE=%d but %d device(s) are visible" % (shared_device, n_dev))
        if n_ranks > 1 and backend != "gloo":
            raise LaunchRefused("%d ranks on one device need SART_BENCH_BACKEND=gloo (RCCL wants one device per rank)" % n_ranks)
        if n_ranks > MAX_RANKS_ON_ONE_DEVICE:
            raise LaunchRefused("%d ranks on one device: at most %d processes may share a card" % (n_ranks, MAX_RANKS_ON_ONE_DEVICE))
    elif n_ranks > n_dev:
        raise LaunchRefused("--gpus %d but only %d device(s) are visible" % (n_ranks, n_dev))


def free_port() -> int:
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return int(s.getsockname()[1])


def launch_ranks_if_needed(n_ranks: int, script: str, argv: list) -> int | None:
    """Makes ``<script> --gpus N`` mean N ranks whoever starts it.

    * Under a launcher (WORLD_SIZE set: torchrun, the driver's command line): this process is one rank; returns None after
      checking that WORLD_SIZE == ``n_ranks`` and that enough devices exist (LaunchRefused = exit code 2 otherwise: a
      benchmark line whose ``n_gpus`` is not what was asked for must not be printed).
    * Stand-alone with ``n_ranks`` == 1: returns None (single process, no process group).
    * Stand-alone with ``n_ranks`` > 1: starts ``n_ranks`` fresh copies of ``script`` (children of this process, one rank
      each, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT in their environment) BEFORE any HIP call
      here, waits for them and returns the exit code to leave with (rank 0's stdout is this process's stdout).  A rank that
      fails ends the others (exact PIDs).

    Environment: SART_BENCH_BACKEND (nccl = RCCL by default, gloo for rehearsals), SART_BENCH_DEVICE (all ranks share that
    device: rehearsal of the multi-rank path on a one-GPU box)."""
    import subprocess
    import time

    # the pool's host driver shares device memory between processes through dmabuf only; RCCL needs this before the first
    # HIP call of a rank (it is exported on the GPU boxes already: this keeps a hand-built environment from losing it)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    backend = os.environ.get("SART_BENCH_BACKEND") or "nccl"
    shared = int(os.environ["SART_BENCH_DEVICE"]) if "SART_BENCH_DEVICE" in os.environ else None
    if "WORLD_SIZE" in os.environ:
        world = int(os.environ["WORLD_SIZE"])
        if world != n_ranks:
            raise LaunchRefused("--gpus %d but WORLD_SIZE=%d" % (n_ranks, world))
        n_dev = visible_devices()
        if shared is None and n_dev == 1 and world > 1:
            # a launcher that shows every rank its own card only (HIP_VISIBLE_DEVICES per rank): this rank uses device 0.  If
            # the ranks really share one card, RCCL refuses the duplicate device when the process group comes up - loudly.
            return None
        check_world(world, backend, shared, n_dev)
        return None
    check_world(n_ranks, backend, shared)
    if n_ranks == 1:
        return None
    port = free_port()
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SART_BENCH_BACKEND=backend)
        procs.append(subprocess.Popen([sys.executable, script] + list(argv), env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    import signal

    def stop_children(signum, _frame):   # a killed launcher must not leave ranks behind on the GPUs (exact PIDs)
        for q in procs:
            if q.poll() is None:
                q.terminate()
        raise SystemExit(128 + signum)

    old = {sig: signal.signal(sig, stop_children) for sig in (signal.SIGTERM, signal.SIGINT)}
    rc = 0
    alive = list(procs)
    try:
        while alive:
            for p in list(alive):
                code = p.poll()
                if code is None:
                    continue
                alive.remove(p)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 1
                    for q in alive:          # a dead rank leaves the others waiting in a collective: end them
                        q.terminate()
            time.sleep(0.05)
    finally:
        for q in procs:                      # whatever ends the wait (an exception, a signal): no rank outlives the launcher
            if q.poll() is None:
                q.terminate()
        for q in procs:
            try:
                q.wait(timeout=10)
            except Exception:
                q.kill()
        for sig, h in old.items():
            signal.signal(sig, h)
    return rc


def init_process_group_from_env(backend: str | None = None):
    """Initialises torch.distributed from RANK / WORLD_SIZE / MASTER_* (torchrun contract).  Returns
    (rank, world_size, local_rank) - local_rank = the device index of this rank: LOCAL_RANK, or 0 when the launcher shows
    every rank one card only.  world_size 1 needs no process group."""
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and visible_devices() == 1:
        local_rank = 0
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local_rank


def reduce_accumulator(acc, dst: int | None = 0, fixed64: bool = False):
    """The single collective of the path: sum the fused accumulator tensor over ranks (``dst`` = root rank, or
    None for an all-reduce).  No-op without a process group.  ``fixed64``: the 8-byte slots hold the int64 of a raw
    SART_ACCUM_FIXED64 accumulator - they are summed as integers (exact, so the result does not depend on the number of
    ranks or on the reduction tree)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return acc
    buf = acc.view(torch.int64) if (fixed64 and acc.dtype != torch.int64) else acc
    if dst is None:
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    else:
        dist.reduce(buf, dst=dst, op=dist.ReduceOp.SUM)
    return acc


def trace_sharded(trace_fn: Callable[[int, int], "np.ndarray"], n_total: int, rank: int, world_size: int,
                  to_tensor: Callable | None = None, dst: int | None = 0, fixed64: bool = False):
    """Runs ``trace_fn(ray_id_offset, n_rays) -> accumulator`` on this rank's shard of [0, n_total) and reduces.
    ``trace_fn`` returns either a torch tensor (device accumulator on the GPU box) or a numpy array."""
    import torch

    lo, hi = shard_range(n_total, rank, world_size)
    acc = trace_fn(lo, hi - lo)
    if not isinstance(acc, torch.Tensor):
        acc = torch.from_numpy(np.ascontiguousarray(acc)) if to_tensor is None else to_tensor(acc)
    return reduce_accumulator(acc, dst, fixed64)


def gather_scan(values_local, indices_local, n_total: int):
    """Angular scan sharded by angle bin: every rank holds the fluxes of its bins; returns the full curve on
    every rank (one all-reduce of an n_total vector with zeros elsewhere)."""
    import torch
    import torch.distributed as dist

    full = torch.zeros(n_total, dtype=torch.float64, device=values_local.device if isinstance(values_local, torch.Tensor) else "cpu")
    vals = values_local if isinstance(values_local, torch.Tensor) else torch.as_tensor(np.asarray(values_local), dtype=torch.float64)
    if len(indices_local):
        full[torch.as_tensor(indices_local, dtype=torch.long, device=full.device)] = vals.to(full.device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(full, op=dist.ReduceOp.SUM)
    return full
