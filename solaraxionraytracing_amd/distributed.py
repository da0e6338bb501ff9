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
    """Number of HIP devices this process could use.  torch.cuda.device_count() does not create a HIP context on this image
    (torch.cuda.is_available() or any tensor on a device would), but on ROCm it may still call hipGetDeviceCount and so load
    the HSA runtime in this process.  That is harmless for how the result is used here - the launcher below only ever starts
    fresh CHILD processes after it (subprocess.Popen) and never re-executes itself; a process that has asked must not
    os.exec* another program."""
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


START_TIMEOUT_S = 600         # from the launch to the rendezvous of every rank: interpreter + `import torch` on a fresh box (SART_START_TIMEOUT)
RDZV_TIMEOUT_S = 120          # torch.distributed rendezvous + communicator bring-up of one rank (SART_RDZV_TIMEOUT overrides)
LAUNCH_TIMEOUT_S = 0          # wall-clock limit of a self-launched multi-rank run: none unless SART_LAUNCH_TIMEOUT asks for one
TERM_GRACE_S = 10             # between SIGTERM and SIGKILL for ranks that are being ended (SART_TERM_GRACE)
STALL_TIMEOUT_S = 900         # silence of EVERY live rank after one of them has reported progress (heartbeat): SART_STALL_TIMEOUT, 0 = off
STAGES = ("started", "rendezvous", "up", "done")   # + "beat:<what>" (heartbeat): the group is up and the rank is working


def report_stage(stage: str) -> None:
    """Tells the launcher how far this rank has come (one small file per rank in SART_RANK_STATUS_DIR; no-op without it):
    when a run hangs, the launcher names the rank that never reported and the stage the others reached."""
    d = os.environ.get("SART_RANK_STATUS_DIR")
    if not d:
        return
    try:
        with open(os.path.join(d, "rank%s" % os.environ.get("RANK", "0")), "w") as f:
            f.write(stage)
    except OSError:
        pass


def heartbeat(what: str) -> None:
    """Progress report of a rank whose process group is up ("beat:<what>" in its status file; no-op without a launcher).  A run
    without a wall limit (the default: a long production run is not a hang) still must not wait for ever on a collective that
    never completes - with every rank alive nothing else would end it.  Scripts that call this between their steps
    get a stall clock in the launcher: when EVERY live rank has been silent for SART_STALL_TIMEOUT seconds (default 900) the
    ranks are ended, the exit code is 3 and the last report of each rank says where it stopped.  Scripts that never call it are
    not watched."""
    report_stage("beat:" + what)


def _is_up(stage: str) -> bool:
    return stage in ("up", "done") or stage.startswith("beat:")


def _last_reports(status_dir: str, ranks) -> float:
    """Seconds since the most recent status report of any of `ranks` (inf if none of them ever reported)."""
    import time
    newest = 0.0
    for r in ranks:
        try:
            newest = max(newest, os.path.getmtime(os.path.join(status_dir, "rank%d" % r)))
        except OSError:
            pass
    return time.time() - newest if newest else float("inf")


def _rank_stages(status_dir: str, n_ranks: int) -> list:
    out = []
    for r in range(n_ranks):
        try:
            out.append(open(os.path.join(status_dir, "rank%d" % r)).read().strip() or "nothing")
        except OSError:
            out.append("nothing")
    return out


def _forward(pipe, prefix: str, sink) -> None:
    """Copies a child's output line by line with the rank in front (eight ranks' stderr interleaved untagged is unreadable)."""
    try:
        for line in iter(pipe.readline, b""):
            sink.write(prefix.encode() + line)
            sink.flush()
    except (OSError, ValueError):
        pass
    finally:
        try:
            pipe.close()
        except OSError:
            pass


def launch_ranks_if_needed(n_ranks: int, script: str, argv: list, need_devices: bool = True) -> int | None:
    """Makes ``<script> --gpus N`` mean N ranks whoever starts it.

    * Under a launcher (WORLD_SIZE set: torchrun, the driver's command line): this process is one rank; returns None after
      checking that WORLD_SIZE == ``n_ranks`` and that enough devices exist (LaunchRefused = exit code 2 otherwise: a
      benchmark line whose ``n_gpus`` is not what was asked for must not be printed).
    * Stand-alone with ``n_ranks`` == 1: returns None (single process, no process group).
    * Stand-alone with ``n_ranks`` > 1: starts ``n_ranks`` fresh copies of ``script`` (children of this process, one rank
      each, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT in their environment) BEFORE any HIP call
      here, waits for them and returns the exit code to leave with.  Rank 0's stdout is this process's stdout; everything else
      the ranks print (stderr of all, stdout of ranks > 0) arrives on this process's stderr with "[rank r] " in front.  A rank
      that fails ends the others (exact PIDs).  Clocks guard a first contact with an 8-GPU node: every rank must reach the
      rendezvous within SART_START_TIMEOUT (default 600 s: the first `import torch` on a fresh box takes minutes), every rank
      must report its process group up (report_stage) within SART_RDZV_TIMEOUT + 60 s of the first rank reaching the
      rendezvous; past either the ranks are ended and the exit code is 3, with one line per rank saying what it last
      reported - a hang becomes a diagnosable failure instead of the driver's own limit.  A limit on the whole run is opt-in
      (SART_LAUNCH_TIMEOUT=<seconds>; unset or 0 = none: a long production run is not a hang); what ends a run that hangs with
      every rank alive (a collective that never completes) is the stall clock: ranks that report progress (heartbeat) and then
      ALL fall silent for SART_STALL_TIMEOUT (900 s) are ended with exit code 3.  Ending ranks means SIGTERM by
      exact PID, then - for ranks still alive SART_TERM_GRACE (10 s) later, e.g. blocked in a driver call - SIGKILL, with their
      numbers on stderr: the launcher itself always returns.

    Environment: SART_BENCH_BACKEND (nccl = RCCL by default, gloo for rehearsals), SART_BENCH_DEVICE (all ranks share that
    device: rehearsal of the multi-rank path on a one-GPU box).  need_devices=False (a gloo preflight: nothing will touch a
    card) skips the device count."""
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
        if not need_devices:
            return None
        n_dev = visible_devices()
        if shared is None and n_dev == 1 and world > 1:
            # a launcher that shows every rank its own card only (HIP_VISIBLE_DEVICES per rank): this rank uses device 0.  If
            # the ranks really share one card, RCCL refuses the duplicate device when the process group comes up - loudly.
            return None
        check_world(world, backend, shared, n_dev)
        return None
    if need_devices:
        check_world(n_ranks, backend, shared)
    elif n_ranks < 1:
        raise LaunchRefused("--gpus must be >= 1")
    if n_ranks == 1:
        return None
    import tempfile
    import threading
    port = free_port()
    rdzv_limit = float(os.environ.get("SART_RDZV_TIMEOUT", RDZV_TIMEOUT_S)) + float(os.environ.get("SART_RDZV_MARGIN", 60.0))
    start_limit = float(os.environ.get("SART_START_TIMEOUT", START_TIMEOUT_S))
    wall_limit = float(os.environ.get("SART_LAUNCH_TIMEOUT", LAUNCH_TIMEOUT_S))
    stall_limit = float(os.environ.get("SART_STALL_TIMEOUT", STALL_TIMEOUT_S))
    status_dir = tempfile.mkdtemp(prefix="sart_ranks_")
    procs, pumps = [], []
    err_sink = getattr(sys.stderr, "buffer", None) or sys.stderr
    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SART_BENCH_BACKEND=backend, SART_RANK_STATUS_DIR=status_dir,
                   PYTHONUNBUFFERED="1")
        p = subprocess.Popen([sys.executable, script] + list(argv), env=env, stdout=None if r == 0 else subprocess.PIPE,
                             stderr=subprocess.PIPE)
        procs.append(p)
        for pipe in ((p.stderr,) if r == 0 else (p.stderr, p.stdout)):
            t = threading.Thread(target=_forward, args=(pipe, "[rank %d] " % r, err_sink), daemon=True)
            t.start()
            pumps.append(t)
    import signal

    def stop_children(signum, _frame):   # a killed launcher must not leave ranks behind on the GPUs (exact PIDs)
        for q in procs:
            if q.poll() is None:
                q.terminate()
        raise SystemExit(128 + signum)

    old = {sig: signal.signal(sig, stop_children) for sig in (signal.SIGTERM, signal.SIGINT)}
    rc = 0
    alive = list(procs)
    t_start = time.monotonic()
    t_rdzv = None       # when the first rank reached the rendezvous
    t_term = None       # when the remaining ranks were told to end (SIGTERM)
    all_up = False
    beating = False     # a rank has sent a heartbeat: the stall clock applies
    grace = float(os.environ.get("SART_TERM_GRACE", TERM_GRACE_S))
    try:
        while alive:
            for p in list(alive):
                code = p.poll()
                if code is None:
                    continue
                alive.remove(p)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 1
                    others = [procs.index(q) for q in alive]
                    print("launcher: rank %d exited with code %d%s" % (procs.index(p), code, "; ending rank(s) %s, which would wait for it in the "
                          "next collective" % ", ".join(map(str, others)) if others else ""), file=sys.stderr)
                    for q in alive:          # a dead rank leaves the others waiting in a collective: end them
                        q.terminate()
                    t_term = time.monotonic()
            if alive and t_term is not None and time.monotonic() - t_term > grace:
                # SIGTERM was not enough (a rank blocked in a driver call does not see it): SIGKILL, by exact PID, and say which
                print("launcher: rank(s) %s still alive %.0f s after SIGTERM: killed" % (", ".join(str(procs.index(q)) for q in alive), grace),
                      file=sys.stderr)
                for q in alive:
                    q.kill()
                t_term = time.monotonic() + 1e9   # (once)
            elapsed = time.monotonic() - t_start
            if alive and rc == 0:
                what = None
                if not all_up:
                    stages = _rank_stages(status_dir, n_ranks)
                    all_up = all(_is_up(s) for s in stages)
                    if t_rdzv is None and any(s != "nothing" and s != "started" for s in stages):
                        t_rdzv = time.monotonic()
                    if not all_up and t_rdzv is not None and time.monotonic() - t_rdzv > rdzv_limit:
                        what = "the process group did not come up within %.0f s of the first rank's rendezvous" % rdzv_limit
                    elif not all_up and elapsed > start_limit and any(s in ("nothing", "started") for s in stages):
                        what = "not every rank reached the rendezvous within %.0f s" % start_limit
                if what is None and wall_limit > 0 and elapsed > wall_limit:
                    what = "the run did not end within %.0f s (SART_LAUNCH_TIMEOUT)" % wall_limit
                if what is None and all_up and stall_limit > 0:
                    # the stall clock runs for scripts that report progress (heartbeat): every live rank silent for stall_limit
                    beating = beating or any(s.startswith("beat:") for s in _rank_stages(status_dir, n_ranks))
                    if beating and _last_reports(status_dir, [procs.index(q) for q in alive]) > stall_limit:
                        what = "no live rank reported progress for %.0f s (SART_STALL_TIMEOUT): a collective that never completes?" % stall_limit
                if what is not None:
                    stages = _rank_stages(status_dir, n_ranks)
                    print("launcher: %s; last report of every rank: %s" % (what, ", ".join("rank %d: %s" % (r, s) for r, s in enumerate(stages))),
                          file=sys.stderr)
                    never = [r for r, s in enumerate(stages) if not _is_up(s)]
                    if never and not all_up:
                        print("launcher: rank(s) %s never reported their process group up" % ", ".join(map(str, never)), file=sys.stderr)
                    rc = 3
                    for q in alive:
                        q.terminate()
                    t_term = time.monotonic()
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
        for t in pumps:
            t.join(timeout=2)
        try:
            import shutil
            shutil.rmtree(status_dir, ignore_errors=True)
        except Exception:
            pass
    return rc


def init_process_group_from_env(backend: str | None = None):
    """Initialises torch.distributed from RANK / WORLD_SIZE / MASTER_* (torchrun contract).  Returns
    (rank, world_size, local_rank) - local_rank = the device index of this rank: LOCAL_RANK, or 0 when the launcher shows
    every rank one card only.  world_size 1 needs no process group."""
    import torch
    import torch.distributed as dist

    from datetime import timedelta
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    report_stage("started")
    if world > 1 and visible_devices() == 1:
        local_rank = 0
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # a rendezvous or RCCL bring-up that stalls must end in an exception (non-zero exit), not in a wait for ever
        timeout = timedelta(seconds=float(os.environ.get("SART_RDZV_TIMEOUT", RDZV_TIMEOUT_S)))
        report_stage("rendezvous")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=timeout, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=timeout)
    report_stage("up")
    return rank, world, local_rank


def preflight(backend: str | None = None, n_bytes: int = 512 * 1024) -> dict:
    """First contact with a multi-GPU node without the workload: bring the process group up, reduce one buffer of the size of
    the fused accumulator (512 KB) to rank 0 once, report the time.  Returns the dictionary rank 0 prints as JSON."""
    import time
    import torch
    import torch.distributed as dist
    t0 = time.perf_counter()
    rank, world, local_rank = init_process_group_from_env(backend)
    up_ms = (time.perf_counter() - t0) * 1e3
    be = dist.get_backend() if world > 1 else "none"
    on_gpu = be == "nccl" or (world == 1 and torch.cuda.is_available())
    if "SART_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["SART_BENCH_DEVICE"])
    dev = torch.device("cuda", local_rank) if on_gpu else torch.device("cpu")
    buf = torch.full((n_bytes // 8,), float(rank + 1), dtype=torch.float64, device=dev)
    if on_gpu:
        torch.cuda.synchronize(dev)
    t1 = time.perf_counter()
    reduce_accumulator(buf, dst=0)
    if on_gpu:
        torch.cuda.synchronize(dev)
    reduce_ms = (time.perf_counter() - t1) * 1e3
    ok = True
    if rank == 0:
        ok = bool((buf == float(world * (world + 1) // 2)).all().item())
    if world > 1:
        dist.barrier()
    out = {"preflight": "ok" if ok else "wrong sum", "world_size": world, "backend": be, "device": str(dev), "process_group_up_ms": up_ms,
           "reduce_ms": reduce_ms, "reduce_bytes": n_bytes}
    if world > 1:
        dist.destroy_process_group()
    report_stage("done")
    return out


def reduce_accumulator(acc, dst: int | None = 0, fixed64: bool = False, even_alone: bool = False):
    """The single collective of the path: sum the fused accumulator tensor over ranks (``dst`` = root rank, or
    None for an all-reduce).  No-op without a process group.  ``fixed64``: the 8-byte slots hold the int64 of a raw
    SART_ACCUM_FIXED64 accumulator - they are summed as integers (exact, so the result does not depend on the number of
    ranks or on the reduction tree).  ``even_alone``: issue the collective in a one-rank group too (the one-GPU rehearsal of
    the RCCL call: communicator, dtype and stream handling are what an N-rank run uses; tests/test_gpu_parity.py)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not even_alone):
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
