"""world_size-2 gloo tests of the multi-GPU path's host logic (sharding by global ray id + the single reduce
of the fused accumulator).  No GPU here, so the per-rank tracer is the CPU oracle standing in for libsart
(the sharding / reduce code under test is the product's solaraxionraytracing_amd.distributed)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from solaraxionraytracing_amd import distributed as D

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 1000, 10 ** 9 + 7):
        for w in (1, 2, 3, 8):
            spans = [D.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    assert sorted(sum((D.shard_angles(16, r, 3) for r in range(3)), [])) == list(range(16))


def test_step_shard_covers_every_ray_id_exactly_once():
    """bench.py / production stepping: weak (fixed rays per GPU) and strong (fixed total) splits tile the id space."""
    for scaling, n in (("weak", 1000), ("strong", 1003), ("strong", 5)):
        for world in (1, 2, 3, 8):
            for steps in (1, 3):
                seen = np.zeros(0, dtype=np.int64)
                total = None
                for k in range(steps):
                    for rank in range(world):
                        rays, step_total, lo = D.step_shard(scaling, n, rank, world)
                        total = step_total
                        seen = np.concatenate([seen, k * step_total + lo + np.arange(rays)])
                assert total == (n * world if scaling == "weak" else n)
                assert np.array_equal(np.sort(seen), np.arange(steps * total)), (scaling, n, world)
    with pytest.raises(ValueError):
        D.step_shard("other", 10, 0, 1)


def _worker(rank, world, port, n_total, seed, out_path):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from tests.conftest import make_setup
    from oracle.oracle import Oracle
    r, w, _ = D.init_process_group_from_env("gloo")
    assert (r, w) == (rank, world)
    o = Oracle(make_setup("babyiaxo_xmm"))

    def trace_fn(offset, n):   # stands in for RayTracer.trace_histogram_device on this rank's GPU
        img, summ, _ = o.trace_histogram(n, seed=seed, ray_id_offset=offset, n_threads=2)
        from solaraxionraytracing_amd import _lib
        acc = np.zeros(img.size + _lib.SART_ACC_COUNT)
        acc[:img.size] = img.ravel()
        for k, i in _lib.ACC.items():
            acc[img.size + i] = summ[k]
        return acc

    acc = D.trace_sharded(trace_fn, n_total, rank, world, dst=0)
    # angle-bin sharding of a scan: each rank "measures" f(angle) for its bins
    idx = D.shard_angles(7, rank, world)
    curve = D.gather_scan(np.array([10.0 + i for i in idx]), idx, 7)
    if rank == 0:
        np.save(out_path, acc.numpy())
        np.save(out_path + ".curve.npy", curve.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_trace_equals_single_process(tmp_path):
    from tests.conftest import make_setup
    from oracle.oracle import Oracle
    from solaraxionraytracing_amd import _lib
    n_total, seed = 30_001, 77
    out = str(tmp_path / "acc.npy")
    mp.spawn(_worker, args=(2, 29611, n_total, seed, out), nprocs=2, join=True)
    acc = np.load(out)
    img, summ, _ = Oracle(make_setup("babyiaxo_xmm")).trace_histogram(n_total, seed=seed)
    n_img = img.size
    # counts are exact; sums agree up to f64 summation order
    for k in ("N_RAYS", "N_PASSED", "N_HIT_NICKEL", "N_REACHED_TELESCOPE", "N_SHELL_SELECTED", "N_PASSED_TILL_WINDOW"):
        assert acc[n_img + _lib.ACC[k]] == summ[k], k
    assert acc[n_img + _lib.ACC["SUM_WEIGHTS"]] == pytest.approx(summ["SUM_WEIGHTS"], rel=1e-12)
    np.testing.assert_allclose(acc[:n_img].reshape(img.shape), img, rtol=1e-12, atol=1e-30)
    np.testing.assert_array_equal(np.load(out + ".curve.npy"), 10.0 + np.arange(7))


# ---- `--gpus N` launches N ranks by itself and refuses to lie (bench.py, tools/scan.py) -------------------------------

_RANK_SCRIPT = '''
import os, sys
sys.path.insert(0, %r)
import torch, torch.distributed as dist
from solaraxionraytracing_amd import distributed as D
r, w, lr = D.init_process_group_from_env("gloo")
t = torch.tensor([float(r + 1)], dtype=torch.float64)
dist.all_reduce(t)
open(os.path.join(sys.argv[1], "rank%%d" %% r), "w").write("%%d %%d %%g" %% (r, w, t.item()))
if r == 0:
    print("hello from rank 0 of", w)
dist.barrier()
dist.destroy_process_group()
if len(sys.argv) > 2 and int(sys.argv[2]) == r:
    sys.exit(7)
'''


def test_launch_ranks_starts_n_fresh_processes(tmp_path, monkeypatch):
    """Stand-alone `--gpus 3`: three children, one rank each, a working process group between them; the parent never
    touches a device (here there is none: the device count is patched, the children use gloo on the CPU)."""
    script = tmp_path / "rank_script.py"
    script.write_text(_RANK_SCRIPT % ROOT)
    monkeypatch.setattr(D, "visible_devices", lambda: 3)
    monkeypatch.setenv("SART_BENCH_BACKEND", "gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "SART_BENCH_DEVICE"):
        monkeypatch.delenv(k, raising=False)
    out = tmp_path / "ok"
    out.mkdir()
    assert D.launch_ranks_if_needed(3, str(script), [str(out)]) == 0
    got = sorted((out / f).read_text() for f in os.listdir(out))
    assert got == ["0 3 6", "1 3 6", "2 3 6"]
    # a failing rank gives a non-zero exit code
    out2 = tmp_path / "fail"
    out2.mkdir()
    assert D.launch_ranks_if_needed(2, str(script), [str(out2), "1"]) == 7
    # one rank: nothing to start
    assert D.launch_ranks_if_needed(1, str(script), [str(out)]) is None


def test_launch_refuses_what_it_cannot_honour(monkeypatch):
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "SART_BENCH_DEVICE", "SART_BENCH_BACKEND"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(D, "visible_devices", lambda: 1)
    with pytest.raises(SystemExit) as e:          # more ranks than devices
        D.launch_ranks_if_needed(8, "unused.py", [])
    assert e.value.code == 2
    monkeypatch.setenv("WORLD_SIZE", "2")         # under a launcher: WORLD_SIZE must be what --gpus says
    with pytest.raises(SystemExit) as e:
        D.launch_ranks_if_needed(4, "unused.py", [])
    assert e.value.code == 2
    monkeypatch.setenv("WORLD_SIZE", "8")         # a launcher that shows every rank one card only: accepted, the rank uses device 0
    assert D.launch_ranks_if_needed(8, "unused.py", []) is None
    monkeypatch.setattr(D, "visible_devices", lambda: 4)   # but four cards for eight ranks is refused
    with pytest.raises(SystemExit):
        D.launch_ranks_if_needed(8, "unused.py", [])
    monkeypatch.setattr(D, "visible_devices", lambda: 1)
    monkeypatch.delenv("WORLD_SIZE")
    monkeypatch.setenv("SART_BENCH_DEVICE", "0")  # several ranks on one card: gloo only, at most six
    with pytest.raises(SystemExit):
        D.launch_ranks_if_needed(2, "unused.py", [])
    monkeypatch.setenv("SART_BENCH_BACKEND", "gloo")
    with pytest.raises(SystemExit):
        D.launch_ranks_if_needed(8, "unused.py", [])
    monkeypatch.setenv("SART_BENCH_DEVICE", "3")  # a device that does not exist
    with pytest.raises(SystemExit):
        D.launch_ranks_if_needed(2, "unused.py", [])


@pytest.mark.parametrize("script", ["bench.py", os.path.join("tools", "scan.py")])
def test_scripts_exit_nonzero_before_touching_a_gpu(script):
    """`python bench.py --gpus 8` where eight devices are not visible (here: none) must not print a benchmark line."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "SART_BENCH_DEVICE")}
    args = ["--gpus", "8"] if script == "bench.py" else ["mass", "--gpus", "8"]
    r = subprocess.run([sys.executable, os.path.join(ROOT, script)] + args, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "refused" in r.stderr and "{" not in r.stdout
    r = subprocess.run([sys.executable, os.path.join(ROOT, script)] + args, env=dict(env, WORLD_SIZE="2", RANK="0"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "WORLD_SIZE" in r.stderr


# ---- SART_ACCUM_FIXED64: raw integer accumulators reduce exactly (include/sart.h "accumulation mode") -------------------

def _raw_fixed_accumulator(rec, weight_exp, chip=14.0, n=256):
    """What the FIXED64 kernels leave in a raw accumulator, rebuilt from Axion records: every passed ray adds
    rint(weight / 2^weight_exp) to its pixel; SUM_WEIGHTS / SUM_X / SUM_Y / SUM_R in two limbs (hi * 2^40 + lo); counters."""
    from solaraxionraytracing_amd import _lib
    acc = np.zeros(n * n + _lib.SART_ACC_COUNT, dtype=np.int64)
    ok = rec["passed"] == 1
    W = np.rint(np.ldexp(rec["weights"][ok], -weight_exp)).astype(np.int64)
    ix = np.floor(rec["pointdataX"][ok] / (chip / n)).astype(np.int64)
    iy = np.floor(rec["pointdataY"][ok] / (chip / n)).astype(np.int64)
    inside = (ix >= 0) & (ix < n) & (iy >= 0) & (iy < n)
    np.add.at(acc, (iy * n + ix)[inside], W[inside])
    mask = (1 << _lib.FIXED_LIMB_BITS) - 1

    def two_limb(name, total):
        acc[n * n + _lib.ACC[name]] = total & mask
        acc[n * n + _lib.ACC_HI[name]] = total >> _lib.FIXED_LIMB_BITS
    two_limb("SUM_WEIGHTS", int(W.astype(object).sum()) if W.size else 0)
    for name, col in (("SUM_X", "pointdataX"), ("SUM_Y", "pointdataY"), ("SUM_R", "pointdataR")):
        v = np.rint(rec[col][ok] * 2.0 ** 32).astype(np.int64)
        two_limb(name, int(v.astype(object).sum()) if v.size else 0)
    acc[n * n + _lib.ACC["N_PASSED"]] = int(ok.sum())
    acc[n * n + _lib.ACC["N_RAYS"]] = rec.size
    acc[n * n + _lib.ACC["N_OUTSIDE_IMAGE"]] = int((~inside).sum())
    return acc


def _value(acc, name, n=256):
    from solaraxionraytracing_amd import _lib
    return (int(acc[n * n + _lib.ACC_HI[name]]) << _lib.FIXED_LIMB_BITS) + int(acc[n * n + _lib.ACC[name]])


def _fixed_worker(rank, world, port, n_total, seed, out_path):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from tests.conftest import make_setup
    from oracle.oracle import Oracle
    D.init_process_group_from_env("gloo")
    o = Oracle(make_setup("babyiaxo_xmm"))

    def trace_fn(offset, n):   # stands in for trace_histogram_device in FIXED64 mode on this rank's GPU
        return torch.from_numpy(_raw_fixed_accumulator(o.trace_records(n, seed=seed, ray_id_offset=offset), -61))

    acc = D.trace_sharded(trace_fn, n_total, rank, world, dst=0, fixed64=True)
    assert acc.dtype == torch.int64
    if rank == 0:
        np.save(out_path, acc.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_fixed64_reduce_is_exact(tmp_path):
    """Integer accumulators: the 2-rank result equals the single-process one bit for bit (the f64 mode above: 1e-12)."""
    from tests.conftest import make_setup
    from oracle.oracle import Oracle
    from solaraxionraytracing_amd import _lib
    n_total, seed = 40_001, 78
    out = str(tmp_path / "acc_fixed.npy")
    mp.spawn(_fixed_worker, args=(2, 29613, n_total, seed, out), nprocs=2, join=True)
    acc = np.load(out)
    one = _raw_fixed_accumulator(Oracle(make_setup("babyiaxo_xmm")).trace_records(n_total, seed=seed), -61)
    n_img = 256 * 256
    assert np.array_equal(acc[:n_img], one[:n_img]) and acc[:n_img].sum() > 0
    for name in ("SUM_WEIGHTS", "SUM_X", "SUM_Y", "SUM_R"):
        assert _value(acc, name) == _value(one, name), name       # the limbs of a sum need not be normalised, its value is exact
    for name in ("N_PASSED", "N_RAYS", "N_OUTSIDE_IMAGE"):
        assert acc[n_img + _lib.ACC[name]] == one[n_img + _lib.ACC[name]], name
    assert _value(acc, "SUM_WEIGHTS") == int(acc[:n_img].astype(object).sum())   # image sums to SUM_WEIGHTS exactly
    # f64 tensors that carry int64 bit patterns (a caller that allocated doubles) reduce the same way
    t = torch.from_numpy(one.copy()).view(torch.float64)
    assert D.reduce_accumulator(t, dst=0, fixed64=True) is t


def test_killed_launcher_takes_its_ranks_with_it(tmp_path):
    """SIGTERM to the launching process (a driver's timeout) ends the ranks it started: none stays behind on a GPU."""
    import signal
    import subprocess
    import time
    rank_script = tmp_path / "sleepy_rank.py"
    rank_script.write_text("import os, sys, time\nopen(os.path.join(sys.argv[1], 'pid%s' % os.environ['RANK']), 'w').write(str(os.getpid()))\ntime.sleep(120)\n")
    launcher = tmp_path / "launcher.py"
    launcher.write_text("import sys\nsys.path.insert(0, %r)\nfrom solaraxionraytracing_amd import distributed as D\nD.visible_devices = lambda: 2\n"
                        "raise SystemExit(D.launch_ranks_if_needed(2, %r, [%r]))\n" % (ROOT, str(rank_script), str(tmp_path)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "SART_BENCH_DEVICE")}
    p = subprocess.Popen([sys.executable, str(launcher)], env=dict(env, SART_BENCH_BACKEND="gloo"))
    deadline = time.time() + 60
    while time.time() < deadline and not all((tmp_path / ("pid%d" % r)).exists() for r in (0, 1)):
        time.sleep(0.1)
    pids = [int((tmp_path / ("pid%d" % r)).read_text()) for r in (0, 1)]
    p.send_signal(signal.SIGTERM)
    assert p.wait(timeout=30) == 128 + signal.SIGTERM
    time.sleep(0.5)
    for pid in pids:
        alive = True
        try:
            os.kill(pid, 0)
            # a terminated child that has not been reaped yet shows as a zombie of init for a moment: look at its state
            state = open("/proc/%d/stat" % pid).read().split(")")[-1].split()[0]
            alive = state not in ("Z", "X")
        except (ProcessLookupError, FileNotFoundError):
            alive = False
        assert not alive, pid


# ---- round 4: first contact with an 8-GPU node must end in a diagnosable exit, never in a wait for ever -----------------------

def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "SART_BENCH_DEVICE", "MASTER_PORT")}
    env.update(extra)
    return env


def test_preflight_brings_up_eight_ranks_through_the_launcher():
    """`python bench.py --gpus 8 --preflight`: eight ranks started by the launcher, process group up, one 512 KB reduce, one JSON
    line on stdout; every other line any rank prints arrives on stderr with its rank in front."""
    import json
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--preflight"], env=_clean_env(SART_BENCH_BACKEND="gloo"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]     # (gloo itself prints a "[Gloo] Rank 0 is connected" line)
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["preflight"] == "ok" and d["world_size"] == 8 and d["backend"] == "gloo" and d["reduce_bytes"] == 512 * 1024 and d["reduce_ms"] > 0
    for l in r.stderr.splitlines():
        assert l.startswith("[rank ") or l.startswith("launcher:"), l


_STUCK_SCRIPT = '''
import os, sys, time
sys.path.insert(0, %r)
from solaraxionraytracing_amd import distributed as D
mode, stuck = sys.argv[1], int(sys.argv[2])
if mode == "before" and int(os.environ["RANK"]) == stuck:
    time.sleep(300)                      # never reaches the rendezvous
if mode == "inside":
    D.report_stage("rendezvous")         # as if hanging inside init_process_group without ever raising
    time.sleep(300)
print("rank %%s at the rendezvous" %% os.environ["RANK"], file=sys.stderr, flush=True)
D.init_process_group_from_env("gloo")
'''


@pytest.mark.parametrize("mode", ["before", "inside"])
def test_a_rank_stuck_at_the_rendezvous_ends_the_launch_with_a_nonzero_exit(tmp_path, mode):
    """init_process_group had no timeout and the launcher waited for ever.  `before`: one rank never reaches the
    rendezvous - the others' init_process_group raises after SART_RDZV_TIMEOUT and the launcher ends the straggler;
    `inside`: every rank hangs in the bring-up without raising - the launcher's own clock ends them and names the ranks."""
    import subprocess
    import time
    script = tmp_path / "stuck.py"
    script.write_text(_STUCK_SCRIPT % ROOT)
    launcher = tmp_path / "launcher.py"
    launcher.write_text("import sys\nsys.path.insert(0, %r)\nfrom solaraxionraytracing_amd import distributed as D\nD.visible_devices = lambda: 3\n"
                        "raise SystemExit(D.launch_ranks_if_needed(3, %r, [%r, '1']))\n" % (ROOT, str(script), mode))
    t0 = time.time()
    r = subprocess.run([sys.executable, str(launcher)], env=_clean_env(SART_BENCH_BACKEND="gloo", SART_RDZV_TIMEOUT="4", SART_RDZV_MARGIN="2"),
                       capture_output=True, text=True, timeout=400)
    assert r.returncode != 0, (r.stdout, r.stderr)
    assert time.time() - t0 < 300      # (seconds here; the margin is for a cold `import torch` in every rank)
    if mode == "inside":
        assert r.returncode == 3 and "never reported their process group up" in r.stderr and "rank 0: rendezvous" in r.stderr
    else:
        assert "[rank 0] " in r.stderr or "[rank 2] " in r.stderr      # the ranks that waited say why they gave up


def test_report_stage_is_silent_without_a_launcher(tmp_path, monkeypatch):
    monkeypatch.delenv("SART_RANK_STATUS_DIR", raising=False)
    D.report_stage("up")                                               # no directory: nothing happens
    monkeypatch.setenv("SART_RANK_STATUS_DIR", str(tmp_path))
    monkeypatch.setenv("RANK", "5")
    D.report_stage("up")
    assert (tmp_path / "rank5").read_text() == "up" and D._rank_stages(str(tmp_path), 6)[5] == "up" and D._rank_stages(str(tmp_path), 6)[0] == "nothing"


# ---- the fused mass scan's multi-rank shape: every rank its share of the ray ids for ALL masses, one reduce of (K + 1) rows ----

_SCAN_MASSES = (0.0, 0.008235, 0.02)


def _scan_rows(o, full, lo, n, seed):
    """What sart_trace_mass_scan_device leaves in a rank's scan accumulator for the ray ids [lo, lo + n), rebuilt from the CPU
    oracle (one oracle run per mass on the same ray ids; the mass-independent counters from the first)."""
    from solaraxionraytracing_amd import _lib
    rows = np.zeros((len(_SCAN_MASSES) + 1, _lib.SCAN_ROW))
    for k, m in enumerate(_SCAN_MASSES):
        s = full.setup.copy()
        s.m_axion = m
        summ = o.trace_histogram(n, seed=seed, ray_id_offset=lo, setup=s, n_threads=2)[1]
        rows[k, _lib.SCAN["SUM_WEIGHTS"]] = summ["SUM_WEIGHTS"]
        rows[k, _lib.SCAN["SUM_WEIGHTS_SQ"]] = summ["SUM_WEIGHTS_SQ"]
        rows[k, _lib.SCAN["N_PASSED"]] = summ["N_PASSED"]
        if k == 0:
            for name in ("N_RAYS", "N_REACHED_TELESCOPE", "N_SHELL_SELECTED", "N_HIT_NICKEL"):
                rows[len(_SCAN_MASSES), _lib.SCAN_SHARED[name]] = summ[name]
    return rows


def _scan_worker(rank, world, port, n_total, seed, out_path):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from tests.conftest import make_setup
    from oracle.oracle import Oracle
    D.init_process_group_from_env("gloo")
    full = make_setup("babyiaxo_xmm_gas")
    o = Oracle(full)
    acc = D.trace_sharded(lambda lo, n: _scan_rows(o, full, lo, n, seed).ravel(), n_total, rank, world, dst=0)
    if rank == 0:
        np.save(out_path, acc.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_mass_scan_reduce_equals_single_process(tmp_path):
    """tools/scan.py mass / bench.py --workload babyiaxo_xmm_gas_scan32 on N ranks: shard the ray ids, trace every mass on the
    shard, ONE reduce of the (K + 1) x 8 scan accumulator.  Counts exact, sums to 1e-12 of a single process; the curve peaks
    on the resonance."""
    import solaraxionraytracing_amd as sa
    from tests.conftest import make_setup
    from oracle.oracle import Oracle
    n_total, seed = 20_001, 79
    out = str(tmp_path / "scan.npy")
    mp.spawn(_scan_worker, args=(2, 29617, n_total, seed, out), nprocs=2, join=True)
    full = make_setup("babyiaxo_xmm_gas")
    one = _scan_rows(Oracle(full), full, 0, n_total, seed)
    per_mass, shared = sa.split_mass_scan(np.load(out), len(_SCAN_MASSES))
    ref_mass, ref_shared = sa.split_mass_scan(one.ravel(), len(_SCAN_MASSES))
    assert shared == ref_shared and shared["N_RAYS"] == n_total
    assert np.array_equal(per_mass["N_PASSED"], ref_mass["N_PASSED"]) and per_mass["N_PASSED"].min() > 1000
    np.testing.assert_allclose(per_mass["SUM_WEIGHTS"], ref_mass["SUM_WEIGHTS"], rtol=1e-12)
    np.testing.assert_allclose(per_mass["SUM_WEIGHTS_SQ"], ref_mass["SUM_WEIGHTS_SQ"], rtol=1e-12)
    assert int(np.argmax(per_mass["SUM_WEIGHTS"])) == 1
    assert sa.mass_scan_len(len(_SCAN_MASSES)) == one.size


_FAIL_AFTER_REDUCE_SCRIPT = '''
import os, sys, time
sys.path.insert(0, %r)
import torch
import torch.distributed as dist
from solaraxionraytracing_amd import distributed as D
rank, world, _ = D.init_process_group_from_env("gloo")
acc = torch.ones(1000, dtype=torch.float64)
D.reduce_accumulator(acc, dst=0)          # the path's one collective
dist.barrier()                            # the barrier that closes bench.py's timed region
if rank == 0:
    if sys.argv[1] == "raise":
        raise RuntimeError("rank 0 fails behind the reduce")       # (bench.py: a FIXED64 status surfacing in rt.synchronize())
    os._exit(7)
times = torch.zeros(2, dtype=torch.float64)
if sys.argv[2] == "deaf":
    import signal
    signal.signal(signal.SIGTERM, signal.SIG_IGN)    # a rank that SIGTERM does not end (as if blocked in a driver call)
    try:
        dist.all_reduce(times, op=dist.ReduceOp.MAX)
    except Exception:
        pass
    time.sleep(300)
dist.all_reduce(times, op=dist.ReduceOp.MAX)   # the other ranks wait here for a rank that is gone
'''


@pytest.mark.parametrize("how,deaf", [("raise", "no"), ("exit", "no"), ("raise", "deaf")])
def test_rank0_failing_behind_the_reduce_ends_the_launch_and_is_named(tmp_path, how, deaf):
    """VERDICT r04 (5) + ADVICE r04: only rank 0 finalizes and can raise behind the reduce; the other ranks then wait in the next
    collective.  The launcher must come back within seconds with rank 0's exit code, say which rank failed and that it ended the
    others - and, when a rank does not die on SIGTERM, kill it after the grace period instead of waiting for ever."""
    import subprocess
    import time
    script = tmp_path / "fail.py"
    script.write_text(_FAIL_AFTER_REDUCE_SCRIPT % ROOT)
    launcher = tmp_path / "launcher.py"
    launcher.write_text("import sys\nsys.path.insert(0, %r)\nfrom solaraxionraytracing_amd import distributed as D\nD.visible_devices = lambda: 3\n"
                        "raise SystemExit(D.launch_ranks_if_needed(3, %r, [%r, %r]))\n" % (ROOT, str(script), how, deaf))
    t0 = time.time()
    r = subprocess.run([sys.executable, str(launcher)], env=_clean_env(SART_BENCH_BACKEND="gloo", SART_TERM_GRACE="3"), capture_output=True, text=True,
                       timeout=400)
    dt = time.time() - t0
    assert r.returncode == (1 if how == "raise" else 7), (r.returncode, r.stderr[-1500:])
    assert "launcher: rank 0 exited with code %d; ending rank(s) 1, 2" % r.returncode in r.stderr, r.stderr[-1500:]
    if how == "raise":
        assert any(l.startswith("[rank 0]") and "RuntimeError: rank 0 fails behind the reduce" in l for l in r.stderr.splitlines()), r.stderr[-1500:]
    assert dt < 200      # seconds (the margin is a cold `import torch` in every rank); gloo's own collective timeout is 30 min
    if deaf == "deaf":
        assert "launcher: rank(s) 1, 2 still alive 3 s after SIGTERM: killed" in r.stderr, r.stderr[-1500:]


_STALL_SCRIPT = '''
import os, sys, time
sys.path.insert(0, %r)
from solaraxionraytracing_amd import distributed as D
rank, world, _ = D.init_process_group_from_env("gloo")
mode = sys.argv[1]
for k in range(3):
    if mode != "silent":
        D.heartbeat("step %%d" %% k)
    time.sleep(0.3)
if mode == "hang":
    D.heartbeat("reduce")
    time.sleep(300)                      # every rank alive, nobody makes progress: a collective that never completes
if mode == "slow_rank0" and rank == 0:
    for k in range(8):                   # rank 0 works on (and says so) while the others wait for it: not a stall
        time.sleep(1.0)
        D.heartbeat("side work %%d" %% k)
if mode == "silent":
    time.sleep(6)                        # a script that never reports progress is not watched
import torch.distributed as dist
dist.barrier()
D.report_stage("done")
'''


@pytest.mark.parametrize("mode", ["hang", "slow_rank0", "silent"])
def test_stall_clock_ends_a_run_whose_ranks_all_fall_silent(tmp_path, mode):
    """with the wall limit opt-in, a run that hangs with every rank alive (a stuck collective) was never ended by the
    launcher.  Ranks that report progress (heartbeat) get a stall clock: all of them silent for SART_STALL_TIMEOUT -> exit 3 and
    the last report of each rank; one rank that keeps reporting keeps the run alive; a script without heartbeats is not watched."""
    import subprocess
    import time
    script = tmp_path / "stall.py"
    script.write_text(_STALL_SCRIPT % ROOT)
    launcher = tmp_path / "launcher.py"
    launcher.write_text("import sys\nsys.path.insert(0, %r)\nfrom solaraxionraytracing_amd import distributed as D\nD.visible_devices = lambda: 2\n"
                        "raise SystemExit(D.launch_ranks_if_needed(2, %r, [%r]))\n" % (ROOT, str(script), mode))
    t0 = time.time()
    r = subprocess.run([sys.executable, str(launcher)], env=_clean_env(SART_BENCH_BACKEND="gloo", SART_STALL_TIMEOUT="3"), capture_output=True,
                       text=True, timeout=280)
    if mode == "hang":
        assert r.returncode == 3 and "SART_STALL_TIMEOUT" in r.stderr and "rank 0: beat:reduce" in r.stderr and "rank 1: beat:reduce" in r.stderr, (
            r.returncode, r.stderr[-1500:])
        assert time.time() - t0 < 250
    else:
        assert r.returncode == 0, (r.returncode, r.stderr[-1500:])
    from solaraxionraytracing_amd import distributed as D
    assert D.STALL_TIMEOUT_S == 900


def test_a_wall_limit_is_opt_in(tmp_path):
    """the launcher used to end any self-launched run after 25 minutes.  Now only when SART_LAUNCH_TIMEOUT asks."""
    import subprocess
    script = tmp_path / "slow.py"
    script.write_text("import time\ntime.sleep(4)\n")
    launcher = tmp_path / "launcher.py"
    launcher.write_text("import sys\nsys.path.insert(0, %r)\nfrom solaraxionraytracing_amd import distributed as D\nD.visible_devices = lambda: 2\n"
                        "raise SystemExit(D.launch_ranks_if_needed(2, %r, []))\n" % (ROOT, str(script)))
    r = subprocess.run([sys.executable, str(launcher)], env=_clean_env(SART_BENCH_BACKEND="gloo"), capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([sys.executable, str(launcher)], env=_clean_env(SART_BENCH_BACKEND="gloo", SART_LAUNCH_TIMEOUT="1"), capture_output=True, text=True,
                       timeout=120)
    assert r.returncode == 3 and "SART_LAUNCH_TIMEOUT" in r.stderr, (r.returncode, r.stderr)
    from solaraxionraytracing_amd import distributed as D
    assert D.LAUNCH_TIMEOUT_S == 0


# ---- fused angular scan on N ranks (BASELINE configs[3]: "8 GPUs sharded by angle bin"), rebuilt from the CPU oracle -----------------
_SCAN_ANGLES = (0.0, 0.05, 0.1, 0.15, 0.2)


def _ascan_rows(o, full, angles, lo, n, seed):
    """What sart_trace_angular_scan_device leaves in a rank's scan accumulator for `angles` on the ray ids [lo, lo + n): one oracle
    run per angle on the same ray ids; the angle-independent counters from the first."""
    from solaraxionraytracing_amd import _lib
    rows = np.zeros((len(angles) + 1, _lib.ASCAN_ROW))
    for k, a in enumerate(angles):
        s = full.setup.copy()
        s.telescope_turned_y_deg = float(a)
        summ = o.trace_histogram(n, seed=seed, ray_id_offset=lo, setup=s, flags=full.flags, n_threads=2)[1]
        for name, slot in _lib.ASCAN.items():
            rows[k, slot] = summ[name]
        if k == 0:
            for name, slot in _lib.ASCAN_SHARED.items():
                rows[len(angles), slot] = summ[name]
    return rows


def _ascan_worker(rank, world, port, n_total, seed, shard, out_path):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import torch
    from tests.conftest import make_setup
    from oracle.oracle import Oracle
    import solaraxionraytracing_amd as sa
    D.init_process_group_from_env("gloo")
    full = make_setup("babyiaxo_xmm_rot")
    o = Oracle(full)
    if shard == "rays":      # every rank turns its share of the ray ids through all angles; one reduce of (K + 1) x 8 slots
        acc = D.trace_sharded(lambda lo, n: _ascan_rows(o, full, _SCAN_ANGLES, lo, n, seed).ravel(), n_total, rank, world, dst=0).numpy()
        curve = sa.split_angular_scan(acc, len(_SCAN_ANGLES))[0]["SUM_WEIGHTS"]
    else:                    # the angles are dealt out (tools/scan.py angular --fused): every rank on ALL ray ids for its group
        mine = D.shard_angles(len(_SCAN_ANGLES), rank, world)
        rows = _ascan_rows(o, full, [_SCAN_ANGLES[i] for i in mine], 0, n_total, seed)
        vals = sa.split_angular_scan(rows.ravel(), len(mine))[0]["SUM_WEIGHTS"]
        curve = D.gather_scan(torch.tensor(vals, dtype=torch.float64), mine, len(_SCAN_ANGLES)).numpy()
    if rank == 0:
        np.save(out_path, curve)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("shard,world", [("rays", 2), ("bins", 2), ("bins", 3)])
def test_fused_angular_scan_on_n_ranks_equals_single_process(tmp_path, shard, world):
    """The two shapes of the fused angular scan on N ranks (tools/scan.py angular --fused [--shard rays]): the curve of one
    process, whatever the number of ranks - angle groups: every rank sees all ray ids, so its rows ARE the single process's rows;
    ray shards: counts exact, sums to 1e-12."""
    import solaraxionraytracing_amd as sa
    from tests.conftest import make_setup
    from oracle.oracle import Oracle
    n_total, seed = 20_001, 83
    out = str(tmp_path / "ascan.npy")
    mp.spawn(_ascan_worker, args=(world, 29631 + world + (7 if shard == "rays" else 0), n_total, seed, shard, out), nprocs=world, join=True)
    full = make_setup("babyiaxo_xmm_rot")
    one = sa.split_angular_scan(_ascan_rows(Oracle(full), full, _SCAN_ANGLES, 0, n_total, seed).ravel(), len(_SCAN_ANGLES))[0]["SUM_WEIGHTS"]
    got = np.load(out)
    if shard == "bins":
        assert np.array_equal(got, one)
    else:
        np.testing.assert_allclose(got, one, rtol=1e-12)
    assert got.min() > 0 and got[0] > got[-1]
    assert sa.angular_scan_len(len(_SCAN_ANGLES)) == (len(_SCAN_ANGLES) + 1) * 8
