"""The rank launcher of bench.py --gpus N (multigridsolver_amd/launch.py) on CPU: both ways in (plain `python script --gpus N`
spawning its own ranks; per-rank supervisors under torch.distributed.run), the generation fallback after a failed or hung
attempt, and the done-marker that keeps a teardown crash from restarting anything.  gloo, world 2."""
import json
import os
import subprocess
import sys

import pytest

from conftest import REPO

DUMMY = os.path.join(REPO, "tests", "launch_dummy.py")


def _run(cmd, extra_env=None, timeout=300):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    for k in ("MGS_BENCH_WORKER", "MGS_BENCH_GEN", "RANK", "WORLD_SIZE", "LOCAL_RANK", "MGS_NATIVE_RCCL", "MGS_NATIVE_GRAPH", "MGS_DIST_BACKEND"):
        env.pop(k, None)
    env.update(extra_env or {})
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=REPO)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return r, [json.loads(l) for l in lines]


@pytest.mark.parametrize("ok_from,mode", [(0, "exit"), (3, "exit"), (1, "hang"), (0, "crash_after_done"), (0, "die_before_print")])
def test_spawn_ranks_generations(ok_from, mode):
    r, out = _run([sys.executable, DUMMY, str(ok_from), mode, "2"], {"MGS_BENCH_WATCHDOG_S": "4"})
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert len(out) == 1 and out[0]["generation"] == ok_from and out[0]["sum"] == 1.0 and out[0]["world"] == 2, out
    # a run that completed in a later generation says so, with the reason every earlier generation was abandoned for
    assert out[0]["degraded"] == (ok_from > 0) and len(out[0]["abandoned_generations"]) == ok_from, out
    assert all(a["reason"] for a in out[0]["abandoned_generations"])
    if ok_from == 0:
        assert out[0]["native"] == "1" and out[0]["graph"] == "1"
    if ok_from == 3:
        assert out[0]["native"] == "0" and out[0]["name"] == "torch.distributed-callbacks"


def test_spawn_ranks_gives_up_with_the_workers_code():
    r, out = _run([sys.executable, DUMMY, "99", "exit", "2"])
    assert r.returncode != 0 and out == []


@pytest.mark.parametrize("ok_from,mode", [(0, "exit"), (1, "exit"), (1, "hang"), (0, "die_before_print")])
def test_supervisors_under_torchrun(ok_from, mode):
    port = 29300 + (os.getpid() % 500)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           DUMMY, str(ok_from), mode, "2"]
    r, out = _run(cmd, {"MGS_BENCH_WATCHDOG_S": "4"})
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert len(out) == 1 and out[0]["generation"] == ok_from and out[0]["sum"] == 1.0, out
    assert out[0]["degraded"] == (ok_from > 0) and len(out[0]["abandoned_generations"]) == ok_from, out


def test_supervisors_change_generation_together():
    """One rank's worker fails at once, the other hangs, and the watchdog is as long as in production (150 s): the supervisors
    share the run directory, the hanging worker is ended the moment the failure is posted, and both ranks meet in the next
    generation long before any watchdog would have fired — they cannot end up in different generations."""
    import time
    port = 29300 + ((os.getpid() + 250) % 500)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           DUMMY, "1", "hang", "2"]
    t0 = time.monotonic()
    r, out = _run(cmd, {"MGS_BENCH_WATCHDOG_S": "150"}, timeout=140)
    dt = time.monotonic() - t0
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert len(out) == 1 and out[0]["generation"] == 1 and out[0]["sum"] == 1.0, out
    assert dt < 90, f"generation change waited for a watchdog ({dt:.0f} s)"
    assert out[0]["abandoned_generations"][0]["generation"] == "native-p2p+graph" and "exited with" in out[0]["abandoned_generations"][0]["reason"]


def test_first_generation_honours_explicit_choices():
    from multigridsolver_amd import launch
    names = [g[0] for g in launch.GENERATIONS]
    assert names == ["native-p2p+graph", "native-rccl+graph", "native-rccl", "torch.distributed-callbacks", "gloo-host-staged"]
    assert launch.first_generation({}) == 0
    assert launch.first_generation({"MGS_NATIVE_TRANSPORT": "rccl"}) == 1
    assert launch.first_generation({"MGS_NATIVE_GRAPH": "0"}) == 2
    assert launch.first_generation({"MGS_NATIVE_RCCL": "0"}) == 3
    assert launch.first_generation({"MGS_DIST_BACKEND": "gloo"}) == 4
    assert launch.first_generation({"MGS_DIST_BACKEND": "gloo", "MGS_NATIVE_RCCL": "force"}) == 2
    assert launch.first_generation({"MGS_DIST_BACKEND": "gloo", "MGS_NATIVE_TRANSPORT": "p2p"}) == 0
    # generation 0 tries the peer-to-peer transport, then RCCL, inside one generation; the RCCL generations restrict themselves to RCCL
    assert launch.GENERATIONS[0][1]["MGS_NATIVE_TRANSPORT"] == "p2p,rccl" and launch.GENERATIONS[1][1]["MGS_NATIVE_TRANSPORT"] == "rccl"


@pytest.mark.parametrize("launcher", ["spawn", "torchrun"])
def test_time_budget_skips_to_the_last_generation(launcher):
    """once the run's time budget is spent (here: 0 s) a failed generation is followed by the LAST one, on every rank together —
    the driver's bench limit must not be eaten by generation after generation"""
    port = 29300 + ((os.getpid() + 123) % 500)
    cmd = [sys.executable, DUMMY, "4", "exit", "2"] if launcher == "spawn" else \
          [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", str(port), DUMMY, "4", "exit", "2"]
    r, out = _run(cmd, {"MGS_BENCH_WATCHDOG_S": "4", "MGS_BENCH_BUDGET_S": "0"})
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert len(out) == 1 and out[0]["generation"] == 4 and out[0]["name"] == "gloo-host-staged", out
    assert [a["generation"] for a in out[0]["abandoned_generations"]] == ["native-p2p+graph"], out


def test_run_directory_is_unique_and_removed():
    """the supervisors' shared directory is named after the launcher's pid AND start time (a recycled pid/port pair of an earlier run cannot
    collide) and is gone after a successful run"""
    import glob
    import tempfile
    from multigridsolver_amd import launch
    assert launch._proc_start(os.getpid()) > 0
    port = 29300 + ((os.getpid() + 321) % 500)
    before = set(glob.glob(os.path.join(tempfile.gettempdir(), f"mgs_bench_{port}_*")))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", str(port), DUMMY, "1", "exit", "2"]
    r, out = _run(cmd, {"MGS_BENCH_WATCHDOG_S": "4"})
    assert r.returncode == 0 and len(out) == 1 and out[0]["generation"] == 1, r.stderr[-3000:]
    assert set(glob.glob(os.path.join(tempfile.gettempdir(), f"mgs_bench_{port}_*"))) == before
