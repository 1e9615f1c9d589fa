"""Rank launcher / supervisor of the multi-GPU runs (bench.py --gpus N, one process per GPU).

Two ways in, one protocol:
  * `python bench.py --gpus N` (no launcher): `spawn_ranks` starts N rank workers itself as CHILD processes — before this
    process makes any GPU/HIP call, never by exec — and relays their exit;
  * `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`: every launched process is a thin per-rank
    supervisor (`supervise_rank`) that starts its worker as a child.

A worker is the same script with MGS_BENCH_WORKER=1.  Workers are started in GENERATIONS of decreasing ambition
(`GENERATIONS`): the native RCCL transport with the cycle captured in a hipGraph first, the torch.distributed callback
path next, host-staged gloo last.  A generation ends for everyone when a worker exits non-zero or its watchdog
(`Watchdog`, no heartbeat for `MGS_BENCH_WATCHDOG_S` seconds → exit code 77) fires: a hang inside a collective that has
never run on this machine costs one watchdog period instead of the whole run.  A worker that finished its part writes a
marker file, so a crash during teardown does not restart anything; rank 0 persists its JSON line in the run directory BEFORE it
tears anything down, and its supervisor prints that file if the worker dies before printing it itself.  The change of generation
is COLLECTIVE in both entry points: under torch.distributed.run the per-rank supervisors share the run directory, the one
whose worker failed drops `fail.g<gen>` there, every supervisor polls for it, ends its own worker (exact pid) and all start
generation g+1 together.  A run that completed in a later generation says so in its JSON line (`"degraded": true` and the
reasons the earlier generations were abandoned for).

Nothing here touches the GPU or imports torch.
"""
import os
import signal
import socket
import subprocess
import sys
import tempfile
import threading
import time

EXIT_RETRY = 77
# (name, environment overrides) — most capable first
GENERATIONS = (
    ("native-rccl+graph", {"MGS_NATIVE_RCCL": "1", "MGS_NATIVE_GRAPH": "1"}),
    ("native-rccl", {"MGS_NATIVE_RCCL": "1", "MGS_NATIVE_GRAPH": "0"}),
    ("torch.distributed-callbacks", {"MGS_NATIVE_RCCL": "0", "MGS_NATIVE_GRAPH": "0"}),
    ("gloo-host-staged", {"MGS_NATIVE_RCCL": "0", "MGS_NATIVE_GRAPH": "0", "MGS_DIST_BACKEND": "gloo"}),
)


def first_generation(env=None):
    """the generation a run starts in: an explicit MGS_NATIVE_RCCL=0 / MGS_DIST_BACKEND=gloo skips what it rules out"""
    env = os.environ if env is None else env
    if env.get("MGS_DIST_BACKEND") == "gloo":
        return 3 if env.get("MGS_NATIVE_RCCL") != "force" else 1
    if env.get("MGS_NATIVE_RCCL") == "0":
        return 2
    if env.get("MGS_NATIVE_GRAPH") == "0":
        return 1
    return 0


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class Watchdog:
    """worker side: `beat(phase)` at every milestone; no beat for `limit` seconds → diagnostic + os._exit(77)"""

    def __init__(self, limit=None, out=sys.stderr):
        self.limit = float(os.environ.get("MGS_BENCH_WATCHDOG_S", "150")) if limit is None else float(limit)
        self.phase, self.t, self.out, self._stop = "start", time.monotonic(), out, False
        self.th = threading.Thread(target=self._run, daemon=True)
        self.th.start()

    def beat(self, phase):
        self.phase, self.t = phase, time.monotonic()

    def stop(self):
        self._stop = True

    def _run(self):
        while not self._stop:
            time.sleep(1.0)
            if time.monotonic() - self.t > self.limit:
                try:
                    print(f"[watchdog] rank {os.environ.get('RANK', '0')}: no progress for {self.limit:.0f}s in phase '{self.phase}' "
                          f"(generation {os.environ.get('MGS_BENCH_GEN', '0')}) -> exit {EXIT_RETRY}", file=self.out, flush=True)
                finally:
                    os._exit(EXIT_RETRY)


def _marker(rundir, gen, rank):
    return os.path.join(rundir, f"done.g{gen}.r{rank}")


def _fail_file(rundir, gen):
    return os.path.join(rundir, f"fail.g{gen}")


def _result_file(rundir, gen):
    return os.path.join(rundir, f"result.g{gen}.json")


def mark_done():
    """worker side: this rank's part of the generation is complete (the JSON line, if any, is out or persisted)"""
    d = os.environ.get("MGS_BENCH_RUNDIR")
    if d:
        try:
            with open(_marker(d, os.environ.get("MGS_BENCH_GEN", "0"), os.environ.get("RANK", "0")), "w") as f:
                f.write("ok\n")
        except OSError:
            pass


def persist_result(line):
    """worker side (rank 0): the finished measurement, written before any teardown; the supervisor prints it if this worker
    dies before it printed the line itself"""
    d = os.environ.get("MGS_BENCH_RUNDIR")
    if d:
        try:
            path = _result_file(d, os.environ.get("MGS_BENCH_GEN", "0"))
            with open(path + ".tmp", "w") as f:
                f.write(line)
            os.replace(path + ".tmp", path)
        except OSError:
            pass


def mark_printed():
    """worker side (rank 0): the JSON line went out on stdout — nothing left for the supervisor to relay"""
    d = os.environ.get("MGS_BENCH_RUNDIR")
    if d:
        try:
            os.unlink(_result_file(d, os.environ.get("MGS_BENCH_GEN", "0")))
        except OSError:
            pass


def _relay_unprinted(rundir, gen, out=None):
    """supervisor side: rank 0's worker is gone; a result file still there was never printed"""
    try:
        path = _result_file(rundir, gen)
        with open(path) as f:
            line = f.read().strip()
        os.unlink(path)
    except OSError:
        return False
    if line:
        (out or sys.stdout).write(line + "\n"); (out or sys.stdout).flush()
    return bool(line)


def abandoned_generations():
    """worker side: [{generation, reason}] of the generations this run gave up before the current one (empty: first attempt)"""
    import json
    try:
        return json.loads(os.environ.get("MGS_BENCH_ABANDONED", "[]"))
    except ValueError:
        return []


def _worker_env(base, gen, rank, local_rank, world, addr, port, rundir):
    env = dict(base)
    env.update(GENERATIONS[gen][1])
    if base.get("MGS_NATIVE_RCCL") == "force" and GENERATIONS[gen][1].get("MGS_NATIVE_RCCL") == "1":
        env["MGS_NATIVE_RCCL"] = "force"          # tests: stand-in RCCL without the nccl backend
    if base.get("MGS_DIST_BACKEND") and "MGS_DIST_BACKEND" not in GENERATIONS[gen][1]:
        env["MGS_DIST_BACKEND"] = base["MGS_DIST_BACKEND"]
    env.update(MGS_BENCH_WORKER="1", MGS_BENCH_GEN=str(gen), MGS_BENCH_GEN_NAME=GENERATIONS[gen][0], MGS_BENCH_RUNDIR=rundir,
               RANK=str(rank), LOCAL_RANK=str(local_rank), WORLD_SIZE=str(world), MASTER_ADDR=addr, MASTER_PORT=str(port),
               HSA_ENABLE_IPC_MODE_LEGACY=base.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return env


def _kill(p):
    """exact pid only (never a pattern): TERM, then KILL"""
    if p.poll() is None:
        try:
            p.send_signal(signal.SIGTERM)
            p.wait(timeout=10)
        except Exception:  # noqa: BLE001
            try:
                p.kill(); p.wait(timeout=10)
            except Exception:  # noqa: BLE001
                pass


def spawn_ranks(argv, world, log=lambda *a: None, total_timeout=None):
    """`python bench.py --gpus N` without a launcher: N workers as child processes, generation after generation.
    Returns the exit code for the caller to exit with (0 = some generation completed on every rank)."""
    import json
    rundir = tempfile.mkdtemp(prefix="mgs_bench_")
    limit = float(os.environ.get("MGS_BENCH_GEN_TIMEOUT_S", "1500")) if total_timeout is None else total_timeout
    rc_final = 1
    abandoned = []
    for gen in range(first_generation(), len(GENERATIONS)):
        port = free_port()
        base = dict(os.environ, MGS_BENCH_ABANDONED=json.dumps(abandoned))
        procs = [subprocess.Popen([sys.executable] + argv, env=_worker_env(base, gen, r, r, world, "127.0.0.1", port, rundir))
                 for r in range(world)]
        log(f"generation {gen} ({GENERATIONS[gen][0]}): started {world} rank workers, rendezvous 127.0.0.1:{port}")
        t0, bad = time.monotonic(), None
        while True:
            codes = [p.poll() for p in procs]
            done = [c == 0 or (c is not None and os.path.exists(_marker(rundir, gen, r))) for r, c in enumerate(codes)]
            if all(done):
                _relay_unprinted(rundir, gen)            # rank 0 finished its measurement but died before printing it
                return 0
            failed = [r for r, c in enumerate(codes) if c is not None and not done[r]]
            if failed:
                bad = f"rank {failed[0]} exited with {codes[failed[0]]}"
                break
            if time.monotonic() - t0 > limit:
                bad = f"no completion within {limit:.0f}s"
                break
            time.sleep(0.2)
        for p in procs:
            _kill(p)
        rc_final = next((c for c in (p.returncode for p in procs) if c not in (0, None)), 1)
        abandoned.append({"generation": GENERATIONS[gen][0], "reason": bad})
        log(f"generation {gen} ({GENERATIONS[gen][0]}) abandoned: {bad}")
    return rc_final


def supervise_rank(argv, log=lambda *a: None):
    """one torch.distributed.run worker slot: start this rank's worker as a child, next generation on failure.
    Generation 0 meets on the launcher's own rendezvous (env://); later generations on MASTER_PORT + 1 + gen, hosted by
    rank 0's worker, so nothing of an abandoned generation is reused.  The supervisors of one node share `rundir`: the one
    whose worker fails writes fail.g<gen> (with the reason), all of them poll for it and leave the generation TOGETHER —
    no rank waits in the rendezvous of generation g+1 while another still sits out its watchdog in generation g."""
    import json
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    addr, port0 = os.environ.get("MASTER_ADDR", "127.0.0.1"), int(os.environ.get("MASTER_PORT", "29500"))
    # the launcher agent is the common parent of the rank slots: its pid makes the directory unique to this run
    rundir = os.path.join(tempfile.gettempdir(), f"mgs_bench_{port0}_{os.getppid()}")
    os.makedirs(rundir, exist_ok=True)
    for gen in range(len(GENERATIONS)):               # leftovers of an earlier run with the same pid and port (own files only:
        for path in (_marker(rundir, gen, rank),):    # the shared fail/result files are keyed by generation and cleared by rank 0 below)
            try:
                os.unlink(path)
            except OSError:
                pass
    limit = float(os.environ.get("MGS_BENCH_GEN_TIMEOUT_S", "1500"))
    rc = 1
    g0 = first_generation()
    abandoned = []
    for gen in range(g0, len(GENERATIONS)):
        base = dict(os.environ, MGS_BENCH_ABANDONED=json.dumps(abandoned))
        env = _worker_env(base, gen, rank, local_rank, world, addr, port0 if gen == g0 else port0 + 1 + gen, rundir)
        if gen != g0:
            env["MGS_BENCH_OWN_STORE"] = "1"      # rank 0's worker hosts the TCPStore of this generation
        p = subprocess.Popen([sys.executable] + argv, env=env)
        t0, reason = time.monotonic(), None
        while True:
            rc = p.poll()
            if rc is not None:
                if rc == 0 or os.path.exists(_marker(rundir, gen, rank)):
                    if rank == 0:
                        _relay_unprinted(rundir, gen)
                    return 0
                reason = f"rank {rank} worker exited with {rc}"
                break
            if os.path.exists(_fail_file(rundir, gen)):          # another rank's worker failed: leave this generation with it
                try:
                    reason = open(_fail_file(rundir, gen)).read().strip() or "another rank failed"
                except OSError:
                    reason = "another rank failed"
                if os.path.exists(_marker(rundir, gen, rank)):   # this rank's part was complete already: nothing to redo here
                    _kill(p)
                    return 0
                _kill(p); rc = EXIT_RETRY
                break
            if time.monotonic() - t0 > limit:
                reason = f"rank {rank}: no completion within {limit:.0f}s"
                _kill(p); rc = EXIT_RETRY
                break
            time.sleep(0.1)
        if not os.path.exists(_fail_file(rundir, gen)):            # first to fail: tell the others (atomic create; losers keep the winner's reason)
            try:
                fd = os.open(_fail_file(rundir, gen), os.O_CREAT | os.O_EXCL | os.O_WRONLY, 0o644)
                os.write(fd, (reason or "failed").encode()); os.close(fd)
            except OSError:
                pass
        try:
            reason = open(_fail_file(rundir, gen)).read().strip() or reason
        except OSError:
            pass
        abandoned.append({"generation": GENERATIONS[gen][0], "reason": reason})
        if rank == 0:
            log(f"generation {gen} ({GENERATIONS[gen][0]}) abandoned: {reason}")
    return rc


def init_process_group(backend):
    """worker side: env:// rendezvous, or this generation's own store (see supervise_rank)"""
    import datetime

    import torch.distributed as dist
    if os.environ.get("MGS_BENCH_OWN_STORE"):
        rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        store = dist.TCPStore(os.environ["MASTER_ADDR"], int(os.environ["MASTER_PORT"]), world, is_master=(rank == 0),
                              timeout=datetime.timedelta(seconds=900), wait_for_workers=False)
        dist.init_process_group(backend=backend, store=store, rank=rank, world_size=world)
    else:
        dist.init_process_group(backend=backend)
