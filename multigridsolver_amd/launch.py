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
import shutil
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
    # the library's own peer-to-peer exchange kernels over IPC-mapped device windows (csrc/comm_p2p.hip), cycle captured in a hipGraph;
    # a wait that never ends times out on the device (MGS_P2P_TIMEOUT_S) and the worker exits non-zero: no watchdog period is spent
    # (a peer-to-peer set-up that fails CLEANLY — window allocation, IPC mapping, self-test, cross-check — is followed by RCCL inside the same generation)
    ("native-p2p+graph", {"MGS_NATIVE_RCCL": "1", "MGS_NATIVE_GRAPH": "1", "MGS_NATIVE_TRANSPORT": "p2p,rccl"}),
    # RCCL send/recv groups inside the captured cycle, in their most conservative form: one packed message per peer (dist.py's default for
    # this transport: the pack-free range sends post several ncclSend/ncclRecv per peer in one group — never run on real RCCL)
    ("native-rccl+graph", {"MGS_NATIVE_RCCL": "1", "MGS_NATIVE_GRAPH": "1", "MGS_NATIVE_TRANSPORT": "rccl"}),
    ("native-rccl", {"MGS_NATIVE_RCCL": "1", "MGS_NATIVE_GRAPH": "0", "MGS_NATIVE_TRANSPORT": "rccl"}),
    ("torch.distributed-callbacks", {"MGS_NATIVE_RCCL": "0", "MGS_NATIVE_GRAPH": "0"}),
    ("gloo-host-staged", {"MGS_NATIVE_RCCL": "0", "MGS_NATIVE_GRAPH": "0", "MGS_DIST_BACKEND": "gloo"}),
)
# Time budget (the driver gives one bench run 600 s): a generation that hangs costs one watchdog period — 100 s by default
# (MGS_BENCH_WATCHDOG_S; the longest silent phase of a healthy run is the first `import torch` on a fresh box, 1-2 min, which happens
# BEFORE the watchdog starts) — and generations are skipped once MGS_BENCH_BUDGET_S (default 420 s) of the run are gone, so that the last
# resort (host-staged gloo) still has time to finish: two failed generations + one good one + the bounded CPU baseline stay under 600 s.
BUDGET_S = float(os.environ.get("MGS_BENCH_BUDGET_S", "420"))


def first_generation(env=None):
    """the generation a run starts in: an explicit MGS_NATIVE_RCCL=0 / MGS_DIST_BACKEND=gloo skips what it rules out"""
    env = os.environ if env is None else env
    if env.get("MGS_DIST_BACKEND") == "gloo":
        if env.get("MGS_NATIVE_TRANSPORT", "").startswith("p2p") and env.get("MGS_NATIVE_RCCL") != "0":
            return 0          # the peer-to-peer transport needs torch.distributed for its setup handshakes only: any backend serves (ranks sharing one GPU: rehearsals)
        return 4 if env.get("MGS_NATIVE_RCCL") != "force" else 2
    if env.get("MGS_NATIVE_RCCL") == "0":
        return 3
    if env.get("MGS_NATIVE_GRAPH") == "0":
        return 2
    if env.get("MGS_NATIVE_TRANSPORT") == "rccl":
        return 1
    return 0


def next_generation(gen, t_start, log=lambda *a: None):
    """the generation to try after `gen` failed: the next one, or — once the run's time budget is spent — the last resort"""
    nxt = gen + 1
    if nxt < len(GENERATIONS) - 1 and time.monotonic() - t_start > BUDGET_S:
        log(f"{time.monotonic() - t_start:.0f}s of the run are gone (budget {BUDGET_S:.0f}s): skipping to the last generation")
        return len(GENERATIONS) - 1
    return nxt


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class Watchdog:
    """worker side: `beat(phase)` at every milestone; no beat for `limit` seconds → diagnostic + os._exit(77)"""

    def __init__(self, limit=None, out=sys.stderr):
        self.limit = float(os.environ.get("MGS_BENCH_WATCHDOG_S", "100")) if limit is None else float(limit)
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


def _proc_start(pid):
    """start time of a process in clock ticks since boot (field 22 of /proc/<pid>/stat); 0 if unreadable"""
    try:
        with open(f"/proc/{pid}/stat") as f:
            return int(f.read().rsplit(")", 1)[1].split()[19])
    except (OSError, ValueError, IndexError):
        return 0


def _proc_start_monotonic(pid):
    """time.monotonic() value at which process `pid` started (CLOCK_MONOTONIC and /proc start times both count from boot)"""
    st = _proc_start(pid)
    if not st:
        return time.monotonic()
    try:
        return st / os.sysconf("SC_CLK_TCK")
    except (ValueError, OSError):
        return time.monotonic()


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
    limit = float(os.environ.get("MGS_BENCH_GEN_TIMEOUT_S", "400")) if total_timeout is None else total_timeout
    rc_final = 1
    abandoned = []
    t_start = time.monotonic()
    gen = first_generation() - 1
    while True:
        gen = next_generation(gen, t_start, log) if gen >= first_generation() else first_generation()
        if gen >= len(GENERATIONS):
            break
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
                shutil.rmtree(rundir, ignore_errors=True)
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
    shutil.rmtree(rundir, ignore_errors=True)
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
    # the launcher agent is the common parent of the rank slots: its pid AND its start time (plus the launcher's run id) make the
    # directory unique to this run — a recycled pid/port pair of an earlier run in the same container names another directory, so no
    # stale fail.g* / result.g* / done.g* file can ever be seen (advisor, round 3)
    rundir = os.path.join(tempfile.gettempdir(), f"mgs_bench_{port0}_{os.getppid()}_{_proc_start(os.getppid())}_{os.environ.get('TORCHELASTIC_RUN_ID', 'x')}")
    os.makedirs(rundir, exist_ok=True)
    limit = float(os.environ.get("MGS_BENCH_GEN_TIMEOUT_S", "400"))
    rc = 1
    g0 = first_generation()
    abandoned = []
    t_start = _proc_start_monotonic(os.getppid())      # run time counts from the launcher's start

    def leave(code):
        """this slot's files go; the last slot out removes the directory (rank 0 also drops the shared files once the run succeeded)"""
        for g in range(len(GENERATIONS)):
            for path in [_marker(rundir, g, rank)] + ([_fail_file(rundir, g), _result_file(rundir, g)] if (rank == 0 and code == 0) else []):
                try:
                    os.unlink(path)
                except OSError:
                    pass
        try:
            os.rmdir(rundir)
        except OSError:
            pass
        return code

    gen = g0
    while gen < len(GENERATIONS):
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
                    return leave(0)
                reason = f"rank {rank} worker exited with {rc}"
                break
            if os.path.exists(_fail_file(rundir, gen)):          # another rank's worker failed: leave this generation with it
                reason = "another rank failed"
                if os.path.exists(_marker(rundir, gen, rank)):   # this rank's part was complete already: nothing to redo here
                    _kill(p)
                    return leave(0)
                _kill(p); rc = EXIT_RETRY
                break
            if time.monotonic() - t0 > limit:
                reason = f"rank {rank}: no completion within {limit:.0f}s"
                _kill(p); rc = EXIT_RETRY
                break
            time.sleep(0.1)
        # first to fail: tell the others why AND which generation comes next (atomic create; the losers keep the winner's record, so every
        # slot moves to the SAME generation even when the time budget runs out just now)
        nxt = next_generation(gen, t_start)
        if not os.path.exists(_fail_file(rundir, gen)):
            try:
                fd = os.open(_fail_file(rundir, gen) + f".{rank}", os.O_CREAT | os.O_EXCL | os.O_WRONLY, 0o644)
                os.write(fd, f"next={nxt}\n{reason or 'failed'}".encode()); os.close(fd)
                os.link(_fail_file(rundir, gen) + f".{rank}", _fail_file(rundir, gen))     # appears complete or not at all
            except OSError:
                pass
            try:
                os.unlink(_fail_file(rundir, gen) + f".{rank}")
            except OSError:
                pass
        try:
            rec = open(_fail_file(rundir, gen)).read().strip().split("\n", 1)
            if rec[0].startswith("next="):
                nxt = int(rec[0][5:]); rec = rec[1:]
            reason = (rec[0] if rec else "") or reason
        except (OSError, ValueError):
            pass
        abandoned.append({"generation": GENERATIONS[gen][0], "reason": reason})
        if rank == 0:
            log(f"generation {gen} ({GENERATIONS[gen][0]}) abandoned: {reason}" + (f"; skipping to generation {nxt}" if nxt != gen + 1 and nxt < len(GENERATIONS) else ""))
        gen = max(nxt, gen + 1)
    return leave(rc)


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
