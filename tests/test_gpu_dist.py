"""Multi-rank GPU test of the sharded path on a single-GPU box: 2 (and 3) ranks share GPU 0,
backend gloo with host-staged halo buffers.  Everything except the RCCL transport is the code
the 8-GPU run executes: plane shards with local column numbering, mgs_aggregate_shard, the
handshake, mgs_galerkin_shard, the C++ V-cycle with halo/coarse-tail callbacks, all-reduced dots."""
import os
import subprocess
import sys

import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world,N,tail,overlap,fused", [(2, 20, 1500, 1, 1), (3, 18, 800, 1, 1), (2, 40, 3000, 1, 1), (2, 20, 1500, 0, 1),
                                                        (2, 20, 1500, 1, 0), (3, 18, 800, 0, 0), (4, 24, 1200, 1, 1),
                                                        (4, 128, 40000, 1, 1)])      # 524 288 rows per rank: the product's own thresholds pick the forms
def test_sharded_vcycle_matches_oracle(world, N, tail, overlap, fused):
    """callback transport (torch.distributed / gloo, halo buffers staged through the host), 2-4 ranks on this one GPU, against the oracle's cycle on
    the globally assembled hierarchy; the worker also checks Galerkin products, K-cycles and a solve to 1e-10"""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MGS_NATIVE_RCCL="0")     # callbacks only (a native transport would take over otherwise)
    if N >= 100:
        env["MGS_OPTIONS"] = "split_min_rows=400000"      # the worker leaves the interior/boundary split threshold alone when the option is given
    port = 29600 + (os.getpid() % 1000) + world
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(REPO, "tests", "dist_gpu_worker.py"), str(N), str(tail), str(overlap), str(fused)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=REPO)
    assert r.returncode == 0 and "DIST_OK" in r.stdout, r.stdout[-3000:] + r.stderr[-6000:]


@pytest.mark.parametrize("world,N,tail,native", [(2, 40, 700, 0), (3, 36, 500, 0), (2, 40, 700, 1), (2, 40, 700, 2), (3, 36, 500, 2)])
def test_kcycle_on_row_shards(world, N, tail, native):
    """K-cycle on sharded levels: the inner products of the two Krylov steps summed over the ranks (callback transport: host
    all-reduce; native 1: ncclAllReduce of the stand-in; native 2: the peer-to-peer transport's rank-ordered sum) — K-cycle
    applications vs the oracle's K-cycle (≤1e-9, energy form at every size)"""
    fake = os.path.join(REPO, "tests", "fake_rccl", "libfake_rccl.so")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MGS_NATIVE_RCCL="0")
    if native == 1:
        env.update(MGS_NATIVE_RCCL="force", MGS_LIBRCCL=fake)
    elif native == 2:
        env.update(MGS_NATIVE_RCCL="1", MGS_NATIVE_TRANSPORT="p2p")
    port = 29850 + (os.getpid() % 1000) + world + 10 * native
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(REPO, "tests", "dist_gpu_worker.py"), str(N), str(tail), "1", "1"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=REPO)
    assert r.returncode == 0 and "DIST_OK" in r.stdout and "kcycle_err=" in r.stdout and "kcycle_err=None" not in r.stdout, r.stdout[-3000:] + r.stderr[-6000:]


@pytest.mark.parametrize("world,N,tail,native", [(2, 40, 3000, 0), (3, 36, 2000, 0), (2, 40, 3000, 1), (3, 36, 2000, 1), (3, 36, 2000, 2)])
def test_grouped_pre_pass_on_row_shards(world, N, tail, native):
    """the grouped pre pass (restriction inside the pre pass, t-form post pass) on row shards — halo payload exchanged first, halo-tagged
    pattern codes — forced onto these small grids (group_min_blocks=1, 60 % strays allowed); callback transport and native transport
    (stand-in RCCL); cycle vs the oracle, solve to 1e-10"""
    fake = os.path.join(REPO, "tests", "fake_rccl", "libfake_rccl.so")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MGS_OPTIONS="group_min_blocks=1,group_stray_pct=60,split_min_rows=100000000", MGS_NATIVE_RCCL="0")
    if native == 1:
        env.update(MGS_NATIVE_RCCL="force", MGS_LIBRCCL=fake)
    elif native == 2:      # peer-to-peer transport (csrc/comm_p2p.hip)
        env.update(MGS_NATIVE_RCCL="1", MGS_NATIVE_TRANSPORT="p2p")
    port = 29750 + (os.getpid() % 1000) + world + 10 * native
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(REPO, "tests", "dist_gpu_worker.py"), str(N), str(tail), "1", "1"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=REPO)
    assert r.returncode == 0 and "DIST_OK" in r.stdout and "grouped_levels=" in r.stdout and "grouped_levels=0" not in r.stdout, r.stdout[-3000:] + r.stderr[-6000:]


@pytest.mark.parametrize("world,N,tail,opts", [(2, 20, 1500, ""), (3, 18, 800, ""), (2, 40, 3000, ""), (4, 18, 600, ""),
                                                  (2, 40, 3000, "group_min_blocks=1,group_stray_pct=60,split_min_rows=100000000"), (2, 40, 700, ""),
                                                  # 128³ on four ranks (524 288 rows each): the thresholds of the product decide the forms, as in the N-GPU run —
                                                  # grouped pre pass with halo columns, strip-major block map, pack-free range sends, zoned shard aggregation
                                                  (4, 128, 40000, "split_min_rows=400000")])
@pytest.mark.parametrize("transport", ["rccl-stand-in", "p2p"])
def test_native_cycle_captured_in_a_graph_multi_rank(world, N, tail, opts, transport):
    """The DEFAULT multi-GPU path — native transport with the whole cycle (exchanges, tail all-gather, tail cycle, K-cycle scalars
    summed over the ranks) captured in one hipGraph — with several ranks on this one GPU: tests/fake_rccl in its stream-ordered mode
    (device→pinned copy, host function moving the bytes through files, pinned→device copy: capturable like RCCL's kernels).  Asserts
    in the worker: capture after two eager cycles, replay == eager bit for bit, more (rhs, out) pairs than cache slots, cycle vs the
    oracle, K-cycle vs the oracle (tail 700: three sharded levels), solve to 1e-10.
    transport "p2p": the product's own peer-to-peer transport (csrc/comm_p2p.hip) — the ranks' processes map each other's device windows
    (hipIpcOpenMemHandle) and every exchange, the tail's all-gather and the all-reduces are ONE kernel each, captured like any other;
    the setup cross-checks its exchanges against torch.distributed (gloo) bit for bit, the worker the cycle against the oracle."""
    fake = os.path.join(REPO, "tests", "fake_rccl", "libfake_rccl.so")
    if not os.path.exists(fake):
        subprocess.run(["make", "-C", os.path.dirname(fake)], check=True, capture_output=True)
    if transport == "p2p":
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MGS_NATIVE_RCCL="1", MGS_NATIVE_TRANSPORT="p2p", MGS_EXPECT_GRAPH="1")
    else:
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MGS_NATIVE_RCCL="force", MGS_LIBRCCL=fake, MGS_FAKE_RCCL_STREAM="1", MGS_NATIVE_SEGMENTS="1")
    if opts:
        env["MGS_OPTIONS"] = opts
    port = 29950 + (os.getpid() % 1000) + world + (7 if opts else 0) + (3 if tail == 700 else 0) + (40 if transport == "p2p" else 0)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(REPO, "tests", "dist_gpu_worker.py"), str(N), str(tail), "1", "1"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=REPO)
    assert r.returncode == 0 and "DIST_OK" in r.stdout and "captured_cycles" in r.stdout, r.stdout[-3000:] + r.stderr[-6000:]


@pytest.mark.parametrize("world,N,tail", [(2, 20, 1500), (3, 18, 800), (4, 18, 600), (2, 40, 3000)])
def test_native_transport_multi_rank_matches_oracle(world, N, tail):
    """the NATIVE transport of the C++ cycle (ncclSend/ncclRecv groups, tail all-gather, all-reduced dots) with several
    ranks on this one GPU: RCCL's entry points are served by tests/fake_rccl (file-based stand-in, real RCCL refuses
    two ranks per device), everything above them — plan hand-over, pack kernels, peer offsets, uneven tail shards
    (world 4 on 18 planes) — is the code of the N-GPU run; cycle vs the oracle, solve to 1e-10."""
    fake = os.path.join(REPO, "tests", "fake_rccl", "libfake_rccl.so")
    if not os.path.exists(fake):
        subprocess.run(["make", "-C", os.path.dirname(fake)], check=True, capture_output=True)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MGS_NATIVE_RCCL="force", MGS_LIBRCCL=fake, MGS_NATIVE_SEGMENTS="1")
    port = 29650 + (os.getpid() % 1000) + world
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(REPO, "tests", "dist_gpu_worker.py"), str(N), str(tail), "1", "1"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=REPO)
    assert r.returncode == 0 and "DIST_OK" in r.stdout, r.stdout[-3000:] + r.stderr[-6000:]


@pytest.mark.parametrize("world,native", [(2, 0), (3, 0), (2, 1), (3, 2), (3, 3)])
def test_sharded_general_operator_matches_oracle(world, native, inputs):
    """a bundled, nonsymmetric operator (CSky3d30) sharded by contiguous row ranges with the generic halo plan
    (shard_from_global): sharded cycle vs the oracle on the globally assembled hierarchy, sharded solve to 1e-10.
    native 1: the native transport on the stand-in RCCL (send lists that are NOT plane ranges: packed or multi-range exchanges,
    zones over irregular lists, halo of the tail level from the replicated solution); 2: the same, stream-ordered and captured;
    3: the peer-to-peer transport (captured)."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MGS_NATIVE_RCCL="0")
    if native == 3:
        env.update(MGS_NATIVE_RCCL="1", MGS_NATIVE_TRANSPORT="p2p", MGS_EXPECT_GRAPH="1")
    elif native:
        env.update(MGS_NATIVE_RCCL="force", MGS_LIBRCCL=os.path.join(REPO, "tests", "fake_rccl", "libfake_rccl.so"), MGS_NATIVE_SEGMENTS="1")
        if native == 2:
            env["MGS_FAKE_RCCL_STREAM"] = "1"
    port = 29700 + (os.getpid() % 1000) + world + 20 * native
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(REPO, "tests", "dist_gpu_worker.py"), "0", "3000", "1", "1", inputs["CSky3d30"]]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=REPO)
    assert r.returncode == 0 and "DIST_OK" in r.stdout, r.stdout[-3000:] + r.stderr[-6000:]


def test_solve_cli_single_and_sharded(inputs, orc, tmp_path):
    """python -m multigridsolver_amd.solve: one GPU, and 2 ranks (rows of CSky3d30 sharded) — same right-hand side as
    the reference driver, true residual checked with the oracle"""
    import numpy as np
    Ao = orc.Csr.read(inputs["CSky3d30"]); b = orc.rand_rhs(Ao.shape[0])
    for world in (1, 2):
        dump = str(tmp_path / f"x{world}.bin")
        base = [sys.executable, "-m", "multigridsolver_amd.solve", inputs["CSky3d30"], "--tol", "1e-9", "--dump-x", dump]
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MGS_DIST_BACKEND="gloo", MGS_DIST_SHARE_GPU="1")
        if world > 1:
            base = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
                    "--master-port", str(29800 + os.getpid() % 1000), "-m", "multigridsolver_amd.solve", inputs["CSky3d30"], "--tol", "1e-9", "--dump-x", dump]
        r = subprocess.run(base, capture_output=True, text=True, timeout=600, env=env, cwd=REPO)
        assert r.returncode == 0 and "Number of iterations BICG" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
        x = np.fromfile(dump, dtype="<f8")
        assert np.linalg.norm(Ao.residual(x, b)) / np.linalg.norm(b) < 1.5e-9


def test_rccl_transport_on_library_memory_world1():
    """the RCCL transport itself (backend nccl) at world 1: collectives on library-owned device memory, blocking and async"""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29900 + os.getpid() % 90), RANK="0", WORLD_SIZE="1")
    r = subprocess.run([sys.executable, os.path.join(REPO, "tests", "nccl_world1_worker.py")], capture_output=True, text=True, timeout=300, env=env, cwd=REPO)
    assert r.returncode == 0 and "NCCL_W1_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_native_transports_same_bits_rank_of_8_at_256():
    """tools/emulate_rank.py as a test (round-3 hygiene item): a MIDDLE rank of 8 of the 256^3 problem (2.1 M rows, every level above the product's
    own thresholds: grouped pre pass with halo columns, pack-free range sends, zoned aggregation, tail halo fill), exchanging both halo
    planes with itself at world 1 over REAL RCCL (backend nccl) and over the peer-to-peer transport, the whole cycle captured in a
    hipGraph and replayed: the captured cycle's result has the same bits on both transports."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29200 + os.getpid() % 90), RANK="0", WORLD_SIZE="1", EMU_ONE="1", EMU_REPS="10",
               EMU_TRANSPORTS="p2p,rccl")
    for k in ("MGS_NATIVE_RCCL", "MGS_NATIVE_TRANSPORT", "MGS_OPTIONS"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "emulate_rank.py"), "256", "8", "100000"], capture_output=True, text=True, timeout=600, env=env, cwd=REPO)
    assert r.returncode == 0 and "EMU_OK" in r.stdout and r.stdout.count("captured_cycles") >= 2, r.stdout[-3000:] + r.stderr[-6000:]
    assert "transport=p2p" in r.stdout and "transport=rccl" in r.stdout


@pytest.mark.parametrize("mode", ["exchange", "timeout"])
def test_p2p_transport_raw_two_processes(mode, tmp_path):
    """the peer-to-peer transport by itself through the C-ABI (mgs_comm_p2p_create / _connect / _exchange_raw / _allgather_raw / _allreduce_raw /
    _selftest / _info), two processes on this GPU, handles through files.  "exchange": data exact over both window slots, odd lengths, rank-ordered
    all-reduce sum.  "timeout": a peer that connects and stays silent — the wait is bounded (MGS_P2P_TIMEOUT_S), the error word is set, later exchanges
    return at once: no wave spins forever (what the launcher's non-zero exit and generation change rest on)."""
    env = dict(os.environ, MGS_P2P_TIMEOUT_S="1" if mode == "timeout" else "20")
    procs = [subprocess.Popen([sys.executable, os.path.join(REPO, "tests", "p2p_raw_worker.py"), str(r), str(tmp_path), mode], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                              text=True, env=env, cwd=REPO) for r in (0, 1)]
    outs = [p.communicate(timeout=180) for p in procs]
    assert all(p.returncode == 0 for p in procs), [(p.returncode, o[0][-1000:], o[1][-3000:]) for p, o in zip(procs, outs)]
    if mode == "timeout":
        assert "P2P_TIMEOUT_OK" in outs[0][0] and "P2P_PEER_SILENT" in outs[1][0]
    else:
        assert all("P2P_RAW_OK" in o[0] for o in outs)


@pytest.mark.parametrize("world", [2, 4])
def test_bench_line_under_the_drivers_launcher(world):
    """the driver's own N > 1 command — `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py
    --gpus N --steps K --warmup W` — with the ranks sharing this one GPU (gloo rendezvous: RCCL refuses two ranks per device; the peer-to-peer transport
    is the real one, IPC windows between the processes): supervisors, workers, first generation, ONE JSON line from rank 0 with the contract's fields,
    not degraded, on the peer-to-peer transport with its cycle captured, its solve converged."""
    import json
    env = dict(os.environ, MGS_DIST_BACKEND="gloo", MGS_DIST_SHARE_GPU="1", MGS_ARENA_GB="0", MGS_NATIVE_TRANSPORT="p2p")   # launch.first_generation: gloo + p2p starts in generation 0
    for k in ("MGS_NATIVE_RCCL", "MGS_OPTIONS", "MGS_BENCH_WORKER"):
        env.pop(k, None)
    port = 29400 + (os.getpid() % 500) + world
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(REPO, "bench.py"), "--gpus", str(world), "--steps", "5", "--warmup", "2", "--grid", "128", "--no-cpu", "--kernel-reps", "5"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=REPO)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-6000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-3000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == world and d["steps"] == 5 and d["warmup"] == 2 and d["scaling"] == "strong" and d["unit"] == "V-cycles/s"
    assert abs(d["value"] - 1e3 / d["ms_per_step"]) <= 1e-6 * d["value"]
    assert d["degraded"] is False and d["abandoned_generations"] == [], d.get("abandoned_generations")
    t = d["transport"]
    assert t["native"] == "p2p" and t["world"] == world and t["p2p"] is not None, t
    assert "multi_gpu_note" in d
