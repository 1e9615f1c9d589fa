#!/usr/bin/env python3
"""One-GPU emulation of a MIDDLE rank of the N-GPU strong-scaling run (backend nccl = RCCL, world 1): the rank owns
planes [lo, hi) of the grid^3 operator and exchanges both halo planes WITH ITSELF through the same callbacks, pack
kernels and all_to_all calls the real run uses (numerically a periodic wrap of the slab — timing only).  Gives the
per-GPU cycle time including host/launch/collective-call overhead; link latency and the full-size replicated tail
are not in it.  usage: emulate_rank.py [grid=512] [ranks=8] [tail_rows=100000]
env: EMU_ONE=1 native configurations only; EMU_TRANSPORTS=p2p,rccl (default) the native transports to run, one after the other — the
cycle's result must have the SAME BITS on every transport (asserted; exit code 1 otherwise)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29931")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
os.environ.setdefault("MGS_NATIVE_SEGMENTS", "1")     # both transports in their pack-free form (the product's default keeps RCCL on packed messages: never run on real links)
import torch, torch.distributed as dist
import multigridsolver_amd as mg
from multigridsolver_amd import dist as mgd

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
R = int(sys.argv[2]) if len(sys.argv) > 2 else 8
tail_rows = int(sys.argv[3]) if len(sys.argv) > 3 else 100000
dist.init_process_group(os.environ.get("EMU_BACKEND", "nccl"))
torch.cuda.set_device(0)
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
ctx = mg.Context(0, stream.cuda_stream)
if R >= 3 and os.environ.get("EMU_SPLIT", "1") != "0":
    ctx.set_option("emu_split_self", 1)     # a middle rank talks to two neighbours: a packed exchange with itself goes out as two messages too
                                            # (EMU_SPLIT=0: one message, the convention of the round-2 rehearsals)
comm = mgd.Comm()
lo, hi = mgd.plane_range(N, R, R // 2)
n2 = N * N; n_loc = (hi - lo) * n2
A = ctx.poisson3d(N, lo, hi, local_cols=True)
if R >= 3:
    ids = np.concatenate([np.arange(n_loc - n2, n_loc), np.arange(0, n2)]).astype(np.int32)   # lower halo <- my last plane, upper <- my first
else:                 # two ranks: rank 1 has a lower neighbour only
    ids = np.arange(n_loc - n2, n_loc).astype(np.int32)
assert A.shape[1] - A.shape[0] == ids.size, "emulated plan does not fit the slab"
plan = mgd.LevelPlan(n_loc, [ids], [ids])
transports = [t for t in os.environ.get("EMU_TRANSPORTS", "p2p,rccl").split(",") if t]
configs = [(True, True, True, t) for t in transports]
if not os.environ.get("EMU_ONE"):
    configs += [(True, True, False, None), (False, True, False, None), (True, False, False, None)]
bits_differ = False
for overlap, fused, native, transport in configs:
    if transport:
        os.environ["MGS_NATIVE_TRANSPORT"] = transport
    sh = mgd.ShardedHierarchy(ctx, A, plan, 0.6, 1, 1, comm)
    t0 = time.perf_counter()
    sh.build(10.0, 2, 8.0, tail_rows=tail_rows, coarse_rows=2500, overlap=overlap, fused=fused, native=native, log=print)
    ctx.sync(); t_setup = time.perf_counter() - t0
    b = ctx.vec(n_loc).rand(seed=0); x = ctx.vec(A.shape[1])
    time.sleep(float(os.environ.get("EMU_SLEEP", "0")))     # lets torch's collective watchdog retire the setup collectives first
    for i in range(5):
        sh.vcycle(b, x)
        if os.environ.get("EMU_VERBOSE"):
            ctx.sync(); print("  cycle", i, sh.h.graph_info(), mg.lib().mgs_last_error(ctx.h), flush=True)
    ctx.sync(); torch.cuda.synchronize()
    xn = x.numpy(n_loc)
    if native:
        assert sh.native and sh.native_transport == transport, f"transport {transport} was not installed"
    if fused:
        if "ref" in globals():
            if native and not np.array_equal(xn, ref):
                bits_differ = True
            print("   same bits as the first configuration:", bool(np.array_equal(xn, ref)),
                  f"(relative difference {np.linalg.norm(xn - ref) / np.linalg.norm(ref):.2e}; the grouped t-form and the r/b form of the cycle agree to rounding, not bit for bit)", flush=True)
        else:
            ref = xn
    ex0 = sh.n_exchanges
    t0 = time.perf_counter()
    reps = int(os.environ.get("EMU_REPS", "50"))
    for _ in range(reps):
        sh.vcycle(b, x)
    t_host = time.perf_counter() - t0          # time to ENQUEUE the cycles (host side)
    ctx.sync(); torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"native={int(sh.native)} transport={getattr(sh, 'native_transport', None)} overlap={int(overlap)} fused={int(fused)}: sharded levels {[p.n_loc for p in sh.plans]} (+tail {sh.tail.nlev} levels), setup {t_setup:.2f} s, "
          f"{t_all / reps * 1e3:.3f} ms per cycle (host enqueue {t_host / reps * 1e3:.3f} ms), {(sh.n_exchanges - ex0) / reps:.0f} exchanges per cycle; "
          f"level-0 form {sh.h.fused_info(0)}; graph {sh.h.graph_info()}", flush=True)
    sh.close()
    del sh, b, x
# the same slab as an unsharded-size reference: one GPU's share of the work without any exchange
ctx.close()
dist.destroy_process_group()
if bits_differ:
    print("TRANSPORTS DISAGREE", flush=True)
    sys.exit(1)
print("EMU_OK", flush=True)
