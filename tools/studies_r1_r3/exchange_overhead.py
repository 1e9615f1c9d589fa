#!/usr/bin/env python3
"""Host + RCCL launch overhead of one halo exchange as dist.py issues it (world 1: the 'peer' is
this rank itself, so bytes move by a local copy — what remains is the per-exchange fixed cost)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29741")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import torch
import torch.distributed as dist
import multigridsolver_amd as mg
from multigridsolver_amd import dist as mgd

dist.init_process_group("nccl")
torch.cuda.set_device(0)
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
ctx = mg.Context(0, stream.cuda_stream)
comm = mgd.Comm()
N = 256
A = ctx.poisson3d(N, 0, N, local_cols=False)
n = A.shape[0]
n2 = N * N
plan = mgd.LevelPlan(n, [np.arange(0, n2, dtype=np.int32)], [np.arange(0, n2, dtype=np.int32)])
sh = mgd.ShardedHierarchy(ctx, A, plan, 0.6, 1, 1, comm)
sh._prepare_plan(0)
x = ctx.vec(n + n2).rand(seed=1)
for mode in ("sync", "split"):
    for _ in range(5):
        sh._exchange(0, x.ptr)
    ctx.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 200
    for _ in range(reps):
        if mode == "sync":
            sh._exchange(0, x.ptr)
        else:
            sh._exchange_begin(0, x.ptr); sh._exchange_end(0, x.ptr)
    t_host = (time.perf_counter() - t0) / reps
    ctx.sync(); torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / reps
    print(f"{mode}: host issue {t_host*1e6:.1f} us per exchange, end-to-end {t_all*1e6:.1f} us per exchange ({n2*8/1e6:.1f} MB payload)")
dist.destroy_process_group()
