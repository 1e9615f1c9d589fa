#!/usr/bin/env python3
"""How contiguous are the send lists of a sharded hierarchy's halo plans?  One-GPU rehearsal of a middle rank (as tools/emulate_rank.py):
per level the number of rows sent, the contiguous runs they form and their span.  usage: halo_segments.py [grid=256] [ranks=8] [slack=2.5]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29941")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import torch, torch.distributed as dist
import multigridsolver_amd as mg
from multigridsolver_amd import dist as mgd

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
R = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dist.init_process_group("gloo")
torch.cuda.set_device(0)
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
ctx = mg.Context(0, stream.cuda_stream)
comm = mgd.Comm()
lo, hi = mgd.plane_range(N, R, R // 2)
n2 = N * N; n_loc = (hi - lo) * n2
A = ctx.poisson3d(N, lo, hi, local_cols=True)
ids = np.concatenate([np.arange(n_loc - n2, n_loc), np.arange(0, n2)]).astype(np.int32)
plan = mgd.LevelPlan(n_loc, [ids], [ids])
sh = mgd.ShardedHierarchy(ctx, A, plan, 0.6, 1, 1, comm)
sh.build(10.0, 2, 8.0, tail_rows=20000, coarse_rows=2500, native=False, log=print)
for l, p in enumerate(sh.plans):
    for peer, idx in enumerate(p.send_idx):
        idx = np.asarray(idx, dtype=np.int64)
        if idx.size == 0:
            continue
        cuts = np.nonzero(np.diff(idx) != 1)[0] + 1
        runs = np.diff(np.concatenate([[0], cuts, [idx.size]]))
        print(f"level {l} peer {peer}: {idx.size} rows of {p.n_loc}, {runs.size} contiguous runs (longest {sorted(runs)[-5:]}), ids {idx[0]}..{idx[-1]}, "
              f"first run starts {idx[np.concatenate([[0], cuts])][:6]}", flush=True)
ctx.close()
dist.destroy_process_group()
