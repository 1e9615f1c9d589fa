#!/usr/bin/env python3
"""Would row WINDOWS (runs of consecutive rows owned by one row block, where a row's owner is the block of its aggregate's first member) keep
the aggregates of the coarse levels inside one workgroup?  Per level of the N^3 hierarchy: runs per owner block, run lengths, and the share of
aggregates that stay stray when a group holds at most 4 windows of at most 256 rows.
usage: range_sim.py [grid=512] [levels=3]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import multigridsolver_amd as mg
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
L = int(sys.argv[2]) if len(sys.argv) > 2 else 3
RB, SLOTS = 256, 4
ctx = mg.Context(0)
A = ctx.poisson3d(N)
h = mg.Hierarchy(A, 0.6, 1, 1).coarsen(10.0, 2, 8.0, 2500, 32).finalize()
print("levels", [h.level_shape(l)[0] for l in range(h.nlev)], flush=True)
for l in range(0, min(L + 1, h.nlev - 1)):
    agg = h.level_P(l).agg().astype(np.int64); n = agg.size
    rows = np.nonzero(agg >= 0)[0]
    nc = int(agg.max()) + 1
    first = np.full(nc, n, dtype=np.int64)
    np.minimum.at(first, agg[rows], rows)
    owner = np.arange(n, dtype=np.int64) // RB
    owner[rows] = first[agg[rows]] // RB
    start = np.r_[True, owner[1:] != owner[:-1]]
    s = np.nonzero(start)[0]
    ln = np.diff(np.r_[s, n])
    # cut runs longer than RB
    nsplit = int(np.sum((ln - 1) // RB))
    own = owner[s]
    nruns = s.size + nsplit
    per_owner = np.bincount(own, minlength=(n + RB - 1) // RB)
    owners = np.count_nonzero(per_owner)
    print(f"level {l}: {n} rows, {nc} aggregates, {nruns} windows ({nsplit} from cuts) for {owners} owner blocks of {(n + RB - 1) // RB}; "
          f"windows per owner histogram {np.bincount(per_owner[per_owner > 0])[:10].tolist()}", flush=True)
    q = np.percentile(ln, [5, 25, 50, 75, 95])
    print(f"   run length percentiles 5/25/50/75/95: {q.tolist()}, mean {ln.mean():.1f}; rows in runs < 64: {ln[ln < 64].sum() / n:.3f}")
    # rank of each run inside its owner (runs ascending by row): those with rank >= SLOTS fall out of the owner's group
    order = np.lexsort((s, own))
    so, oo = s[order], own[order]
    newo = np.r_[True, oo[1:] != oo[:-1]]
    idx0 = np.maximum.accumulate(np.where(newo, np.arange(oo.size), 0))
    rank = np.arange(oo.size) - idx0
    out_runs = order[rank >= SLOTS]
    out_row = np.zeros(n, dtype=bool)
    for r in out_runs:
        out_row[s[r]:s[r] + ln[r]] = True
    stray = np.zeros(nc, dtype=bool)
    stray[agg[rows[out_row[rows]]]] = True
    print(f"   windows outside their owner's group: {out_runs.size}; stray aggregates {stray.sum()} = {100.0 * stray.sum() / nc:.2f} %")
    # lane use: rows / (windows * RB)
    print(f"   lane use (rows / (windows x 256)): {n / (nruns * RB):.3f}")
