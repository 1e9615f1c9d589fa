#!/usr/bin/env python3
"""Why do the aggregates of the coarse levels leave their row-block groups?  Per level of the N^3 hierarchy: the most frequent member-offset
tuples (rows of an aggregate relative to its first member), how many distinct 256-row blocks an aggregate touches and how far apart they are.
usage: stray_diag.py [grid=512] [levels=3]"""
import collections, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import multigridsolver_amd as mg
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
L = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ctx = mg.Context(0)
A = ctx.poisson3d(N)
h = mg.Hierarchy(A, 0.6, 1, 1).coarsen(10.0, 2, 8.0, 2500, 32).finalize()
print("levels", [h.level_shape(l)[0] for l in range(h.nlev)], flush=True)
for l in range(0, min(L + 1, h.nlev - 1)):
    agg = h.level_P(l).agg(); n = agg.size
    rows = np.nonzero(agg >= 0)[0]
    order = rows[np.argsort(agg[rows], kind="stable")]
    a = agg[order]
    starts = np.r_[0, np.nonzero(np.diff(a))[0] + 1]
    cnt = np.diff(np.r_[starts, a.size])
    first = order[starts]
    print(f"level {l}: {n} rows, {starts.size} aggregates, size histogram {np.bincount(cnt)[:9].tolist()}", flush=True)
    full = cnt == 4
    idx = starts[full]
    offs = np.stack([order[idx + q] - order[idx] for q in (1, 2, 3)], axis=1)
    u, c = np.unique(offs, axis=0, return_counts=True)
    top = np.argsort(-c)[:8]
    print("   4-member offset tuples:", [(u[t].tolist(), int(c[t])) for t in top], f"({u.shape[0]} distinct)")
    blk = order // 256
    bmin = np.minimum.reduceat(blk, starts); bmax = np.maximum.reduceat(blk, starts)
    span = bmax - bmin
    us, cs = np.unique(span, return_counts=True)
    top = np.argsort(-cs)[:8]
    print("   block span (last block - first block):", [(int(us[t]), int(cs[t])) for t in top])
    # distinct blocks per aggregate
    nb = np.add.reduceat(np.r_[1, (np.diff(blk) != 0).astype(np.int64)], starts) - np.r_[0, (blk[starts[1:]] != blk[starts[1:] - 1]).astype(np.int64)] + 0
    print("   group info:", h.group_info(l))
