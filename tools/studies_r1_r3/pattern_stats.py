#!/usr/bin/env python3
"""Row-pattern statistics of a device-built hierarchy: how many distinct (col − row) offset tuples the rows of a
level have, globally and per 256-row block — the feasibility data behind the pattern-coded index."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import multigridsolver_amd as mg
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
ctx = mg.Context(0)
A = ctx.poisson3d(N)
h = mg.Hierarchy(A, 0.6, 1, 1).coarsen(10.0, 2, 8.0, 2500, 32).finalize()
for l in range(h.nlev - 1):
    rp, ci, v = h.level_A(l).download()
    n = rp.size - 1
    rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(rp))
    off = (ci.astype(np.int64) - rows).astype(np.uint64)
    pos = (np.arange(ci.size, dtype=np.int64) - np.repeat(rp[:-1].astype(np.int64), np.diff(rp))).astype(np.uint64)
    key = (off + np.uint64(0x9E3779B97F4A7C15)) * (np.uint64(2) * pos + np.uint64(0xBF58476D1CE4E5B9))
    key ^= key >> np.uint64(29)
    hsh = np.add.reduceat(key, rp[:-1].astype(np.int64)) + np.diff(rp).astype(np.uint64) * np.uint64(0x94D049BB133111EB)
    uniq, inv, cnt = np.unique(hsh, return_inverse=True, return_counts=True)
    order = np.argsort(-cnt)
    cover255 = cnt[order[:255]].sum() / n
    cover64k = cnt[order[:65535]].sum() / n
    nb = (n + 255) // 256
    blk = np.arange(n) // 256
    pair = np.unique(blk.astype(np.int64) * uniq.size + inv)
    per_blk = np.bincount((pair // uniq.size).astype(np.int64), minlength=nb)
    # per-block pattern-table ints (sum of pattern lengths)
    rowlen = np.diff(rp)
    first_row_of = np.zeros(uniq.size, dtype=np.int64); first_row_of[inv[::-1]] = np.arange(n)[::-1]
    plen = rowlen[first_row_of]
    ints = np.bincount((pair // uniq.size).astype(np.int64), weights=plen[(pair % uniq.size).astype(np.int64)], minlength=nb)
    print(f"L{l}: n={n} nnz/row={ci.size/n:.2f} distinct={uniq.size} cover(top255)={cover255:.4f} cover(top64k)={cover64k:.4f} "
          f"per-block patterns: median={np.median(per_blk):.0f} p90={np.percentile(per_blk,90):.0f} p99={np.percentile(per_blk,99):.0f} max={per_blk.max()} "
          f"blocks<=16: {np.mean(per_blk<=16):.3f} <=32: {np.mean(per_blk<=32):.3f} <=64: {np.mean(per_blk<=64):.3f}; table ints median={np.median(ints):.0f} p90={np.percentile(ints,90):.0f} "
          f"mean table/vals bytes={4*ints.mean()/(8*ci.size/nb):.3f}", flush=True)
