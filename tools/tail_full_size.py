#!/usr/bin/env python3
"""What the replicated tail costs at its REAL size: the one-GPU rehearsal of a rank of R (tools/emulate_rank.py) carries a tail built from the
rank's own share of the last sharded level (1/R of the rows); the real run replicates the WHOLE level on every GPU.  This times both as
standalone hierarchies (captured cycle, min of 3 x 50): level 4 of the 512^3 hierarchy (524 k rows: the real tail's entry level at 8 ranks
with tail_rows = 600 k) against level 4 of the 256^3 hierarchy (66 k rows: what the rehearsal runs).  usage: tail_full_size.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import multigridsolver_amd as mg

ctx = mg.Context(0)
out = {}
for N in (512, 256):
    A = ctx.poisson3d(N)
    h = mg.Hierarchy(A, 0.6, 1, 1).coarsen(10.0, 2, 8.0, 2500, 32).finalize()
    rows = [h.level_shape(l)[0] for l in range(h.nlev)]
    rp, ci, v = h.level_A(4).download(); n4 = rows[4]
    del h, A
    T = ctx.csr(n4, n4, rp, ci, v)
    ht = mg.Hierarchy(T, 0.6, 1, 1).coarsen(10.0, 2, 8.0, 2500, 32).finalize()
    b = ctx.vec(n4).rand(seed=0); x = ctx.vec(n4)
    for _ in range(5):
        ht.vcycle(b, x)
    ms = min(ht.time_vcycle(b, x, reps=50) for _ in range(3))
    out[N] = (n4, ht.nlev, ms)
    print(f"level 4 of the {N}^3 hierarchy as a hierarchy of its own: {n4} rows, {ht.nlev} levels, {ms * 1e3:.1f} us per cycle", flush=True)
    del ht, T, b, x
print(f"full-size tail minus rehearsed tail: {(out[512][2] - out[256][2]) * 1e3:.1f} us per cycle")
ctx.close()
