#!/usr/bin/env python3
"""per-level Jacobi-kernel time and GB/s of the 512^3 hierarchy, strip-major sweep on/off (one process)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import multigridsolver_amd as mg
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ctx = mg.Context(0)
A = ctx.poisson3d(N)
h = mg.Hierarchy(A, 0.6, 1, 1).coarsen(10.0, 2, 8.0, 1024, 32).finalize()
for l in range(min(h.nlev - 1, 6)):
    Al = h.level_A(l); n, nnz = h.level_shape(l)
    x = ctx.vec(n).rand(seed=1); b = ctx.vec(n).rand(seed=2); y = ctx.vec(n); d = Al.diag_inv()
    byts = 12 * nnz + 36 * n
    res = {}
    for rnd in range(3):
        for strip in (0, -1, 16, 128):
            ctx.set_option("strip", strip)
            res.setdefault(strip, []).append(Al.time_kernel(mg.OP_JACOBI, x, b=b, dinv=d, out=y, reps=10))
    ctx.set_option("strip", -1)
    print(f"L{l} n={n} nnz/row={nnz/n:.2f} plan={Al.plan_info()}")
    for strip, t in res.items():
        print(f"    strip {strip:4d}: {np.median(t)*1e3:9.1f} us  {byts/np.median(t)/1e6:6.0f} GB/s")
