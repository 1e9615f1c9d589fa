#!/usr/bin/env python3
"""FGCR(10) + K-cycle(4) at N^3: iterations to 1e-10 for several right-hand sides (seeds) — how much does the count of this
nonlinear preconditioner move with the data?  (Between two builds that differ only in how the inner products are blocked, the
seed-0 count moved from 118 to 160.)  usage: fgcr_sensitivity.py [N=512]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import multigridsolver_amd as mg
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ctx = mg.Context(0); n = N ** 3
A = ctx.poisson3d(N)
h = mg.Hierarchy(A, 0.6, 1, 1).coarsen(10.0, 2, 8.0, 2500, 32).finalize()
h.set_kcycle(4)
for seed in range(5):
    b = ctx.vec(n).rand(seed=seed); x = ctx.vec(n)
    t0 = time.perf_counter(); st, it, tol = mg.fgcr(A, x, b, h, 10, 400, 1e-10); dt = time.perf_counter() - t0
    print(f"seed {seed}: FGCR(10)+K(4) status {st}, {it} iterations, {dt:.2f} s", flush=True)
h.set_kcycle(0)
for seed in range(3):
    b = ctx.vec(n).rand(seed=seed); x = ctx.vec(n)
    st, it, tol = mg.fgcr(A, x, b, h, 10, 400, 1e-10)
    st2, it2, tol2 = mg.bicgstab(A, ctx.vec(n), b, h, 400, 1e-10)
    print(f"seed {seed}: FGCR(10)+V-cycle {it} iterations; BiCGSTAB+V-cycle {it2}", flush=True)
