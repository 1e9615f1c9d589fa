#!/usr/bin/env python3
"""where does a solve's wall time go beyond its iterations?  (one-off diagnostic: allocation of the work vectors, first-call effects)
usage: solve_overhead.py [grid=512]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import multigridsolver_amd as mg
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ctx = mg.Context(0); n = N ** 3
A = ctx.poisson3d(N)
h = mg.Hierarchy(A, 0.6, 1, 1).coarsen(10.0, 2, 8.0, 2500, 32).finalize()
b = ctx.vec(n).rand(seed=0); x = ctx.vec(n)
h.vcycle(b, x); ctx.sync()
ms = h.time_vcycle(b, x, reps=10)
print(f"cycle {ms:.3f} ms")
for rnd in range(3):
    t0 = time.perf_counter(); vs = [ctx.vec(n) for _ in range(8)]; ctx.sync(); t1 = time.perf_counter()
    del vs; ctx.sync(); t2 = time.perf_counter()
    print(f"round {rnd}: 8 work vectors created in {t1 - t0:.3f}s, freed in {t2 - t1:.3f}s")
for rnd in range(3):
    xs = ctx.vec(n); ctx.sync()
    t0 = time.perf_counter(); st, it, tol = mg.bicgstab(A, xs, b, h, 10000, 1e-10); ctx.sync(); t1 = time.perf_counter()
    print(f"solve {rnd}: status {st}, {it} iterations, {t1 - t0:.3f}s = {1e3 * (t1 - t0) / max(it, 1):.2f} ms per iteration")
    del xs
