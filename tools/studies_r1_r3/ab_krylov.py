#!/usr/bin/env python3
"""BiCGSTAB + V-cycle at N^3 with the dots of the two A·v products fused into the SpMV epilogue (option fuse_dots) on/off:
iterations, seconds, ms per iteration; the SpMV kernel alone for reference.  usage: ab_krylov.py [N=512]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import multigridsolver_amd as mg
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ctx = mg.Context(0)
A = ctx.poisson3d(N); n = N ** 3
h = mg.Hierarchy(A, 0.6, 1, 1).coarsen(10.0, 2, 8.0, 2500, 32).finalize()
b = ctx.vec(n).rand(seed=0)
x = ctx.vec(n); mg.bicgstab(A, x, b, h, 5, 1e-30)        # warm: codes, graphs
for rnd in range(2):
    for opt in (1, 0):
        ctx.set_option("fuse_dots", opt)
        x = ctx.vec(n); ctx.sync()
        t0 = time.perf_counter()
        st, it, tol = mg.bicgstab(A, x, b, h, 300, 1e-10)
        dt = time.perf_counter() - t0
        true = A.residual(x, b).nrm2() / b.nrm2()
        print(f"fuse_dots={opt}: status {st}, {it} iterations, {dt:.3f} s, {dt / it * 1e3:.2f} ms per iteration, true residual {true:.2e}", flush=True)
