#!/usr/bin/env python3
"""BiCGSTAB + V-cycle iterations / time to 1e-10 on the N^3 Poisson operator over smoother settings (omega, nu1, nu2) and aggregate
size (pairwise passes) — which configuration solves fastest.  usage: smoother_scan.py [N=512]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import multigridsolver_amd as mg
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ctx = mg.Context(0)
A = ctx.poisson3d(N); n = N ** 3
b = ctx.vec(n).rand(seed=0)
for npass in (2, 3):
    h = mg.Hierarchy(A, 0.6, 1, 1).coarsen(10.0, npass, 8.0 if npass == 2 else 100.0, 2500, 32).finalize()
    print(f"npass {npass}: levels", [h.level_shape(l)[0] for l in range(h.nlev)], flush=True)
    for (w, n1, n2) in [(0.6, 1, 1), (0.5, 1, 1), (0.7, 1, 1), (0.8, 1, 1), (0.9, 1, 1), (0.6, 2, 2), (0.8, 2, 2), (0.8, 1, 2), (0.8, 0, 1), (0.8, 0, 2)]:
        h.set_smoother(w, n1, n2)
        x = ctx.vec(n)
        ms = h.time_vcycle(b, x, reps=5)
        x.fill(0.0); ctx.sync()
        t0 = time.perf_counter()
        st, it, tol = mg.bicgstab(A, x, b, h, 400, 1e-10)
        dt = time.perf_counter() - t0
        print(f"  V({n1},{n2}) omega={w}: cycle {ms:.2f} ms, status {st}, {it} iterations, {dt:.2f} s", flush=True)
    for sigma in (1.3, 1.6, 1.8, 2.0, 2.3):                 # over-correction x += sigma * P e_c, with the default smoother and with omega = 0.8
        for w in (0.6, 0.8):
            h.set_smoother(w, 1, 1); h.set_correction_scale(sigma)
            x = ctx.vec(n); h.vcycle(b, x); x.fill(0.0); ctx.sync()
            t0 = time.perf_counter()
            st, it, tol = mg.bicgstab(A, x, b, h, 400, 1e-10)
            print(f"  sigma={sigma} omega={w}: status {st}, {it} iterations, {time.perf_counter() - t0:.2f} s", flush=True)
    h.set_correction_scale(1.0)
    del h
