#!/usr/bin/env python3
"""512^3 Poisson, FGCR(m) + K-cycle (energy form): iterations and seconds to 1e-10 against the smoother's damping omega, the number of K levels and the
restart length (the reference's default omega is 0.6; omega is an argument of its solve()).  usage: fgcr_knobs.py [N=512]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MGS_ARENA_GB", "110")
import multigridsolver_amd as mg

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
n = N ** 3
ctx = mg.Context(0)
A = ctx.poisson3d(N)
h = mg.Hierarchy(A, 0.6, 1, 1).coarsen(10.0, 2, 8.0, 2500, 32).finalize()
b = ctx.vec(n).rand(seed=0); nb = b.nrm2(); x = ctx.vec(n)
A.optimize()
ctx.set_option("kcycle_energy", 1)
for omega in (0.6, 0.7, 0.8, 0.9, 1.0):
    h.set_smoother(omega, 1, 1)
    for kl in (3, 4, 5):
        h.set_kcycle(kl)
        for _ in range(2): h.vcycle(b, x)
        ms = h.time_vcycle(b, x, reps=3)
        for m in (10, 16):
            x.fill(0.0); ctx.sync(); t0 = time.perf_counter()
            st, it, tol = mg.fgcr(A, x, b, h, m, 100, 1e-10)
            dt = time.perf_counter() - t0
            print(f"omega {omega} K levels {kl} ({ms:.2f} ms) FGCR({m}): status {st}, {it} iterations, {dt:.3f} s, true residual {A.residual(x, b).nrm2() / nb:.2e}", flush=True)
    h.set_kcycle(0)
ctx.close()
