#!/usr/bin/env python3
"""BiCGSTAB iteration counts / time to 1e-10 for a few smoother settings (dev tool)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import multigridsolver_amd as mg
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ctx = mg.Context(0)
A = ctx.poisson3d(N); n = N ** 3
h = mg.Hierarchy(A, 0.6, 1, 1).coarsen(10.0, 2, 8.0, 1024, 32).finalize()
b = ctx.vec(n).rand(seed=0)
for kl in (4, 5, 6, 7, 8, 10):
    for w in (0.8,):
        n1 = n2 = 1
        h.set_smoother(w, n1, n2); h.set_kcycle(kl)
        ms = h.time_vcycle(b, ctx.vec(n), reps=3)
        print(f"kcycle levels {kl}: {ms:.3f} ms per cycle", flush=True)
        x = ctx.vec(n)
        t0 = time.perf_counter()
        st, it, tol = mg.fgcr(A, x, b, h, 10, 300, 1e-10)
        dt = time.perf_counter() - t0
        true = A.residual(x, b).nrm2() / b.nrm2()
        print(f"N={N} FGCR(10) V({n1},{n2}) omega={w} kl={kl}: status {st} iters {it} tol {tol:.2e} true {true:.2e} time {dt:.3f}s", flush=True)
