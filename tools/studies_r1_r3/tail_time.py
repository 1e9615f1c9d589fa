import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import multigridsolver_amd as mg
ctx = mg.Context(0)
for N in (44, 57, 84):
    A = ctx.poisson3d(N); n = N**3
    for npass in (2, 3):
        h = mg.Hierarchy(A, 0.6, 1, 1).coarsen(10.0, npass, 8.0, 2500, 32).finalize()
        b = ctx.vec(n).rand(seed=1); y = ctx.vec(n)
        h.vcycle(b, y); h.vcycle(b, y)
        t = min(h.time_vcycle(b, y, reps=50) for _ in range(3))
        x = ctx.vec(n); st, it, tol = mg.bicgstab(A, x, b, h, 500, 1e-10)
        print(f"N={N} rows={n} npass={npass}: levels {[h.level_shape(l)[0] for l in range(h.nlev)]} cycle {t*1e3:.1f} us; bicgstab {it} its", flush=True)
        del h, b, y, x
