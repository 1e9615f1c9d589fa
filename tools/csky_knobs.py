#!/usr/bin/env python3
"""csky3d at N^3, BiCGSTAB + V-cycle: iterations and seconds against the two knobs the reference does not have (smoother damping omega, over-correction
sigma of x += sigma P e_c).  usage: csky_knobs.py [N=256] [maxit=600]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MGS_ARENA_GB", "110")
import multigridsolver_amd as mg
from multigridsolver_amd import synthetic

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
maxit = int(sys.argv[2]) if len(sys.argv) > 2 else 600
n = N ** 3
rp, ci, v = synthetic.csky3d(N, rowsum_floor=synthetic.CSKY_ROWSUM_MARGIN)
ctx = mg.Context(0)
A = ctx.csr(n, n, rp, ci, v); del rp, ci, v
h = mg.Hierarchy(A, 0.6, 1, 1).coarsen(10.0, 2, 8.0, 2500, 32).finalize()
b = ctx.vec(n).rand(seed=0); nb = b.nrm2(); x = ctx.vec(n)
print(f"csky3d {N}^3: levels {[h.level_shape(l)[0] for l in range(h.nlev)]}", flush=True)
for omega in (0.6, 0.7, 0.8, 0.9, 1.0):
    for sigma in (1.0, 1.3, 1.6, 2.0):
        h.set_smoother(omega, 1, 1).set_correction_scale(sigma)
        h.vcycle(b, x)
        x.fill(0.0); ctx.sync(); t0 = time.perf_counter()
        st, it, tol = mg.bicgstab(A, x, b, h, maxit, 1e-10)
        dt = time.perf_counter() - t0
        print(f"omega {omega} sigma {sigma}: status {st}, {it} iterations, {dt:.2f} s, true residual {A.residual(x, b).nrm2() / nb:.2e}", flush=True)
ctx.close()
