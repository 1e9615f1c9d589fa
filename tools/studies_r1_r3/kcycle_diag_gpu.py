#!/usr/bin/env python3
"""GPU diagnosis of the K-cycle: on the device-built hierarchy of the N^3 Poisson operator
  (1) one K-cycle application, GPU against the oracle's restatement on the SAME hierarchy (N <= 160 only);
  (2) FGCR(10) iterations to 1e-10 with V(1,1) and with K-cycles on 1, 2, 4, all coarse levels; BiCGSTAB + V beside them.
usage: kcycle_diag_gpu.py N [omega=0.6] [coarse_rows=2500]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import multigridsolver_amd as mg  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
omega = float(sys.argv[2]) if len(sys.argv) > 2 else 0.6
coarse_rows = int(sys.argv[3]) if len(sys.argv) > 3 else 2500
ctx = mg.Context(0)
A = ctx.poisson3d(N); n = N ** 3
h = mg.Hierarchy(A, omega, 1, 1).coarsen(10.0, 2, 8.0, coarse_rows, 32).finalize()
print("levels:", [h.level_shape(l)[0] for l in range(h.nlev)], flush=True)
b = ctx.vec(n).rand(seed=0)
nlev = h.nlev
if N <= 160:
    import scipy.sparse as sps
    from oracle import oracle_py as orc
    As, Ps = [], []
    for l in range(nlev):
        rp, ci, v = h.level_A(l).download(); rows = h.level_shape(l)[0]
        As.append(orc.Csr.from_arrays(rows, rows, rp, ci, v))
        if l < nlev - 1:
            T = h.level_P(l); agg = T.agg(); nf, nc = T.shape; r = np.nonzero(agg >= 0)[0]
            Ps.append(orc.Csr.from_scipy(sps.csr_matrix((np.ones(r.size), (r, agg[r])), shape=(nf, nc))))
    ho = orc.Hier(As[0], Ps, omega=omega, nu1=1, nu2=1, As=As)
    bn = b.numpy()
    for kl in (0, 1, 2, nlev - 2):
        h.set_kcycle(kl); ho.set_kcycle(kl)
        xg = h.vcycle(b).numpy(); xo = ho.vcycle(bn)
        print(f"K-cycle levels {kl}: GPU vs oracle rel err {np.linalg.norm(xg - xo) / np.linalg.norm(xo):.2e}", flush=True)
for energy in (0, 1):
    ctx.set_option("kcycle_energy", energy)
    for kl in ((0, 1, 2, 3, 4, nlev - 2) if energy == 0 else (1, 2, 3, 4, nlev - 2)):
        h.set_kcycle(kl)
        x = ctx.vec(n)
        ms = h.time_vcycle(b, x, reps=3)
        ms = h.time_vcycle(b, x, reps=5)
        x.fill(0.0); ctx.sync()
        t0 = time.perf_counter()
        st, it, tol = mg.fgcr(A, x, b, h, 10, 400, 1e-10)
        dt = time.perf_counter() - t0
        true = A.residual(x, b).nrm2() / b.nrm2()
        print(f"N={N} FGCR(10) + {'V' if kl == 0 else 'K x%d' % kl}{' energy' if energy else ''}: status {st} iterations {it} tol {tol:.2e} true {true:.2e}, {dt:.3f} s, {ms:.3f} ms per cycle", flush=True)
ctx.set_option("kcycle_energy", 0)
h.set_kcycle(0)
x = ctx.vec(n)
t0 = time.perf_counter()
st, it, tol = mg.bicgstab(A, x, b, h, 400, 1e-10)
print(f"N={N} BiCGSTAB + V: status {st} iterations {it} tol {tol:.2e}, {time.perf_counter() - t0:.3f} s", flush=True)
