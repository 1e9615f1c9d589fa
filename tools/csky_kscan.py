#!/usr/bin/env python3
"""csky3d at N^3: how many K-cycle levels (GCR form) pay.  K on every level costs 2^l visits of launch-bound levels at the bottom; K on the top
levels only may not converge.  Per K-level count: cycle ms, FGCR(10) / FGCR(30) / BiCGSTAB iterations and seconds to 1e-10 (one right-hand side).
usage: csky_kscan.py [N=256] [klevs=1,2,3,4,5,6] [maxit=300]"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MGS_ARENA_GB", "110")
import multigridsolver_amd as mg
from multigridsolver_amd import synthetic

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
klevs = [int(t) for t in (sys.argv[2] if len(sys.argv) > 2 else "1,2,3,4,5,6").split(",")]
maxit = int(sys.argv[3]) if len(sys.argv) > 3 else 300
n = N ** 3
rp, ci, v = synthetic.csky3d(N, rowsum_floor=synthetic.CSKY_ROWSUM_MARGIN)
ctx = mg.Context(0)
A = ctx.csr(n, n, rp, ci, v); del rp, ci, v
h = mg.Hierarchy(A, 0.6, 1, 1).coarsen(10.0, 2, 8.0, 2500, 32).finalize()
b = ctx.vec(n).rand(seed=0); nb = b.nrm2(); x = ctx.vec(n)
print(f"csky3d {N}^3: levels {[h.level_shape(l)[0] for l in range(h.nlev)]}", flush=True)
for _ in range(3): h.vcycle(b, x)
print(f"V-cycle {h.time_vcycle(b, x, reps=10):.3f} ms", flush=True)


def run(name, fn):
    x.fill(0.0); ctx.sync(); t0 = time.perf_counter()
    st, it, tol = fn()
    dt = time.perf_counter() - t0
    print(f"   {name}: status {st}, {it} iterations, {dt:.2f} s, true residual {A.residual(x, b).nrm2() / nb:.2e}", flush=True)


run("BiCGSTAB + V", lambda: mg.bicgstab(A, x, b, h, 1000, 1e-10))
for kl in klevs:
    if kl > h.nlev - 2: continue
    h.set_kcycle(kl)
    for _ in range(2): h.vcycle(b, x)
    print(f"K on {kl} levels: cycle {h.time_vcycle(b, x, reps=5):.2f} ms", flush=True)
    run("FGCR(10) + K", lambda: mg.fgcr(A, x, b, h, 10, maxit, 1e-10))
    run("FGCR(30) + K", lambda: mg.fgcr(A, x, b, h, 30, maxit, 1e-10))
    run("BiCGSTAB + K", lambda: mg.bicgstab(A, x, b, h, maxit, 1e-10))
    h.set_kcycle(0)
ctx.close()
