#!/usr/bin/env python3
"""Round 4: the reference's problem class at scale.  Hierarchy and Krylov iterations on the nonsymmetric stand-ins of
multigridsolver_amd/synthetic.py — `csky3d` (the bundled CSky3d30's family, with its row-sum margin; `csky3d_printed`: the printed digits alone) and
`convdiff3d` (rotating flow, column jumps; the 80^3 stand-in of the tests) — BiCGSTAB + V against FGCR(10) + K-cycle (GCR form on 1 level /
on all levels).  usage: convdiff_scan.py [N=128] [maxit=400] [families=csky3d,convdiff3d]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multigridsolver_amd as mg
from multigridsolver_amd import synthetic

N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
maxit = int(sys.argv[2]) if len(sys.argv) > 2 else 400
fams = (sys.argv[3] if len(sys.argv) > 3 else "csky3d,convdiff3d").split(",")


def gen(fam, N):
    if fam == "csky3d":
        return synthetic.csky3d(N, rowsum_floor=synthetic.CSKY_ROWSUM_MARGIN)
    if fam == "csky3d_printed":
        return synthetic.csky3d(N)
    return getattr(synthetic, fam)(N)


ctx = mg.Context(0)
n = N ** 3
for fam in fams:
    t0 = time.perf_counter(); rp, ci, v = gen(fam, N); tg = time.perf_counter() - t0
    A = ctx.csr(n, n, rp, ci, v); del rp, ci, v
    for omega in ((0.6, 0.8) if os.environ.get('SCAN_OMEGAS') is None else tuple(float(t) for t in os.environ['SCAN_OMEGAS'].split(','))):
        h = mg.Hierarchy(A, omega, 1, 1).coarsen(10.0, 2, 8.0, 2500, 32).finalize()
        b = ctx.vec(n).rand(seed=0); nb = b.nrm2(); x = ctx.vec(n)
        for _ in range(3): h.vcycle(b, x)
        ms = h.time_vcycle(b, x, reps=10)
        print(f"{fam} {N}^3 omega={omega}: generated in {tg:.1f}s; levels {[h.level_shape(l)[0] for l in range(h.nlev)]}; cycle {ms:.3f} ms", flush=True)
        x.fill(0.0); ctx.sync(); t0 = time.perf_counter()
        st, it, tol = mg.bicgstab(A, x, b, h, maxit, 1e-10)
        print(f"   BiCGSTAB+V: status {st}, {it} iterations, {time.perf_counter() - t0:.2f}s, true residual {A.residual(x, b).nrm2() / nb:.2e}", flush=True)
        for kl in sorted({1, 2, max(h.nlev - 2, 1)}):
            h.set_kcycle(kl)
            x.fill(0.0); ctx.sync(); t0 = time.perf_counter()
            st, it, tol = mg.fgcr(A, x, b, h, 10, maxit, 1e-10)
            print(f"   FGCR(10)+K(GCR form, {kl} levels): status {st}, {it} iterations, {time.perf_counter() - t0:.2f}s, true residual {A.residual(x, b).nrm2() / nb:.2e}", flush=True)
            x.fill(0.0); ctx.sync(); t0 = time.perf_counter()
            st, it, tol = mg.bicgstab(A, x, b, h, maxit, 1e-10)
            print(f"   BiCGSTAB+K(GCR form, {kl} levels; nonlinear preconditioner): status {st}, {it} iterations, {time.perf_counter() - t0:.2f}s, true residual {A.residual(x, b).nrm2() / nb:.2e}", flush=True)
            h.set_kcycle(0)
        x.fill(0.0); ctx.sync(); t0 = time.perf_counter()
        st, it, tol = mg.fgcr(A, x, b, h, 30, maxit, 1e-10)
        print(f"   FGCR(30)+V: status {st}, {it} iterations, {time.perf_counter() - t0:.2f}s, true residual {A.residual(x, b).nrm2() / nb:.2e}", flush=True)
        del h, b, x
    del A
ctx.close()
