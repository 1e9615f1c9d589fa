#!/usr/bin/env python3
"""CPU diagnosis of the K-cycle (oracle only, no GPU): outer iterations to 1e-10 on the N^3 7-point Poisson operator for
  two-grid with an exact coarse solve | V(1,1) | K-cycle on k levels
as preconditioner of (a) restarted FGCR(m), (b) truncated (sliding-window) FGCR(m), (c) BiCGSTAB (linear preconditioners only).
usage: kcycle_diag_cpu.py [N=48] [omega=0.6] [npass=2]"""
import os
import sys
import time

import numpy as np
import scipy.sparse as sps
import scipy.sparse.linalg as spla

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle_py as orc  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 48
omega = float(sys.argv[2]) if len(sys.argv) > 2 else 0.6
npass = int(sys.argv[3]) if len(sys.argv) > 3 else 2
A = orc.poisson3d(N)
n = A.shape[0]
As, Ps = [A], []
while As[-1].shape[0] > 400 and len(As) < 12:
    P = As[-1].agmg(10.0, npass, 8.0, strict=False)
    if P.shape[1] == 0 or P.shape[1] > 0.9 * P.shape[0]:
        break
    Ps.append(P); As.append(As[-1].galerkin(P))
print("levels:", [a.shape[0] for a in As], flush=True)
b = orc.rand_rhs(n)
Asp = A.to_scipy().tocsr()


def fgcr(prec, m, window, tol=1e-10, maxit=400):
    x = np.zeros(n); r = b.copy(); nb = np.linalg.norm(b)
    Cs, Vs, rho = [], [], []
    for it in range(1, maxit + 1):
        c = prec(r); v = Asp @ c
        for cj, vj, rj in zip(Cs, Vs, rho):
            beta = (vj @ v) / rj
            v = v - beta * vj; c = c - beta * cj
        rh = v @ v
        al = (v @ r) / rh
        x += al * c; r -= al * v
        Cs.append(c); Vs.append(v); rho.append(rh)
        if window:
            if len(Cs) > m - 1:
                Cs.pop(0); Vs.pop(0); rho.pop(0)
        elif len(Cs) == m:
            Cs, Vs, rho = [], [], []
        if np.linalg.norm(r) / nb < tol:
            rt = np.linalg.norm(b - Asp @ x) / nb
            if rt < tol:
                return it, rt
            r = b - Asp @ x
    return maxit, np.linalg.norm(b - Asp @ x) / nb


def twogrid():
    P0 = Ps[0].to_scipy().tocsr()
    Ac = (P0.T @ Asp @ P0).tocsc()
    lu = spla.splu(Ac)
    wd = omega / Asp.diagonal()

    def app(v):     # zero-guess V(1,1) two-grid, exact coarse solve
        x = wd * v
        r = v - Asp @ x
        x = x + P0 @ lu.solve(P0.T @ r)
        return x + wd * (v - Asp @ x)
    return app


H = orc.Hier(As[0], Ps, omega=omega, nu1=1, nu2=1, As=As)
rows = []
for name, prec, linear in [("two-grid exact", twogrid(), True), ("V(1,1)", lambda v: H.set_kcycle(0).vcycle(v), True)] + \
        [(f"K-cycle x{k}" + (" energy" if e else ""), (lambda k, e: (lambda v: H.set_kcycle_energy(e).set_kcycle(k).vcycle(v)))(k, e), False)
         for e in (0, 1) for k in (1, 2, len(As) - 2)]:
    t0 = time.time()
    r1 = fgcr(prec, 10, False); r2 = fgcr(prec, 10, True); r3 = fgcr(prec, 30, True)
    bi = None
    if linear:
        st, it, tol, _ = orc.bicgstab(A, b, prec, 400, 1e-10)
        bi = (it, st)
    rows.append((name, r1, r2, r3, bi))
    print(f"{name:22s} FGCR(10) restart {r1[0]:4d}  window(10) {r2[0]:4d}  window(30) {r3[0]:4d}  BiCGSTAB {bi}   [{time.time() - t0:.1f}s]", flush=True)
