#!/usr/bin/env python3
"""CPU diagnosis (numpy, no GPU): how ill-conditioned is the K-cycle map b -> x?  Prints per level the inner products of the first
K-cycle call (rho1, gamma, beta, rho2 and the cancellation factor beta/rho2) and the relative change of the output under a 1e-16
relative perturbation of the input and under a different summation order of the dots, for the two ways of forming rho2:
  "diff"  : rho2 = beta - gamma^2/rho1        (round 3)
  "orth"  : w2 = (d2 - (gamma/rho1) d1), rho2 = w2 . (v2 - (gamma/rho1) v1)   (explicit orthogonalisation)
usage: kcycle_cond_cpu.py [N=64] [energy=1] [klevels=4] [omega=0.6] [op=poisson|convdiff]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle_py as orc  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
energy = int(sys.argv[2]) if len(sys.argv) > 2 else 1
klev = int(sys.argv[3]) if len(sys.argv) > 3 else 4
omega = float(sys.argv[4]) if len(sys.argv) > 4 else 0.6
opname = sys.argv[5] if len(sys.argv) > 5 else "poisson"
if opname == "convdiff":
    from multigridsolver_amd.synthetic import convdiff3d
    rp_, ci_, v_ = convdiff3d(N)
    A = orc.Csr.from_arrays(N ** 3, N ** 3, rp_, ci_, v_)
else:
    A = orc.poisson3d(N)
As, Ps = [A], []
while As[-1].shape[0] > 400 and len(As) < 12:
    P = As[-1].agmg(10.0, 2, 8.0, strict=False)
    if P.shape[1] == 0 or P.shape[1] > 0.9 * P.shape[0]:
        break
    Ps.append(P); As.append(As[-1].galerkin(P))
print("levels:", [a.shape[0] for a in As], flush=True)
S = [a.to_scipy().tocsr() for a in As]
PS = [p.to_scipy().tocsr() for p in Ps]
wd = [omega / s.diagonal() for s in S]
import scipy.sparse.linalg as spla
lu = spla.splu(S[-1].tocsc()) if S[-1].shape[0] <= 20000 else None      # above: 8 damped-Jacobi sweeps from zero, as mgs_hier_finalize
nlev = len(S)


def dot(a, b, mode):
    if mode == 0:
        return float(a @ b)
    # a different summation order: pairwise over chunks of 1000
    p = a * b
    m = (len(p) + 999) // 1000
    return float(np.add.reduce([np.add.reduce(p[i * 1000:(i + 1) * 1000]) for i in range(m)]))


log = []


def coarse(l, rhs, form, mode):
    if not (1 <= l <= klev and l < nlev - 1):
        return cyc(l, rhs, form, mode)
    c1 = cyc(l, rhs, form, mode); v1 = S[l] @ c1
    d1 = c1 if energy else v1
    rho1 = dot(d1, v1, mode); alpha1 = dot(d1, rhs, mode)
    a = alpha1 / rho1
    rp = rhs - a * v1
    c2 = cyc(l, rp, form, mode); v2 = S[l] @ c2
    d2 = c2 if energy else v2
    gamma = dot(d2, v1, mode)
    if form == "diff":
        beta = dot(d2, v2, mode); alpha2 = dot(d2, rp, mode)
        rho2 = beta - gamma * gamma / rho1
        k1 = alpha1 / rho1; k2 = 0.0
        if rho2 > 0:
            k2 = alpha2 / rho2; k1 -= gamma * k2 / rho1
        log.append((l, rho1, gamma, beta, rho2, beta / rho2 if rho2 else np.inf, alpha1, alpha2, k1, k2))
        return k1 * c1 + k2 * c2
    g = gamma / rho1
    c2o = c2 - g * c1; v2o = v2 - g * v1
    d2o = c2o if energy else v2o
    rho2 = dot(d2o, v2o, mode); alpha2 = dot(d2o, rp, mode)
    k1 = alpha1 / rho1; k2 = alpha2 / rho2 if rho2 > 0 else 0.0
    log.append((l, rho1, gamma, dot(d2, v2, mode), rho2, 0, alpha1, alpha2, k1, k2))
    return k1 * c1 + k2 * c2o


def cyc(l, b, form, mode):
    if l == nlev - 1:
        if lu is not None:
            return lu.solve(b)
        x = wd[l] * b
        for _ in range(7):
            x = x + wd[l] * (b - S[l] @ x)
        return x
    x = wd[l] * b
    r = b - S[l] @ x
    x = x + PS[l] @ coarse(l + 1, PS[l].T @ r, form, mode)
    return x + wd[l] * (b - S[l] @ x)


rng = np.random.default_rng(0)
b = orc.rand_rhs(A.shape[0])
for form in ("diff", "orth"):
    log.clear()
    x0 = cyc(0, b, form, 0)
    seen = set()
    for e in log:
        if e[0] not in seen:
            seen.add(e[0])
            print(f"  {form} level {e[0]}: rho1 {e[1]:.6e} gamma {e[2]:.6e} beta {e[3]:.6e} rho2 {e[4]:.6e} beta/rho2 {e[5]:.3e} "
                  f"a1 {e[6]:.3e} a2 {e[7]:.3e} k1 {e[8]:.4f} k2 {e[9]:.4f}")
    worst = max((e[5] for e in log), default=0)
    x1 = cyc(0, b * (1 + 1e-16 * rng.standard_normal(len(b))), form, 0)
    x2 = cyc(0, b, form, 1)
    nx = np.linalg.norm(x0)
    print(f"{form}: worst beta/rho2 {worst:.3e}; rel change under 1e-16 input perturbation {np.linalg.norm(x1 - x0) / nx:.3e}; "
          f"under another dot order {np.linalg.norm(x2 - x0) / nx:.3e}", flush=True)
xd = cyc(0, b, "diff", 0); xo = cyc(0, b, "orth", 0)
print(f"diff vs orth: {np.linalg.norm(xd - xo) / np.linalg.norm(xd):.3e}")
