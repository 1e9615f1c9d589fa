#!/usr/bin/env python3
"""the same solve again and again in one process: ms per BiCGSTAB iteration of every call (the solver creates and destroys its eight
work vectors per call).  usage: solve_repeat.py [N=512] [calls=8]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import multigridsolver_amd as mg
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ctx = mg.Context(0)
for kv in os.environ.get("OPTS", "").split(","):
    if kv: ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
n = N ** 3
A = ctx.poisson3d(N)
h = mg.Hierarchy(A, 0.6, 1, 1).coarsen(10.0, 2, 8.0, 2500, 32).finalize()
b = ctx.vec(n).rand(seed=0)
xs = ctx.vec(n); mg.bicgstab(A, xs, b, h, 3, 1e-30)
time.sleep(float(os.environ.get("SLEEP", "0")))
for c in range(calls):
    if os.environ.get("NEWX", "1") == "1": xs = ctx.vec(n)
    else: xs.fill(0.0)
    ctx.sync(); t0 = time.perf_counter()
    st, it, tol = mg.bicgstab(A, xs, b, h, 20, 1e-30)
    dt = time.perf_counter() - t0
    print(f"call {c}: {it} iterations, {dt / it * 1e3:.2f} ms per iteration, x@{xs.ptr:#x}", flush=True)
# the bundled-size case: host round trips dominate the iteration
from oracle import oracle_py as orc
A_o = orc.poisson2d(100)
As = ctx.csr(A_o.shape[0], A_o.shape[1], A_o.rowptr, A_o.col, A_o.val)
hs = mg.Hierarchy(As, 0.6, 1, 1).coarsen(10.0, 2, 8.0, 400, 32).finalize()
bs = ctx.vec(orc.rand_rhs(A_o.shape[0]))
for opt in (1, 0, 1, 0):
    ctx.set_option("post_results", opt)
    best = 1e9
    for rep in range(5):
        xq = ctx.vec(A_o.shape[0]); ctx.sync(); t0 = time.perf_counter()
        st, it, tol = mg.bicgstab(As, xq, bs, hs, 200, 1e-10)
        best = min(best, time.perf_counter() - t0)
    print(f"poisson 100^2, post_results={opt}: status {st}, {it} iterations, {best * 1e3:.3f} ms per solve, {best / it * 1e6:.1f} us per iteration", flush=True)
