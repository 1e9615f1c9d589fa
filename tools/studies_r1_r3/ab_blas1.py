#!/usr/bin/env python3
"""BLAS-1 of the Krylov step at N^3: the 16-byte update kernels (option blas1_vec) against the 8-byte grid-stride loops —
ms and TB/s of axpby (24n B) and axpbypcz (32n B) alone, then BiCGSTAB + V-cycle ms per iteration with the option on/off.
usage: ab_blas1.py [N=512]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import multigridsolver_amd as mg
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ctx = mg.Context(0)
n = N ** 3
x = ctx.vec(n).rand(seed=1); y = ctx.vec(n).rand(seed=2); z = ctx.vec(n).rand(seed=3)


def timed(f, reps=20):
    f(); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    ctx.sync()
    return (time.perf_counter() - t0) / reps * 1e3


for rnd in range(2):
    for opt, nt in ((1, 1000000), (1, 0), (0, 0)):
        ctx.set_option("blas1_vec", opt); ctx.set_option("nt_store", nt)
        t1 = timed(lambda: y.axpby(0.5, x, 0.25))
        t2 = timed(lambda: z.axpbypcz(0.5, x, 0.25, y, 0.125))
        print(f"blas1_vec={opt} nt_store={nt}: axpby {t1:.3f} ms = {24 * n / t1 / 1e9:.2f} TB/s; axpbypcz {t2:.3f} ms = {32 * n / t2 / 1e9:.2f} TB/s", flush=True)
ctx.set_option("nt_store", 1000000)
A = ctx.poisson3d(N)
# fixed number of unpreconditioned iterations: 2 SpMV (+ dots) and the four update kernels each, nothing else
for rnd in range(2):
    for opt in (1, 0):
        ctx.set_option("blas1_vec", opt)
        xs = ctx.vec(n); b0 = ctx.vec(n).rand(seed=0); mg.bicgstab(A, xs, b0, None, 2, 1e-30); xs = ctx.vec(n); ctx.sync()
        t0 = time.perf_counter()
        st, it, tol = mg.bicgstab(A, xs, b0, None, 20, 1e-30)
        dt = time.perf_counter() - t0
        print(f"blas1_vec={opt}: unpreconditioned, {it} iterations, {dt / it * 1e3:.3f} ms per iteration", flush=True)
h = mg.Hierarchy(A, 0.6, 1, 1).coarsen(10.0, 2, 8.0, 2500, 32).finalize()
b = ctx.vec(n).rand(seed=0)
xs = ctx.vec(n); mg.bicgstab(A, xs, b, h, 5, 1e-30)        # warm: codes, graphs
for rnd in range(2):
    for opt in (1, 0):
        ctx.set_option("blas1_vec", opt)
        xs = ctx.vec(n); ctx.sync()
        t0 = time.perf_counter()
        st, it, tol = mg.bicgstab(A, xs, b, h, 300, 1e-10)
        dt = time.perf_counter() - t0
        true = A.residual(xs, b).nrm2() / b.nrm2()
        print(f"blas1_vec={opt}: status {st}, {it} iterations, {dt:.3f} s, {dt / it * 1e3:.2f} ms per iteration, true residual {true:.2e}", flush=True)
