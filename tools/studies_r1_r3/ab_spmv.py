#!/usr/bin/env python3
"""A/B of the fine-level SpMV kernel variants in ONE process, interleaved rounds (guide rule 24).
usage: ab_spmv.py [grid]"""
import itertools
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import multigridsolver_amd as mg  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ctx = mg.Context(0)
n = N ** 3
A = ctx.poisson3d(N)
nnz = A.nnz
x = ctx.vec(n).rand(seed=1); y = ctx.vec(n); b = ctx.vec(n).rand(seed=2); dinv = A.diag_inv()
byts = 12 * nnz + 20 * n + 4
variants = [(5, 64, 0), (5, 64, 0, 0)]
ref = None
times = {k: [] for k in variants}
for rnd in range(6):
    for var in variants:
        v, s, nt = var[:3]
        ctx.set_option("blkptr", var[3] if len(var) > 3 else 1)
        ctx.set_option("spmv_variant", v); ctx.set_option("strip", s); ctx.set_option("nontemporal", nt)
        if rnd == 0:
            A.spmv(x, y); out = y.numpy()
            if ref is None:
                ref = out
            assert np.array_equal(ref, out), (v, s, nt)
        times[var].append(A.time_kernel(mg.OP_SPMV, x, out=y, reps=10))
print(f"grid {N}^3  algorithmic bytes {byts/1e9:.3f} GB")
print("chunk strip nt   median_ms   min_ms   GB/s(median)  frac_of_8TB/s")
for var, t in sorted(times.items(), key=lambda kv: np.median(kv[1])):
    med, mn = float(np.median(t)), float(min(t))
    print(f"{str(var):20s} {med:10.3f} {mn:8.3f} {byts/med/1e6:12.0f} {byts/med/1e6/8000:10.3f}")
ctx.set_option("blkptr", 1)
# jacobi / residual with the default (auto) settings
ctx.set_option("spmv_variant", 0); ctx.set_option("strip", -1); ctx.set_option("nontemporal", 1)
for name, op, bb in (("spmv", mg.OP_SPMV, 12 * nnz + 20 * n), ("residual", mg.OP_RESIDUAL, 12 * nnz + 28 * n), ("jacobi", mg.OP_JACOBI, 12 * nnz + 36 * n)):
    t = [A.time_kernel(op, x, b=b, dinv=dinv, out=y, reps=10) for _ in range(3)]
    print(f"default {name}: {np.median(t):.3f} ms  {bb/np.median(t)/1e6:.0f} GB/s")
