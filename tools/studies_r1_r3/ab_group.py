#!/usr/bin/env python3
"""A/B of the grouped pre pass (option fuse_restrict: restriction inside the pre pass, post pass in t-form) on the N^3 Poisson
hierarchy: cycle time on/off, agreement of the two forms, per-level group statistics.  usage: ab_group.py [grid=512]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import multigridsolver_amd as mg
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ctx = mg.Context(0)
A = ctx.poisson3d(N); n = N ** 3
h = mg.Hierarchy(A, 0.6, 1, 1).coarsen(10.0, 2, 8.0, 2500, 32).finalize()
b = ctx.vec(n).rand(seed=0); x1 = ctx.vec(n); x0 = ctx.vec(n)
res = {}
for opt in (1, 0, 1, 0):
    ctx.set_option("fuse_restrict", opt)
    x = x1 if opt else x0
    for _ in range(3): h.vcycle(b, x)
    res.setdefault(opt, []).append(min(h.time_vcycle(b, x, reps=20) for _ in range(3)))
a, c = x1.numpy(), x0.numpy()
print("levels", [h.level_shape(l)[0] for l in range(h.nlev)])
print("groups", [h.group_info(l) for l in range(h.nlev - 1)])
print(f"cycle ms: grouped {min(res[1]):.3f}  separate {min(res[0]):.3f}   rel diff {np.linalg.norm(a - c) / np.linalg.norm(c):.2e}")
ctx.set_option("fuse_restrict", 1)
xs = ctx.vec(n); st, it, tol = mg.bicgstab(A, xs, b, h, 300, 1e-10)
print("bicgstab grouped:", st, it, tol, "true residual", A.residual(xs, b).nrm2() / b.nrm2())
