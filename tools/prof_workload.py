#!/usr/bin/env python3
"""Workload run under rocprofv3 (kernel-trace/--stats pass and separate --pmc passes):
a known-bytes calibration launch, then the fine-level hot-path kernels and a few V-cycles on
the 512^3 Poisson operator.  No CPU work, no oracle.  usage: prof_workload.py [grid] [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multigridsolver_amd as mg  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ctx = mg.Context(0)
n = N ** 3
A = ctx.poisson3d(N)
h = mg.Hierarchy(A, 0.6, 1, 1).coarsen(10.0, 2, 8.0, 1024, 32).finalize()
x = ctx.vec(n).rand(seed=1); b = ctx.vec(n).rand(seed=0); y = ctx.vec(n); dinv = A.diag_inv()
A.optimize()                 # pattern code of the column array (what hierarchies and solvers do for their operators)
ctx.set_option("graph", 0)   # eager launches so every kernel shows up as its own dispatch
# calibration: axpbypcz_kernel z = 2x + 3b + 0.5z reads 24n bytes and writes 8n bytes with 8-byte lanes
# (a kernel the V-cycle itself never launches, so every dispatch of it is a calibration launch)
ctx.set_option("blas1_vec", 0)       # the 8-byte-lane form the earlier summaries were calibrated on (the 16-byte form is what solves launch)
for _ in range(3):
    mg.lib().mgs_axpbypcz(2.0, x.h, 3.0, b.h, 0.5, y.h)
ctx.set_option("blas1_vec", 1)
for _ in range(reps):
    A.spmv(x, y)
for _ in range(reps):
    A.residual(x, b, y)
for _ in range(reps):
    A.jacobi(dinv, 0.6, b, x, y)
for _ in range(3):
    h.vcycle(b, y)
ctx.sync()
print("prof_workload done", N, n, A.nnz)
