#!/usr/bin/env python3
"""Cycle and coded-SpMV time over several values of one integer option, one process, one hierarchy (three rounds).
usage: sweep_option.py OPTION N v1 v2 ..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multigridsolver_amd as mg
opt = sys.argv[1]; N = int(sys.argv[2]); vals = [int(v) for v in sys.argv[3:]]
ctx = mg.Context(0); n = N ** 3
A = ctx.poisson3d(N)
h = mg.Hierarchy(A, 0.6, 1, 1).coarsen(10.0, 2, 8.0, 2500, 32).finalize()
b = ctx.vec(n).rand(seed=0); x = ctx.vec(n); y = ctx.vec(n); xs = ctx.vec(n).rand(seed=1)
A.optimize()
for _ in range(3): h.vcycle(b, x)
for rnd in range(3):
    for v in vals:
        ctx.set_option(opt, v)
        A.time_kernel(mg.OP_SPMV, xs, out=y, reps=3)
        sp = min(A.time_kernel(mg.OP_SPMV, xs, out=y, reps=20) for _ in range(3))
        h.vcycle(b, x); cy = min(h.time_vcycle(b, x, reps=20) for _ in range(3))
        print(f"{opt}={v}: coded SpMV {sp:.3f}  cycle {cy:.3f} ms", flush=True)
ctx.set_option(opt, vals[0])
