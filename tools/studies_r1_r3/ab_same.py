#!/usr/bin/env python3
"""A/B of a launch-time option on ONE hierarchy (no rebuild between the variants, so where the operators landed in HBM is the
same for both): alternating timed cycles and fine-level SpMVs.  Only for options the kernels read at launch (rowoff16, rowcode,
diag_from_values, fuse_dots, group_sweep ...); options that change what setup builds need tools/ab_group2.py.
usage: ab_same.py <grid> <option> [value_a=1] [value_b=0] [more values ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import multigridsolver_amd as mg
N = int(sys.argv[1]); key = sys.argv[2]
vals = [int(v) for v in sys.argv[3:]] or [1, 0]
if len(vals) == 1: vals.append(0)
va = vals[0]
ctx = mg.Context(0)
A = ctx.poisson3d(N); n = N ** 3
b = ctx.vec(n).rand(seed=0)
for inst in range(int(os.environ.get("AB_INSTANCES", "2"))):
    ctx.set_option(key, va)
    h = mg.Hierarchy(A, 0.6, 1, 1).coarsen(10.0, 2, 8.0, 2500, 32).finalize()
    x = ctx.vec(n); y = ctx.vec(n)
    for _ in range(3): h.vcycle(b, x)
    cyc = {v: [] for v in vals}; sp = {v: [] for v in vals}
    for rnd in range(6):
        for v in vals:
            ctx.set_option(key, v)
            h.vcycle(b, x); A.time_kernel(mg.OP_SPMV, b, out=y, reps=3)
            cyc[v].append(h.time_vcycle(b, x, reps=20))
            sp[v].append(A.time_kernel(mg.OP_SPMV, b, out=y, reps=50))
    for v in vals:
        print(f"instance {inst} {key}={v}: cycle min {min(cyc[v]):.3f} med {sorted(cyc[v])[3]:.3f} ms   spmv min {min(sp[v]):.4f} med {sorted(sp[v])[3]:.4f} ms")
    del h, x, y
