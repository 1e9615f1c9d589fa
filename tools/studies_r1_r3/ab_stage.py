#!/usr/bin/env python3
"""loop-free staging (option stage_unroll) on/off on ONE operator and hierarchy: plain CSR kernel, pattern-coded SpMV / residual /
Jacobi, the cycle.  usage: ab_stage.py [N=512]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import multigridsolver_amd as mg
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ctx = mg.Context(0); n = N ** 3
A = ctx.poisson3d(N)
h = mg.Hierarchy(A, 0.6, 1, 1).coarsen(10.0, 2, 8.0, 2500, 32).finalize()
b = ctx.vec(n).rand(seed=0); x = ctx.vec(n); y = ctx.vec(n); xs = ctx.vec(n).rand(seed=1); dinv = A.diag_inv()
A.optimize()
for _ in range(3): h.vcycle(b, x)
for rnd in range(2):
    for su in (1, 0):
        ctx.set_option("stage_unroll", su)
        ctx.set_option("rowcode", 0)
        A.time_kernel(mg.OP_SPMV, xs, out=y, reps=3); csr = min(A.time_kernel(mg.OP_SPMV, xs, out=y, reps=20) for _ in range(3))
        ctx.set_option("rowcode", 1)
        A.time_kernel(mg.OP_SPMV, xs, out=y, reps=3)
        sp = min(A.time_kernel(mg.OP_SPMV, xs, out=y, reps=20) for _ in range(3))
        rs = min(A.time_kernel(mg.OP_RESIDUAL, xs, b=b, out=y, reps=20) for _ in range(3))
        jc = min(A.time_kernel(mg.OP_JACOBI, xs, b=b, dinv=dinv, out=y, reps=20) for _ in range(3))
        h.vcycle(b, x); cy = min(h.time_vcycle(b, x, reps=20) for _ in range(3))
        print(f"stage_unroll={su}: plain CSR {csr:.3f}  coded SpMV {sp:.3f}  residual {rs:.3f}  Jacobi {jc:.3f}  cycle {cy:.3f} ms", flush=True)
