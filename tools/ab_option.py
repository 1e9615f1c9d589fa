#!/usr/bin/env python3
"""One kernel option on/off on ONE operator and hierarchy (same process, same allocations): pattern-coded SpMV / residual / Jacobi and the
cycle, two rounds.  usage: ab_option.py OPTION [N=512] [on=1] [off=0]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multigridsolver_amd as mg
opt = sys.argv[1]
N = int(sys.argv[2]) if len(sys.argv) > 2 else 512
on = int(sys.argv[3]) if len(sys.argv) > 3 else 1
off = int(sys.argv[4]) if len(sys.argv) > 4 else 0
ctx = mg.Context(0); n = N ** 3
A = ctx.poisson3d(N)
h = mg.Hierarchy(A, 0.6, 1, 1).coarsen(10.0, 2, 8.0, 2500, 32).finalize()
b = ctx.vec(n).rand(seed=0); x = ctx.vec(n); y = ctx.vec(n); xs = ctx.vec(n).rand(seed=1); dinv = A.diag_inv()
A.optimize()
for _ in range(3): h.vcycle(b, x)
ref = None
for rnd in range(3):
    for v in (on, off):
        ctx.set_option(opt, v)
        A.time_kernel(mg.OP_SPMV, xs, out=y, reps=3)
        sp = min(A.time_kernel(mg.OP_SPMV, xs, out=y, reps=20) for _ in range(3))
        rs = min(A.time_kernel(mg.OP_RESIDUAL, xs, b=b, out=y, reps=20) for _ in range(3))
        jc = min(A.time_kernel(mg.OP_JACOBI, xs, b=b, dinv=dinv, out=y, reps=20) for _ in range(3))
        h.vcycle(b, x); cy = min(h.time_vcycle(b, x, reps=20) for _ in range(3))
        nrm = x.nrm2()
        if ref is None: ref = nrm
        print(f"{opt}={v}: coded SpMV {sp:.3f}  residual {rs:.3f}  Jacobi {jc:.3f}  cycle {cy:.3f} ms  (|x| {'same bits' if nrm == ref else 'DIFFERS'})", flush=True)
ctx.set_option(opt, on)
