#!/usr/bin/env python3
"""A/B of the pattern-coded index: fine-level kernels and the V-cycle with rowcode on/off (one process)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import multigridsolver_amd as mg
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ctx = mg.Context(0)
A = ctx.poisson3d(N); n = N ** 3
t0 = time.time()
h = mg.Hierarchy(A, 0.6, 1, 1).coarsen(10.0, 2, 8.0, 2500, 32).finalize()
ctx.sync(); print(f"setup {time.time()-t0:.2f} s, levels {[h.level_shape(l)[0] for l in range(h.nlev)]}", flush=True)
b = ctx.vec(n).rand(seed=2); x = ctx.vec(n).rand(seed=1); y = ctx.vec(n); d = A.diag_inv()
t0 = time.time(); h.vcycle(b, y); ctx.sync(); print(f"first cycle (builds operands + codes) {time.time()-t0:.2f} s", flush=True)
for l in range(min(h.nlev - 1, 5)):
    print(f"  L{l} rowcode {h.level_A(l).rowcode_info()}", flush=True)
for rnd in range(2):
    for rc in (0, 1):
        ctx.set_option("rowcode", rc)
        res = {}
        for op, name in ((mg.OP_SPMV, "spmv"), (mg.OP_RESIDUAL, "residual"), (mg.OP_JACOBI, "jacobi")):
            res[name] = A.time_kernel(op, x, b=b, dinv=d, out=y, reps=20)
        tv = h.time_vcycle(b, y, reps=20) if hasattr(h, "time_vcycle") else float("nan")
        print(f"rowcode={rc}: " + " ".join(f"{k} {v:.3f} ms" for k, v in res.items()) + f" | vcycle {tv:.3f} ms", flush=True)
ctx.set_option("rowcode", 1)
for rnd in range(2):
    for nts in (0, 1):
        ctx.set_option("nt_store", nts)
        res = {}
        for op, name in ((mg.OP_SPMV, "spmv"), (mg.OP_RESIDUAL, "residual"), (mg.OP_JACOBI, "jacobi")):
            res[name] = A.time_kernel(op, x, b=b, dinv=d, out=y, reps=20)
        tv = h.time_vcycle(b, y, reps=20)
        print(f"nt_store={nts}: " + " ".join(f"{k} {v:.3f} ms" for k, v in res.items()) + f" | vcycle {tv:.3f} ms", flush=True)
ctx.set_option("nt_store", 0)
