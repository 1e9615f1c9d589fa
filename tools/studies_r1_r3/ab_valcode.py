#!/usr/bin/env python3
"""Opt-in value-pattern coding (option valcode) on the 512^3 operator: fine-level kernels and the V-cycle, default vs valcode."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import multigridsolver_amd as mg
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ctx = mg.Context(0); n = N ** 3
out = {}
for vc in (0, 1):
    ctx.set_option("valcode", vc)
    A = ctx.poisson3d(N)
    t0 = time.time()
    h = mg.Hierarchy(A, 0.6, 1, 1).coarsen(10.0, 2, 8.0, 2500, 32).finalize()
    b = ctx.vec(n).rand(seed=2); x = ctx.vec(n).rand(seed=1); y = ctx.vec(n); d = A.diag_inv()
    h.vcycle(b, y); ctx.sync(); t_setup = time.time() - t0
    res = {}
    for op, name in ((mg.OP_SPMV, "spmv"), (mg.OP_RESIDUAL, "residual"), (mg.OP_JACOBI, "jacobi")):
        A.time_kernel(op, x, b=b, dinv=d, out=y, reps=3)
        res[name] = A.time_kernel(op, x, b=b, dinv=d, out=y, reps=20)
    tv = min(h.time_vcycle(b, y, reps=20) for _ in range(2))
    h.vcycle(b, y); out[vc] = y.numpy()
    xs = ctx.vec(n); t0 = time.time(); st, it, tol = mg.bicgstab(A, xs, b, h, 1000, 1e-10); ts = time.time() - t0
    print(f"valcode={vc}: setup+first cycle {t_setup:.2f} s; " + " ".join(f"{k} {v:.3f} ms" for k, v in res.items()) +
          f" | vcycle {tv:.3f} ms = {1e3/tv:.1f} /s | bicgstab {it} its {ts:.2f} s | codes L0..3 " +
          str([h.level_A(l).rowcode_info()['coded_blocks'] for l in range(min(4, h.nlev - 1))]) + " fused L0 " + str(h.fused_info(0)), flush=True)
    del h, A, b, x, y, d, xs
ctx.set_option("valcode", 0)
print("same bits:", bool(np.array_equal(out[0], out[1])))
