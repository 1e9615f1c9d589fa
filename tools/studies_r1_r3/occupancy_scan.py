#!/usr/bin/env python3
"""SpMV time vs workgroups per CU (extra LDS padding limits residency): latency- or bandwidth-bound?"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import multigridsolver_amd as mg
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ctx = mg.Context(0)
A = ctx.poisson3d(N); n = N ** 3; nnz = A.nnz
x = ctx.vec(n).rand(seed=1); y = ctx.vec(n)
byts = 12 * nnz + 20 * n
base = A.plan_info()["lds_bytes"]
res = {}
for rnd in range(3):
    for pad in (0, 1500, 5000, 10500, 18500, 32000, 58000):
        ctx.set_option("lds_pad", pad)
        res.setdefault(pad, []).append(A.time_kernel(mg.OP_SPMV, x, out=y, reps=10))
ctx.set_option("lds_pad", 0)
for pad, t in res.items():
    wg = min(8, 163840 // (base + pad))
    print(f"lds/WG {base+pad:6d} B -> {wg} WG/CU ({4*wg:2d} waves): {np.median(t):.3f} ms  {byts/np.median(t)/1e6:.0f} GB/s")
