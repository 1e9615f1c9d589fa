#!/usr/bin/env python3
"""coarse placement: the pattern-coded fine-level SpMV with y (then x) at offsets of k GiB inside one 48 GiB pool, everything else
fixed.  Between hierarchies of one process the SpMV moved by up to 12 % with where its output vector had landed; this scan shows
whether that is the output's position at GiB scale.  usage: placement_scan3.py [N=512]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import multigridsolver_amd as mg
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ctx = mg.Context(0)
ctx.set_option("nt_store", int(os.environ.get("NT", "1")))
n = N ** 3
A = ctx.poisson3d(N); A.optimize()
GIB = 1 << 30
pool = ctx.vec(48 * GIB // 8)
base = pool.ptr
x0 = ctx.vec(n).rand(seed=0)
print(f"pool @ {base:#x}, x0 @ {x0.ptr:#x}")


def t_spmv(x, yptr):
    y = mg.Vec.wrap(ctx, yptr, n)
    A.time_kernel(0, x, out=y, reps=2)
    return min(A.time_kernel(0, x, out=y, reps=8) for _ in range(2))


print("y at base + k GiB (x fixed outside the pool):")
print(" ".join(f"{t_spmv(x0, base + k * GIB):.3f}" for k in range(0, 46)), flush=True)
print("y at base + k GiB + 512 MiB:")
print(" ".join(f"{t_spmv(x0, base + k * GIB + GIB // 2):.3f}" for k in range(0, 46)), flush=True)
print("y at base + k * 128 MiB, k = 0..63:")
print(" ".join(f"{t_spmv(x0, base + k * (GIB // 8)):.3f}" for k in range(0, 64)), flush=True)
print("x inside the pool at base + k GiB (copied there), y fixed at base + 46 GiB:")
row = []
for k in range(0, 44, 2):
    xv = mg.Vec.wrap(ctx, base + k * GIB, n); xv.copy_from(x0)
    row.append(t_spmv(xv, base + 46 * GIB))
print(" ".join(f"{v:.3f}" for v in row), flush=True)
