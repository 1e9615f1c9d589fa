#!/usr/bin/env python3
"""fine scan of the pattern-coded fine-level SpMV against the byte offset of y (4 KiB steps over 512 KiB, then 64 KiB steps over
8 MiB); x fixed.  usage: placement_scan2.py [N=512]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import multigridsolver_amd as mg
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ctx = mg.Context(0)
n = N ** 3
A = ctx.poisson3d(N); A.optimize()
PAD = 64 << 20
pool = ctx.vec(2 * n + 4 * (PAD // 8)).rand(seed=0)
base = pool.ptr
x = mg.Vec.wrap(ctx, base, n)


def t_spmv(yo):
    y = mg.Vec.wrap(ctx, base + PAD + 8 * n + PAD // 2 + yo, n)
    A.time_kernel(0, x, out=y, reps=2)
    return min(A.time_kernel(0, x, out=y, reps=8) for _ in range(2))


print("4 KiB steps:")
row = []
for k in range(128):
    row.append(t_spmv(k * 4096))
    if len(row) == 16:
        print(f"{(k - 15) * 4:>6d} KiB " + " ".join(f"{v:.3f}" for v in row), flush=True); row = []
print("64 KiB steps:")
for k in range(0, 128, 16):
    print(f"{k * 64:>6d} KiB " + " ".join(f"{t_spmv((k + j) * 65536):.3f}" for j in range(16)), flush=True)
