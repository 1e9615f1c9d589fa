#!/usr/bin/env python3
"""fine-level SpMV time against WHERE its vectors lie: x and y are views into one large buffer at a sweep of byte offsets (the
operator stays where it is).  If the time moves with the offsets, placement is a tuning knob; if not, the ±4 % seen between
hierarchies comes from the operator's own arrays.  usage: placement_scan.py [N=512]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import multigridsolver_amd as mg
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ctx = mg.Context(0)
n = N ** 3
A = ctx.poisson3d(N)
PAD = 64 << 20
pool = ctx.vec(2 * n + 4 * (PAD // 8)).rand(seed=0)
base = pool.ptr


def t_spmv(xo, yo):
    x = mg.Vec.wrap(ctx, base + xo, n)
    y = mg.Vec.wrap(ctx, base + PAD + 8 * n + PAD // 2 + yo, n)
    A.time_kernel(0, x, out=y, reps=3)
    return min(A.time_kernel(0, x, out=y, reps=10) for _ in range(3))


offs = [0, 256, 4096, 64 << 10, 1 << 20, (2 << 20) + 4096, 8 << 20, (16 << 20) + 12288, 32 << 20]
print("rows: x offset, columns: y offset (bytes); SpMV ms")
print(" " * 10 + "".join(f"{o:>10d}" for o in offs))
for xo in offs:
    print(f"{xo:>10d}" + "".join(f"{t_spmv(xo, yo):10.3f}" for yo in offs), flush=True)
# the same operator built again lands elsewhere: how far does that move the time with x/y fixed?
for k in range(4):
    B = ctx.poisson3d(N)
    x = mg.Vec.wrap(ctx, base, n); y = mg.Vec.wrap(ctx, base + PAD + 8 * n + PAD // 2, n)
    B.time_kernel(0, x, out=y, reps=3)
    print(f"operator instance {k}: {min(B.time_kernel(0, x, out=y, reps=10) for _ in range(3)):.3f} ms", flush=True)
    if k % 2: keep = B
