#!/usr/bin/env python3
"""does the allocation history of the process move the cycle time?  Builds the N^3 hierarchy (a) in a fresh process state, (b) after
large vectors were allocated and freed, (c) with filler allocations of odd sizes alive; prints cycle and fine-level SpMV times and the
device addresses of the work vectors.  usage: alloc_effect.py [N=512]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import multigridsolver_amd as mg
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ctx = mg.Context(0)
n = N ** 3


def build(tag, A=None):
    A = A or ctx.poisson3d(N)
    h = mg.Hierarchy(A, 0.6, 1, 1).coarsen(10.0, 2, 8.0, 2500, 32).finalize()
    b = ctx.vec(n).rand(seed=0); x = ctx.vec(n); y = ctx.vec(n)
    for _ in range(3): h.vcycle(b, x)
    t = sorted(h.time_vcycle(b, x, reps=20) for _ in range(3))
    s = sorted(A.time_kernel(0, b, out=y, reps=20) for _ in range(3))
    print(f"{tag:48s} cycle {t[0]:.3f}/{t[1]:.3f} ms  spmv {s[0]:.3f}/{s[1]:.3f} ms  b@{b.ptr:#x} x@{x.ptr:#x}", flush=True)
    return A, h, b, x, y


keep = build("fresh")
del keep
keep = build("after free of the first hierarchy")
del keep
tmp = [ctx.vec(n) for _ in range(8)]
del tmp
keep = build("after 8 vectors allocated and freed")
del keep
fill = [ctx.vec(1000003 * (i + 1)) for i in range(6)]
keep = build("with 6 odd-sized fillers alive")
del keep
tmp = [ctx.vec(n) for _ in range(8)]
keep2 = build("with 8 big vectors alive")
