#!/usr/bin/env python3
"""host cost of the first cycle on a NEW (rhs, out) vector pair (stream capture + hipGraphInstantiate) against a replay and against
eager launches (option graph=0).  usage: capture_cost.py [N=512]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import multigridsolver_amd as mg
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ctx = mg.Context(0)
n = N ** 3
A = ctx.poisson3d(N)
h = mg.Hierarchy(A, 0.6, 1, 1).coarsen(10.0, 2, 8.0, 2500, 32).finalize()
b = ctx.vec(n).rand(seed=0); x = ctx.vec(n)
h.vcycle(b, x); ctx.sync()                    # operands, codes, first graph


def wall(f):
    ctx.sync(); t0 = time.perf_counter(); f(); ctx.sync(); return (time.perf_counter() - t0) * 1e3


print(f"levels {h.nlev if hasattr(h, 'nlev') else '?'}")
print(f"replay of a cached pair: {wall(lambda: h.vcycle(b, x)):.3f} ms")
for k in range(3):
    b2 = ctx.vec(n).rand(seed=k + 1); x2 = ctx.vec(n)
    print(f"new pair {k}: first cycle {wall(lambda: h.vcycle(b2, x2)):.3f} ms, second {wall(lambda: h.vcycle(b2, x2)):.3f} ms")
ctx.set_option("graph", 0)
print(f"eager: {wall(lambda: h.vcycle(b, x)):.3f} ms, again {wall(lambda: h.vcycle(b, x)):.3f} ms")
