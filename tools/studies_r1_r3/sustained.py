#!/usr/bin/env python3
"""cycle and SpMV time under sustained load: the same timed region again and again for about a minute (does the part throttle, or
does a process slow down for another reason once it has run for a while?).  usage: sustained.py [N=512] [seconds=60]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import multigridsolver_amd as mg
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
T = float(sys.argv[2]) if len(sys.argv) > 2 else 60
ctx = mg.Context(0)
n = N ** 3
A = ctx.poisson3d(N)
h = mg.Hierarchy(A, 0.6, 1, 1).coarsen(10.0, 2, 8.0, 2500, 32).finalize()
b = ctx.vec(n).rand(seed=0); x = ctx.vec(n); y = ctx.vec(n)
for _ in range(3): h.vcycle(b, x)
t0 = time.time(); k = 0
while time.time() - t0 < T:
    c = h.time_vcycle(b, x, reps=50); s = A.time_kernel(0, b, out=y, reps=50)
    if k % 4 == 0: print(f"t={time.time() - t0:6.1f} s  cycle {c:.3f} ms  spmv {s:.3f} ms", flush=True)
    k += 1
# a solve allocates and frees its eight work vectors: does the cycle time change afterwards?
for rnd in range(3):
    xs = ctx.vec(n); st, it, tol = mg.bicgstab(A, xs, b, h, 10, 1e-30)
    print(f"after solve {rnd}: cycle {h.time_vcycle(b, x, reps=50):.3f} ms  spmv {A.time_kernel(0, b, out=y, reps=50):.3f} ms", flush=True)
