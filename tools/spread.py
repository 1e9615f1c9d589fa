#!/usr/bin/env python3
"""One fresh process = one sample of the headline numbers (fine-level SpMV by HIP events, zero-guess V(1,1) cycle) at N^3: run it several
times in a row to see the process-to-process spread (tools/run_spread.sh).  usage: spread.py [grid=512]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multigridsolver_amd as mg
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ctx = mg.Context(0)
A = ctx.poisson3d(N); n = N ** 3
h = mg.Hierarchy(A, 0.6, 1, 1).coarsen(10.0, 2, 8.0, 2500, 32).finalize()
b = ctx.vec(n).rand(seed=0); x = ctx.vec(n); xs = ctx.vec(n).rand(seed=1); y = ctx.vec(n)
A.optimize()
for _ in range(4):
    h.vcycle(b, x)
A.time_kernel(mg.OP_SPMV, xs, out=y, reps=5)
sp = sorted(A.time_kernel(mg.OP_SPMV, xs, out=y, reps=20) for _ in range(5))
cy = sorted(h.time_vcycle(b, x, reps=20) for _ in range(5))
print(f"SPREAD grid {N}: spmv {sp[2]:.4f} ms (min {sp[0]:.4f}), cycle {cy[2]:.4f} ms (min {cy[0]:.4f}), {1e3 / cy[2]:.1f} V-cycles/s  arena={os.environ.get('MGS_ARENA_GB', '-')}", flush=True)
