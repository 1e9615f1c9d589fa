#!/usr/bin/env python3
"""One process, one hierarchy: pattern-coded SpMV and V-cycle times at N^3 (min of 3 x 20 launches) and the cycle's result norm.
usage: cycle_time.py [N=512]   (MGS_LIBMGS=<path> picks another build of libmgs.so: tools/ab_lib.sh)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multigridsolver_amd as mg
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ctx = mg.Context(0); n = N ** 3
A = ctx.poisson3d(N)
h = mg.Hierarchy(A, 0.6, 1, 1).coarsen(10.0, 2, 8.0, 2500, 32).finalize()
b = ctx.vec(n).rand(seed=0); x = ctx.vec(n); y = ctx.vec(n); xs = ctx.vec(n).rand(seed=1)
A.optimize()
for _ in range(3): h.vcycle(b, x)
A.time_kernel(mg.OP_SPMV, xs, out=y, reps=3)
sp = min(A.time_kernel(mg.OP_SPMV, xs, out=y, reps=20) for _ in range(3))
cy = min(h.time_vcycle(b, x, reps=20) for _ in range(3))
print(f"{os.path.basename(os.environ.get('MGS_LIBMGS', 'libmgs.so'))}: coded SpMV {sp:.3f} ms  cycle {cy:.3f} ms  |x| {x.nrm2()!r}", flush=True)
