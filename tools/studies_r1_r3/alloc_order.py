#!/usr/bin/env python3
"""does it matter WHEN a process allocates its vectors?  One fresh process per variant: the SpMV/cycle vectors allocated before the
operator and the hierarchy exist ("early"), or after the setup's scratch memory was allocated and freed ("late", what bench.py does).
usage: alloc_order.py [N=512]   (alloc_order.py N early|late runs one variant)"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
if len(sys.argv) < 3:
    for rnd in range(2):
        for v in ("early", "late"):
            subprocess.run([sys.executable, os.path.abspath(__file__), str(N), v], check=False)
    sys.exit(0)
import multigridsolver_amd as mg
variant = sys.argv[2]
ctx = mg.Context(0)
n = N ** 3
if variant == "early":
    b = ctx.vec(n).rand(seed=0); x = ctx.vec(n); y = ctx.vec(n)
A = ctx.poisson3d(N)
h = mg.Hierarchy(A, 0.6, 1, 1).coarsen(10.0, 2, 8.0, 2500, 32).finalize()
if variant == "late":
    b = ctx.vec(n).rand(seed=0); x = ctx.vec(n); y = ctx.vec(n)
for _ in range(3): h.vcycle(b, x)
A.time_kernel(0, b, out=y, reps=3)
res = []
for nt in (0, 1, 0, 1):
    ctx.set_option("nt_store", nt)
    h.vcycle(b, x)
    res.append(f"nt={nt}: cycle {min(h.time_vcycle(b, x, reps=20) for _ in range(3)):.3f} spmv {min(A.time_kernel(0, b, out=y, reps=20) for _ in range(3)):.3f}")
print(f"{variant:5s} b@{b.ptr:#x} y@{y.ptr:#x}  " + " | ".join(res), flush=True)
