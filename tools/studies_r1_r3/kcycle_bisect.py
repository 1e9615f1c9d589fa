#!/usr/bin/env python3
"""Which option makes FGCR(10) + K-cycle on level 1 stall at 256^3 (314 iterations against 96 with the V-cycle)?  One hierarchy per
option setting, same right-hand side.  usage: kcycle_bisect.py [N=256] [klevels=1]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import multigridsolver_amd as mg  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
kl = int(sys.argv[2]) if len(sys.argv) > 2 else 1
for name, opts in [("default", {}), ("nt_store=0", {"nt_store": 0}), ("graph=0", {"graph": 0}), ("fuse=0", {"fuse": 0}), ("fuse_restrict=0", {"fuse_restrict": 0}),
                   ("merge_ap=0", {"merge_ap": 0}), ("rowcode=0", {"rowcode": 0}), ("fuse_operands=0", {"fuse_operands": 0}), ("stage_unroll=0", {"stage_unroll": 0}),
                   ("blas1_vec=0", {"blas1_vec": 0}), ("strip=0", {"strip": 0}), ("xcd_remap=0", {"xcd_remap": 0})]:
    ctx = mg.Context(0)
    for k, v in opts.items():
        ctx.set_option(k, v)
    A = ctx.poisson3d(N); n = N ** 3
    h = mg.Hierarchy(A, 0.6, 1, 1).coarsen(10.0, 2, 8.0, 2500, 32).finalize()
    b = ctx.vec(n).rand(seed=0)
    h.set_kcycle(kl)
    x = ctx.vec(n)
    t0 = time.perf_counter()
    st, it, tol = mg.fgcr(A, x, b, h, 10, 400, 1e-10)
    print(f"N={N} K x{kl} {name:18s}: status {st} iterations {it} tol {tol:.2e}  {time.perf_counter() - t0:.2f}s", flush=True)
    del h, A, b, x
    ctx.close()
