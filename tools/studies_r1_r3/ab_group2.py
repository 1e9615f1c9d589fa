#!/usr/bin/env python3
"""same-process A/B of grouping variants of the grouped pre pass (option sets given as key=value,key=value strings; groups are
built at a hierarchy's first cycle): alternating timed cycles, so box-to-box and run-to-run drift cancels.
usage: ab_group2.py <grid> <optset1> <optset2> [...]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import multigridsolver_amd as mg
N = int(sys.argv[1]); sets = sys.argv[2:]
REP = int(os.environ.get("AB_INSTANCES", "1"))      # hierarchies per variant (cycle time depends on where a hierarchy landed in HBM by up to ±3 %:
sets = sets * REP                                   # with AB_INSTANCES=2 the variants alternate A B A B and every instance is listed)
ctx = mg.Context(0)
A = ctx.poisson3d(N); n = N ** 3
b = ctx.vec(n).rand(seed=0)
DEFAULTS = "fuse_restrict=1,group_min_link=1,group_blocks=4,group_stray_pct=6,diag_from_values=1,group_concurrent=0,group_sweep=0,merge_ap=1"


def apply(sset):
    for kv in (DEFAULTS + "," + sset).split(","):      # every variant starts from the defaults (an option of the previous variant must not leak)
        k, v = kv.split("="); ctx.set_option(k, int(v))


hs = []
for sset in sets:
    apply(sset)
    h = mg.Hierarchy(A, 0.6, 1, 1).coarsen(10.0, 2, 8.0, 2500, 32).finalize()
    x = ctx.vec(n)
    for _ in range(3): h.vcycle(b, x)
    hs.append((sset, h, x, [h.group_info(l) for l in range(2)]))
res = {i: [] for i in range(len(hs))}
for rnd in range(5):
    for i, (sset, h, x, _) in enumerate(hs):
        apply(sset)
        h.vcycle(b, x)
        res[i].append(h.time_vcycle(b, x, reps=20))
for i, (sset, h, x, info) in enumerate(hs):
    print(f"{sset:40s} min {min(res[i]):.3f} med {sorted(res[i])[2]:.3f} ms   L0 {info[0]}")
