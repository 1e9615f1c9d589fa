#!/usr/bin/env python3
"""where do the aggregates of the N^3 Poisson level-0 matching deviate from the dominant shape?  usage: agg_shapes.py [N=256]"""
import os, sys, collections
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import multigridsolver_amd as mg
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ctx = mg.Context(0)
A = ctx.poisson3d(N); n = N ** 3
h = mg.Hierarchy(A, 0.6, 1, 1).coarsen(10.0, 2, 8.0, 2500, 2)
agg = h.level_P(0).agg(); nc = h.level_shape(1)[0]
e = np.arange(n, dtype=np.int64)
order = np.argsort(agg, kind="stable"); order = order[agg[order] >= 0]
a_sorted = agg[order]; starts = np.r_[0, np.nonzero(np.diff(a_sorted))[0] + 1]
i, j, k = e // (N * N), (e // N) % N, e % N
def ext(v): return np.maximum.reduceat(v[order], starts) - np.minimum.reduceat(v[order], starts)
ei, ej, ek = ext(i), ext(j), ext(k)
sizes = np.diff(np.r_[starts, len(order)])
shape = ei * 100 + ej * 10 + ek
print("N", N, "nc", nc, "G0 rows", int((agg < 0).sum()))
print("shape (di dj dk) histogram:", collections.Counter(shape.tolist()).most_common(10))
lead = e[order][starts]; li, lj, lk = lead // (N * N), (lead // N) % N, lead % N
odd = shape != 11
print("non-(0,1,1) share", odd.mean())
for nm, v in (("i", li), ("j", lj), ("k", lk)):
    hst = np.bincount(v[odd], minlength=N)
    print(f"  odd aggregates by leader {nm}: first 12 {hst[:12].tolist()} ... mid {hst[N//2-4:N//2+4].tolist()} ... last 6 {hst[-6:].tolist()}; parity even/odd {hst[0::2].sum()}/{hst[1::2].sum()}")
# do odd aggregates fill whole lines (i,j)?
line = li * N + lj
per_line_odd = np.bincount(line[odd], minlength=N * N); per_line_all = np.bincount(line, minlength=N * N)
frac = per_line_odd[per_line_all > 0] / per_line_all[per_line_all > 0]
print("lines with leaders:", int((per_line_all > 0).sum()), "fully odd lines:", int((frac == 1).sum()), "fully regular:", int((frac == 0).sum()), "mixed:", int(((frac > 0) & (frac < 1)).sum()))
