import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import multigridsolver_amd as mg
N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ctx = mg.Context(0)
A = ctx.poisson3d(N); n = N ** 3
h = mg.Hierarchy(A, 0.6, 1, 1).coarsen(10.0, 2, 8.0, 1024, 3)
agg = h.level_P(0).agg()
nc = h.level_shape(1)[0]
print("levels", [h.level_shape(l) for l in range(h.nlev)], "plan", h.level_A(1).plan_info())
e = np.arange(n); i, j, k = e // (N * N), (e // N) % N, e % N
sizes = np.bincount(agg[agg >= 0], minlength=nc)
print("G0 rows", (agg < 0).sum(), "agg size hist", np.bincount(sizes))
# shape of aggregates: extent in i,j,k
order = np.argsort(agg, kind="stable"); order = order[agg[order] >= 0]
a_sorted = agg[order]
starts = np.r_[0, np.nonzero(np.diff(a_sorted))[0] + 1]
def ext(v):
    mx = np.maximum.reduceat(v[order], starts); mn = np.minimum.reduceat(v[order], starts); return mx - mn
ei, ej, ek = ext(i), ext(j), ext(k)
import collections
print("extent (di,dj,dk) histogram:", collections.Counter(zip(ei.tolist(), ej.tolist(), ek.tolist())).most_common(8))
# numbering: aggregate id vs leader coordinates
lead = np.minimum.reduceat(e[order], starts)
li, lj, lk = lead // (N * N), (lead // N) % N, lead % N
ids = a_sorted[starts]
print("first ids/leaders:", list(zip(ids[:6].tolist(), li[:6].tolist(), lj[:6].tolist(), lk[:6].tolist())))
mid = len(ids) // 2
print("mid ids/leaders:", list(zip(ids[mid:mid+6].tolist(), li[mid:mid+6].tolist(), lj[mid:mid+6].tolist(), lk[mid:mid+6].tolist())))
rp, ci, v = h.level_A(1).download()
rows = np.repeat(np.arange(nc), np.diff(rp))
off = ci.astype(np.int64) - rows
print("level-1 column offset histogram (top 12):", collections.Counter(off.tolist()).most_common(12))
print("max |offset|", np.abs(off).max())
