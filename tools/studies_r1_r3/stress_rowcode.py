#!/usr/bin/env python3
"""Randomised stress of the pattern-coded index (one-off, GPU): grid operators with random defects (deleted entries, extra
long-range entries, empty rows, irregular tails) so that row blocks land on every side of the coding thresholds; the coded
kernel must give the same BITS as the plain CSR kernel for SpMV / residual / Jacobi and through whole cycles (coded
aggregate-mapped index, G0 rows).  usage: stress_rowcode.py [trials=40] [seed=0]"""
import os, sys
import numpy as np
import scipy.sparse as sps
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import multigridsolver_amd as mg

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
ctx = mg.Context(0)
stats = {"coded_all": 0, "coded_some": 0, "coded_none": 0}
for t in range(trials):
    dim = int(rng.choice([2, 3]))
    N = int(rng.integers(12, 70)) if dim == 2 else int(rng.integers(6, 26))
    n = N ** dim
    idx = np.arange(n)
    offs = [1, N] + ([N * N] if dim == 3 else [])
    rows, cols, vals = [idx], [idx], [np.full(n, 2.0 * dim + 1.0 + rng.random(n) * (t % 2))]
    coord = [(idx // (N ** d)) % N for d in range(dim)]
    for d, o in enumerate(offs):
        for sgn in (-1, 1):
            ok = (coord[d] + sgn >= 0) & (coord[d] + sgn < N)
            r = idx[ok]; c = r + sgn * o
            keep = rng.random(r.size) >= float(rng.choice([0.0, 0.0, 0.02, 0.3]))      # random defects
            w = -1.0 - (rng.random(r.size) * 0.5 if t % 3 == 0 else 0.0)
            rows.append(r[keep]); cols.append(c[keep]); vals.append(np.broadcast_to(w, r.shape)[keep] if np.ndim(w) else np.full(keep.sum(), w))
    if t % 4 == 1:                                         # a sprinkle of long-range entries
        k = max(1, n // 50); r = rng.integers(0, n, k); c = rng.integers(0, n, k)
        rows.append(r); cols.append(c); vals.append(-0.01 * np.ones(k))
    M = sps.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n, n))
    M.sum_duplicates(); M.sort_indices()
    if t % 5 == 2:                                         # irregular tail block appended
        R = sps.random(300, n + 300, density=8.0 / (n + 300), random_state=rng, format="csr"); R.data[:] = -0.1
        M = sps.bmat([[M, None], [R[:, :n], R[:, n:] + sps.identity(300) * 9.0]], format="csr"); M.sort_indices()
        n = M.shape[0]
    M = M.tocsr(); M.sort_indices()
    A = ctx.csr(n, n, M.indptr.astype(np.int32), M.indices.astype(np.int32), M.data)
    x_np = rng.standard_normal(n); b_np = rng.standard_normal(n); x = ctx.vec(x_np); b = ctx.vec(b_np)
    ctx.set_option("rowcode", 0)
    y0 = A.spmv(x).numpy(); r0 = A.residual(x, b).numpy(); d = A.diag_inv(); j0 = A.jacobi(d, 0.7, b, x).numpy()
    ctx.set_option("rowcode", 1)
    A.optimize(); info = A.rowcode_info()
    y1 = A.spmv(x).numpy(); r1 = A.residual(x, b).numpy(); j1 = A.jacobi(d, 0.7, b, x).numpy()
    assert np.array_equal(y0, y1) and np.array_equal(r0, r1) and np.array_equal(j0, j1), (t, dim, N, info)
    assert np.allclose(y1, M @ x_np, rtol=1e-12, atol=1e-12)
    stats["coded_all" if info["coded_blocks"] == info["blocks"] else "coded_some" if info["coded_blocks"] else "coded_none"] += 1
    # whole cycle, device-built hierarchy (G0 rows appear with the long-range / defect variants)
    h = mg.Hierarchy(A, 0.6, 1, 1).coarsen(10.0, 2, 8.0, coarse_rows=max(50, n // 60), max_levels=8)
    if h.nlev >= 2:
        h.finalize()
        c1 = h.vcycle(b).numpy()
        ctx.set_option("rowcode", 0); c0 = h.vcycle(b).numpy(); ctx.set_option("rowcode", 1)
        ctx.set_option("fuse_operands", 0); cg = h.vcycle(b).numpy(); ctx.set_option("fuse_operands", 1)
        ctx.set_option("merge_ap", 0); cm = h.vcycle(b).numpy()                    # post pass on A with mapped columns instead of the merged A·P
        ctx.set_option("rowcode", 0); cm0 = h.vcycle(b).numpy(); ctx.set_option("rowcode", 1); ctx.set_option("merge_ap", 1)
        assert np.array_equal(cm, cm0), (t, "cycle bits, unmerged operand", dim, N)
        assert np.linalg.norm(cm - c1) <= 1e-12 * np.linalg.norm(c1), (t, "merged vs unmerged A·P", dim, N)
        assert np.array_equal(c0, c1), (t, "cycle bits", dim, N)
        assert np.linalg.norm(cg - c1) <= 1e-12 * np.linalg.norm(c1), (t, "operand form", dim, N)
        # grouped pre pass forced onto these small, defective operators (any stray share, any level size): a second hierarchy (groups
        # are built once) against the separate kernels; coded vs plain index and ω/a_ii vs wd must not change a bit
        ctx.set_option("group_min_blocks", 1); ctx.set_option("group_stray_pct", 100); ctx.set_option("group_blocks", int(rng.choice([2, 3, 4])))
        ctx.set_option("group_concurrent", int(t % 2))
        hg = mg.Hierarchy(A, 0.6, 1, 1).coarsen(10.0, 2, 8.0, coarse_rows=max(50, n // 60), max_levels=8).finalize()
        g1 = hg.vcycle(b).numpy()
        stats["grouped_levels"] = stats.get("grouped_levels", 0) + sum(1 for l in range(hg.nlev - 1) if hg.group_info(l)["groups"] > 0)
        stats["stray_aggs"] = stats.get("stray_aggs", 0) + sum(hg.group_info(l)["stray_aggregates"] for l in range(hg.nlev - 1))
        ctx.set_option("rowcode", 0); g0 = hg.vcycle(b).numpy(); ctx.set_option("rowcode", 1)
        ctx.set_option("diag_from_values", 0); gw = hg.vcycle(b).numpy(); ctx.set_option("diag_from_values", 1)
        ctx.set_option("fuse_restrict", 0); gs = hg.vcycle(b).numpy(); ctx.set_option("fuse_restrict", 1)
        for k, v in (("group_min_blocks", 1024), ("group_stray_pct", 6), ("group_blocks", 4), ("group_concurrent", 0)): ctx.set_option(k, v)
        assert np.array_equal(g0, g1) and np.array_equal(gw, g1), (t, "grouped cycle bits", dim, N)
        assert np.array_equal(gs, c1), (t, "same hierarchy, separate kernels", dim, N)
        assert np.linalg.norm(g1 - c1) <= 1e-12 * np.linalg.norm(c1), (t, "grouped vs separate", dim, N, np.linalg.norm(g1 - c1) / np.linalg.norm(c1))
        del hg
    # opt-in value patterns: a second copy of the operator built with valcode = 1 must reproduce the same bits
    ctx.set_option("valcode", 1)
    try:
        A2 = ctx.csr(n, n, M.indptr.astype(np.int32), M.indices.astype(np.int32), M.data); A2.optimize()
        assert np.array_equal(A2.spmv(x).numpy(), y1) and np.array_equal(A2.residual(x, b).numpy(), r1) and np.array_equal(A2.jacobi(d, 0.7, b, x).numpy(), j1), (t, "valcode kernels")
        stats["valcoded"] = stats.get("valcoded", 0) + (1 if A2.rowcode_info()["coded_blocks"] else 0)
        h2 = mg.Hierarchy(A2, 0.6, 1, 1).coarsen(10.0, 2, 8.0, coarse_rows=max(50, n // 60), max_levels=8)
        if h2.nlev >= 2:
            h2.finalize()
            assert np.array_equal(h2.vcycle(b).numpy(), c1), (t, "valcode cycle bits", dim, N)
            h2.set_smoother(0.9, 1, 1); h.set_smoother(0.9, 1, 1)
            assert np.array_equal(h2.vcycle(b).numpy(), h.vcycle(b).numpy()), (t, "valcode cycle bits after a new omega", dim, N)
        del h2, A2
    finally:
        ctx.set_option("valcode", 0)
    del h, A, x, b, d
print("STRESS_OK", trials, stats)
