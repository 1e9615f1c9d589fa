"""Host-side generators of synthetic operators for tests and bench.py (numpy/scipy only; the Poisson operator of BASELINE.json
configs[4] is generated on the device by `mgs_csr_poisson3d`).

`convdiff3d(N)`: a labelled STAND-IN for the reference's `matvf3dSky*` / `CSky3d*` problem class (the paper's nonsymmetric
convection-diffusion with "skyscraper" coefficient jumps, docs/AGMG_For_Convection_Diffusion.pdf §5; matrices/CSky3d30.mtx is the 30^3
member bundled with the reference, the 80^3 one is absent — /root/reference/.MISSING_LARGE_BLOBS).  It is NOT one of those matrices:
7-point upwind finite differences on an N^3 grid, rotating velocity field, diffusion coefficient jumping by `jump` in columns, Dirichlet
boundary.  Rows in lexicographic order e = (i*N + j)*N + k, sorted columns — the CSR contract of readMatrix (src/common/MatrixIO.cpp:29)."""
import numpy as np


def convdiff3d(N, jump=1e3, vel_scale=200.0, chunk_planes=None):
    """Returns (rowptr i32[n+1], col i32[nnz], val f64[nnz]) of the N^3 stand-in operator.  Built plane-chunk by plane-chunk so that
    256^3 (1.2e8 entries) stays within a few GB of host memory."""
    n = N ** 3
    g = (np.arange(N) + 0.5) / N
    hgrid = 1.0 / N
    stride = (N * N, N, 1)
    if chunk_planes is None:
        chunk_planes = max(1, min(N, (1 << 22) // (N * N)))

    def kappa_of(ci, cj, ck):        # coordinates as integer arrays (may be one past the grid: clipped by the caller's mask)
        X, Y, Z = g[np.clip(ci, 0, N - 1)], g[np.clip(cj, 0, N - 1)], g[np.clip(ck, 0, N - 1)]
        return np.where(((np.floor(X * 8) + np.floor(Y * 8)) % 3 == 0) & (Z < 0.6), jump, 1.0)

    rowptr = np.zeros(n + 1, dtype=np.int64)
    cols_out, vals_out = [], []
    for p0 in range(0, N, chunk_planes):
        p1 = min(N, p0 + chunk_planes)
        idx = np.arange(p0 * N * N, p1 * N * N, dtype=np.int64)
        c = [idx // (N * N), (idx // N) % N, idx % N]
        X, Y, Z = g[c[0]], g[c[1]], g[c[2]]
        kap = kappa_of(*c)
        vel = [2 * Y * (1 - X ** 2) * vel_scale, -2 * X * (1 - Y ** 2) * vel_scale, np.sin(np.pi * Z) * (vel_scale / 4.0)]
        m = idx.size
        # 7 candidate entries per row in ascending column order: -N^2, -N, -1, diag, +1, +N, +N^2
        cand_col = np.empty((m, 7), dtype=np.int64); cand_val = np.zeros((m, 7)); present = np.zeros((m, 7), dtype=bool)
        diag = np.zeros(m)
        slot = {(0, -1): 0, (1, -1): 1, (2, -1): 2, (2, 1): 4, (1, 1): 5, (0, 1): 6}
        for d in range(3):
            for sgn in (-1, 1):
                cn = [c[0], c[1], c[2]]; cn[d] = c[d] + sgn
                inside = (cn[d] >= 0) & (cn[d] < N)
                kn = np.where(inside, kappa_of(*cn), kap)
                kf = 2.0 / (1.0 / kap + 1.0 / kn)                     # harmonic mean on the face
                diff = kf / hgrid ** 2
                conv = np.maximum(-sgn * vel[d], 0.0) / hgrid         # upwind: only the inflow neighbour
                w = diff + conv
                diag += np.where(inside, w, diff)                     # Dirichlet: the boundary face keeps its diffusion term
                s = slot[(d, sgn)]
                cand_col[:, s] = idx + sgn * stride[d]; cand_val[:, s] = -w; present[:, s] = inside
        cand_col[:, 3] = idx; cand_val[:, 3] = diag; present[:, 3] = True
        rowptr[idx + 1] = present.sum(axis=1)
        cols_out.append(cand_col[present].astype(np.int32)); vals_out.append(cand_val[present])
    np.cumsum(rowptr, out=rowptr)
    assert rowptr[-1] < 2 ** 31
    return rowptr.astype(np.int32), np.concatenate(cols_out), np.concatenate(vals_out)
