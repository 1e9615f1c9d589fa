"""Host-side generators of synthetic operators for tests and bench.py (numpy/scipy only; the Poisson operator of BASELINE.json
configs[4] is generated on the device by `mgs_csr_poisson3d`).

`convdiff3d(N)`: a labelled STAND-IN for the reference's `matvf3dSky*` / `CSky3d*` problem class (the paper's nonsymmetric
convection-diffusion with "skyscraper" coefficient jumps, docs/AGMG_For_Convection_Diffusion.pdf §5; matrices/CSky3d30.mtx is the 30^3
member bundled with the reference, the 80^3 one is absent — /root/reference/.MISSING_LARGE_BLOBS).  It is NOT one of those matrices:
7-point upwind finite differences on an N^3 grid, rotating velocity field, diffusion coefficient jumping by `jump` in columns, Dirichlet
boundary.  Rows in lexicographic order e = (i*N + j)*N + k, sorted columns — the CSR contract of readMatrix (src/common/MatrixIO.cpp:29)."""
import numpy as np


def convdiff3d(N, jump=1e3, vel_scale=200.0, chunk_planes=None, workers=None):
    """Returns (rowptr i32[n+1], col i32[nnz], val f64[nnz]) of the N^3 stand-in operator.  Built plane-chunk by plane-chunk with
    broadcasting over (planes, N, N) so that 256^3 (1.2e8 entries) takes well under a minute and a few GB of host memory."""
    n = N ** 3
    g = (np.arange(N) + 0.5) / N
    hgrid = 1.0 / N
    if chunk_planes is None:
        chunk_planes = max(1, min(N, (1 << 21) // (N * N), (N + 15) // 16))
    sky2d = ((np.floor(g[:, None] * 8) + np.floor(g[None, :] * 8)) % 3 == 0)           # (i, j): skyscraper footprint
    low = g < 0.6                                                                       # k: skyscraper height

    def kappa(i0, i1, j0, j1, k0, k1):          # coefficient on the index box [i0,i1) x [j0,j1) x [k0,k1) (inside the grid)
        return np.where(sky2d[i0:i1, j0:j1, None] & low[None, None, k0:k1], jump, 1.0)

    counts = np.zeros(n, dtype=np.int64)
    X2 = g[:, None, None]; Y2 = g[None, :, None]; Z2 = g[None, None, :]

    def chunk(p0):
        p1 = min(N, p0 + chunk_planes)
        P = p1 - p0
        Xc = X2[p0:p1]
        kap = kappa(p0, p1, 0, N, 0, N)
        vel = [np.broadcast_to(2 * Y2 * (1 - Xc ** 2) * vel_scale, (P, N, N)), np.broadcast_to(-2 * Xc * (1 - Y2 ** 2) * vel_scale, (P, N, N)),
               np.broadcast_to(np.sin(np.pi * Z2) * (vel_scale / 4.0), (P, N, N))]
        idx = (np.arange(p0, p1, dtype=np.int64)[:, None, None] * N + np.arange(N, dtype=np.int64)[None, :, None]) * N + np.arange(N, dtype=np.int64)[None, None, :]
        cand_col = np.empty((P, N, N, 7), dtype=np.int64); cand_val = np.zeros((P, N, N, 7)); present = np.zeros((P, N, N, 7), dtype=bool)
        diag = np.zeros((P, N, N))
        stride = (N * N, N, 1)
        slot = {(0, -1): 0, (1, -1): 1, (2, -1): 2, (2, 1): 4, (1, 1): 5, (0, 1): 6}
        for d in range(3):
            for sgn in (-1, 1):
                # neighbour's coefficient: the chunk's own array shifted along d, the plane across the chunk boundary fetched for d = 0
                kn = kap.copy(); inside = np.ones((P, N, N), dtype=bool)
                if d == 0:
                    if sgn < 0:
                        kn[1:] = kap[:-1]
                        if p0 > 0:
                            kn[0] = kappa(p0 - 1, p0, 0, N, 0, N)[0]
                        else:
                            inside[0] = False
                    else:
                        kn[:-1] = kap[1:]
                        if p1 < N:
                            kn[-1] = kappa(p1, p1 + 1, 0, N, 0, N)[0]
                        else:
                            inside[-1] = False
                elif d == 1:
                    if sgn < 0:
                        kn[:, 1:] = kap[:, :-1]; inside[:, 0] = False
                    else:
                        kn[:, :-1] = kap[:, 1:]; inside[:, -1] = False
                else:
                    if sgn < 0:
                        kn[:, :, 1:] = kap[:, :, :-1]; inside[:, :, 0] = False
                    else:
                        kn[:, :, :-1] = kap[:, :, 1:]; inside[:, :, -1] = False
                kn = np.where(inside, kn, kap)
                kf = 2.0 / (1.0 / kap + 1.0 / kn)                     # harmonic mean on the face
                diff = kf / hgrid ** 2
                conv = np.maximum(-sgn * vel[d], 0.0) / hgrid         # upwind: only the inflow neighbour
                w = diff + conv
                diag += np.where(inside, w, diff)                     # Dirichlet: the boundary face keeps its diffusion term
                s = slot[(d, sgn)]
                cand_col[..., s] = idx + sgn * stride[d]; cand_val[..., s] = -w; present[..., s] = inside
        cand_col[..., 3] = idx; cand_val[..., 3] = diag; present[..., 3] = True
        counts[p0 * N * N:p1 * N * N] = present.reshape(-1, 7).sum(axis=1)
        return cand_col[present].astype(np.int32), cand_val[present]

    starts = list(range(0, N, chunk_planes))
    if workers is None:
        import os
        workers = max(1, min(8, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else 1))
    if workers > 1 and len(starts) > 1:          # numpy releases the GIL inside its array operations: chunks in a few threads
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=workers) as ex:
            parts = list(ex.map(chunk, starts))
    else:
        parts = [chunk(p0) for p0 in starts]
    cols_out = [p[0] for p in parts]; vals_out = [p[1] for p in parts]
    rowptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(counts, out=rowptr[1:])
    assert rowptr[-1] < 2 ** 31
    return rowptr.astype(np.int32), np.concatenate(cols_out), np.concatenate(vals_out)


def csky3d(N, velocity=1000.0, workers=None):
    """The reference's OWN convection-diffusion family at any size: `matrices/CSky3d30.mtx` decoded (tests/test_synthetic.py holds the
    N = 30 instance against the bundled file to its six printed digits).  -div(D grad u) + v . grad u on the unit cube, h = 1/N, rows
    e = (i*N + j)*N + k, the whole operator scaled by h:
      * D = 1, except in "skyscrapers": 5 x 5 columns with a square footprint in (j, k) (frac(5y) < 1/2 and frac(5z) < 1/2) rising through
        the lower half of the i axis (x < 1/2), D = 1000 * (2 * floor(5y) + 1) — 1e3 ... 9e3;
      * face coefficients are harmonic means of the two cells (2*1*1000/1001 = 1.998 in the file), a boundary face takes its cell's D;
      * v = (velocity, velocity, velocity), first-order upwind: the neighbour at index - stride gets -(D_face*h + v*h^2), the one at
        + stride -D_face*h; Dirichlet data on all six faces (the diagonal always carries six diffusion and three convection terms).
    Returns (rowptr i32, col i32, val f64), sorted columns."""
    n = N ** 3
    h = 1.0 / N
    g = (np.arange(N) + 0.5) / N
    foot = (np.modf(5.0 * g)[0] < 0.5)
    dval = 1000.0 * (2.0 * np.floor(5.0 * g) + 1.0)
    chunk_planes = max(1, min(N, (1 << 21) // (N * N), (N + 15) // 16))
    counts = np.zeros(n, dtype=np.int64)

    def dcell(i0, i1):       # D on planes [i0, i1)
        sky = foot[i0:i1, None, None] & foot[None, :, None] & foot[None, None, :]
        return np.where(sky, dval[None, :, None], 1.0) + np.zeros((i1 - i0, N, N))

    def chunk(p0):
        p1 = min(N, p0 + chunk_planes)
        P = p1 - p0
        D = dcell(p0, p1)
        idx = (np.arange(p0, p1, dtype=np.int64)[:, None, None] * N + np.arange(N, dtype=np.int64)[None, :, None]) * N + np.arange(N, dtype=np.int64)[None, None, :]
        cand_col = np.empty((P, N, N, 7), dtype=np.int64); cand_val = np.zeros((P, N, N, 7)); present = np.zeros((P, N, N, 7), dtype=bool)
        diag = np.full((P, N, N), 3.0 * velocity * h * h)
        stride = (N * N, N, 1)
        slot = {(0, -1): 0, (1, -1): 1, (2, -1): 2, (2, 1): 4, (1, 1): 5, (0, 1): 6}
        for d in range(3):
            for sgn in (-1, 1):
                Dn = D.copy(); inside = np.ones((P, N, N), dtype=bool)
                if d == 0:
                    if sgn < 0:
                        Dn[1:] = D[:-1]
                        if p0 > 0:
                            Dn[0] = dcell(p0 - 1, p0)[0]
                        else:
                            inside[0] = False
                    else:
                        Dn[:-1] = D[1:]
                        if p1 < N:
                            Dn[-1] = dcell(p1, p1 + 1)[0]
                        else:
                            inside[-1] = False
                elif d == 1:
                    if sgn < 0:
                        Dn[:, 1:] = D[:, :-1]; inside[:, 0] = False
                    else:
                        Dn[:, :-1] = D[:, 1:]; inside[:, -1] = False
                else:
                    if sgn < 0:
                        Dn[:, :, 1:] = D[:, :, :-1]; inside[:, :, 0] = False
                    else:
                        Dn[:, :, :-1] = D[:, :, 1:]; inside[:, :, -1] = False
                Dn = np.where(inside, Dn, D)
                face = 2.0 * D * Dn / (D + Dn) * h
                diag += face
                w = face + (velocity * h * h if sgn < 0 else 0.0)
                s = slot[(d, sgn)]
                cand_col[..., s] = idx + sgn * stride[d]; cand_val[..., s] = -w; present[..., s] = inside
        cand_col[..., 3] = idx; cand_val[..., 3] = diag; present[..., 3] = True
        counts[p0 * N * N:p1 * N * N] = present.reshape(-1, 7).sum(axis=1)
        return cand_col[present].astype(np.int32), cand_val[present]

    starts = list(range(0, N, chunk_planes))
    if workers is None:
        import os
        workers = max(1, min(8, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else 1))
    if workers > 1 and len(starts) > 1:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=workers) as ex:
            parts = list(ex.map(chunk, starts))
    else:
        parts = [chunk(p0) for p0 in starts]
    rowptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(counts, out=rowptr[1:])
    assert rowptr[-1] < 2 ** 31
    return rowptr.astype(np.int32), np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts])
