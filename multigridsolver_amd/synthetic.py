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


CSKY_ROWSUM_MARGIN = 2.86e-6      # the bundled CSky3d30's printed interior row sums, relative to the diagonal (median; 95 % of its interior rows)


def csky3d(N, velocity=1000.0, workers=None, digits=6, rowsum_floor=None):
    """The reference's OWN convection-diffusion family at any size: `matrices/CSky3d30.mtx` decoded — the N = 30 instance reproduces the bundled
    file entry for entry to its six printed digits (tests/test_synthetic.py).  -div(D grad u) + v . grad u on the unit cube, nodes at
    t_m = m*h, h = 1/N, rows e = (i*N + j)*N + k, the whole operator scaled by h:
      * D(x, y, z) = 1000 * (2 * floor(5 y) + 1) inside the "skyscraper" cubes {frac(5 t) < 1/2 in all three coordinates}, 1 elsewhere;
      * a face between two nodes takes the harmonic mean of D at two sample points (2*1*1000/1001 = 1.998 in the file): k-faces
        (x_i, y_j, k*h) and (x_i, y_j, (k+1)*h); j-faces (x_i, j*h, z_k) and (x_i, j*h + h, z_k); i-faces (i*h, j*h + h, z_k) and
        (i*h + h, j*h + h, z_k) — the i-faces see the field one node further in y, and `m*h + h` is not `(m+1)*h` in floating point
        (5*h + h < 0.2 <= 6*h at N = 30): both quirks of the generator that wrote the file are kept, they decide which rows pair up;
      * the LOW boundary face of an axis takes D at the node itself, the HIGH boundary face D at the virtual node behind the last (both
        with the unshifted y, also for the i axis);
      * v = (velocity, velocity, velocity), first-order upwind: the neighbour at index - stride gets -(D_face*h + v*h^2), the one at
        + stride -D_face*h; the diagonal always carries six diffusion and three convection terms (Dirichlet data on all six faces).
      * digits = 6 (default): every value is rounded to six significant decimal digits, as `writeMatrix` prints them (src/common/MatrixIO.cpp:39-57,
        default stream precision) — with it csky3d(30) IS the bundled matrix, bit for bit after parsing.  This is not cosmetic: interior rows
        have zero row sums, the reference's pair test `a_ii - s_i + a_jj - s_j >= 0` (src/CPU_C++/AGMG.cpp:159,233; src/GPU_CUDAC++/Aggregation.cu:157) then hangs on the sign of
        rounding noise, and its setup coarsens the exact operator 1.4x per level instead of the 3.7x it reaches on its own files, whose printed
        digits leave most row sums at +2.86e-6 * a_ii (oracle AGMG on N = 30: 27000 -> 18651 exact, -> 7283 printed).  digits = None: exact values;
      * rowsum_floor (default None: values exactly as printed): at other N the printed digits fall differently (N = 64: every background row sum
        is -1e-6 and the reference's rule pairs almost nothing: 262144 -> 207883).  rowsum_floor = f raises the diagonal of every interior row
        whose printed row sum is below f * a_ii by the difference (f = 2.86e-6: the bundled instance's own margin; a relative change of the
        diagonal of at most a few 1e-6) — a stand-in on which the reference's setup behaves as on the file it ships.
    Returns (rowptr i32, col i32, val f64), sorted columns."""
    n = N ** 3
    h = 1.0 / N
    m = np.arange(N + 1, dtype=np.float64)
    t_prod = m * h                        # m*h, m = 0 .. N
    t_sum = m[:N] * h + h                 # m*h + h, m = 0 .. N-1 (differs from (m+1)*h in the last bit at some m)

    def cube(t):
        return (5.0 * t - np.floor(5.0 * t)) < 0.5

    def val(t):
        return 1000.0 * (2.0 * np.floor(5.0 * t) + 1.0)

    def field(cx, cy, vy, cz):            # D on the outer product of three 1-D sample sets
        return np.where(cx[:, None, None] & cy[None, :, None] & cz[None, None, :], vy[None, :, None], 1.0)

    def hmean(a, b):
        return 2.0 * a * b / (a + b)

    Cp, Cs, Vp, Vs = cube(t_prod), cube(t_sum), val(t_prod), val(t_sum)
    chunk_planes = max(1, min(N, (1 << 21) // (N * N), (N + 15) // 16))
    counts = np.zeros(n, dtype=np.int64)

    def chunk(p0):
        p1 = min(N, p0 + chunk_planes)
        P = p1 - p0
        xs = slice(p0, p1)
        # k-faces: face[k] between k and k+1 (k = 0..N-2), low boundary D at node 0, high boundary D at node N
        Dk = field(Cp[xs], Cp[:N], Vp[:N], Cp[:N + 1])                       # (P, N, N+1): z nodes 0..N
        fk_hi = hmean(Dk[:, :, :N], Dk[:, :, 1:]); fk_hi[:, :, N - 1] = Dk[:, :, N]
        fk_lo = np.empty((P, N, N)); fk_lo[:, :, 1:] = fk_hi[:, :, :N - 1]; fk_lo[:, :, 0] = Dk[:, :, 0]
        # j-faces: samples y = j*h and j*h + h
        Dj0 = field(Cp[xs], Cp[:N], Vp[:N], Cp[:N]); Dj1 = field(Cp[xs], Cs, Vs, Cp[:N])
        fj_hi = hmean(Dj0, Dj1); fj_hi[:, N - 1, :] = Dj1[:, N - 1, :]
        fj_lo = np.empty((P, N, N)); fj_lo[:, 1:, :] = fj_hi[:, :N - 1, :]; fj_lo[:, 0, :] = Dj0[:, 0, :]
        # i-faces: y = j*h + h for both samples, x = i*h and i*h + h; the face below plane p0 belongs to plane p0 - 1
        Di0 = field(Cp[xs], Cs, Vs, Cp[:N]); Di1 = field(Cs[xs], Cs, Vs, Cp[:N])
        fi_hi = hmean(Di0, Di1)
        if p1 == N:
            fi_hi[-1] = field(Cs[N - 1:N], Cp[:N], Vp[:N], Cp[:N])[0]         # high boundary: D at the virtual node, unshifted y (as the low boundary)
        fi_lo = np.empty((P, N, N)); fi_lo[1:] = fi_hi[:-1]
        if p0 > 0:
            fi_lo[0] = hmean(field(Cp[p0 - 1:p0], Cs, Vs, Cp[:N]), field(Cs[p0 - 1:p0], Cs, Vs, Cp[:N]))[0]
        else:
            fi_lo[0] = Dj0[0]                                                # low boundary: D at the node itself (unshifted y)
        faces = {(0, -1): fi_lo, (0, 1): fi_hi, (1, -1): fj_lo, (1, 1): fj_hi, (2, -1): fk_lo, (2, 1): fk_hi}
        idx = (np.arange(p0, p1, dtype=np.int64)[:, None, None] * N + np.arange(N, dtype=np.int64)[None, :, None]) * N + np.arange(N, dtype=np.int64)[None, None, :]
        cand_col = np.empty((P, N, N, 7), dtype=np.int64); cand_val = np.zeros((P, N, N, 7)); present = np.zeros((P, N, N, 7), dtype=bool)
        diag = np.full((P, N, N), 3.0 * velocity * h * h)
        stride = (N * N, N, 1)
        slot = {(0, -1): 0, (1, -1): 1, (2, -1): 2, (2, 1): 4, (1, 1): 5, (0, 1): 6}
        ii = np.arange(p0, p1)[:, None, None] + np.zeros((1, N, N), dtype=np.int64)
        jj = np.arange(N)[None, :, None] + np.zeros((P, 1, N), dtype=np.int64)
        kk = np.arange(N)[None, None, :] + np.zeros((P, N, 1), dtype=np.int64)
        coord = (ii, jj, kk)
        for d in range(3):
            for sgn in (-1, 1):
                face = faces[(d, sgn)] * h
                inside = (coord[d] + sgn >= 0) & (coord[d] + sgn < N)
                diag += face
                w = face + (velocity * h * h if sgn < 0 else 0.0)
                s_ = slot[(d, sgn)]
                cand_col[..., s_] = idx + sgn * stride[d]; cand_val[..., s_] = -w; present[..., s_] = inside
        cand_col[..., 3] = idx; cand_val[..., 3] = diag; present[..., 3] = True
        counts[p0 * N * N:p1 * N * N] = present.reshape(-1, 7).sum(axis=1)
        if digits:                                   # a few hundred distinct values per chunk: print and parse each once
            u, inv = np.unique(cand_val[present], return_inverse=True)
            cand_val[present] = np.array([float(f"%.{int(digits)}g" % x) for x in u])[inv]
        if rowsum_floor:
            rs = np.where(present, cand_val, 0.0).sum(axis=-1)
            lift = np.where(present.all(axis=-1), np.maximum(rowsum_floor * cand_val[..., 3] - rs, 0.0), 0.0)
            cand_val[..., 3] += lift
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
