"""Host-side generators of the stand-in operators (multigridsolver_amd/synthetic.py): CSR contract, and that `csky3d` at N = 30 IS the
reference's bundled CSky3d30 (every printed value, bit for bit), so that other N are members of the reference's own family."""
import gzip
import io

import numpy as np


def test_csky3d_at_30_is_the_bundled_csky3d30_bit_for_bit():
    import os
    import scipy.sparse as sps
    from conftest import REPO
    from multigridsolver_amd.synthetic import csky3d
    with gzip.open(os.path.join(REPO, "tests", "golden", "inputs", "CSky3d30.mtx.gz"), "rt") as f:
        lines = [l for l in f if not l.startswith("%")]
    M, N, L = map(int, lines[0].split())
    d = np.loadtxt(io.StringIO("".join(lines[1:])))
    A = sps.csr_matrix((d[:, 2], (d[:, 0].astype(int) - 1, d[:, 1].astype(int) - 1)), shape=(M, N)).tocsr(); A.sort_indices()
    rp, ci, v = csky3d(30)
    assert np.array_equal(A.indptr, rp) and np.array_equal(A.indices, ci)
    assert np.array_equal(A.data, v)                              # the six printed digits of every one of the 183 600 entries
    rows = np.repeat(np.arange(M), np.diff(rp))
    assert np.all(v[rows != ci] < 0) and np.all(v[rows == ci] > 0)
    # unrounded values differ from the file in the seventh digit only
    _, _, vx = csky3d(30, digits=None)
    assert np.max(np.abs(vx - v) / np.abs(v)) < 5.1e-6
    # strong convection: upwind neighbour -(D h + v h^2), downwind -D h with v = 1000, h = 1/30, D = 1 in the background
    e = (15 * 30 + 15) * 30 + 15
    r = dict(zip((A.indices[A.indptr[e]:A.indptr[e + 1]] - e).tolist(), vx[rp[e]:rp[e + 1]].tolist()))
    assert abs(r[-1] + (1 / 30 + 1000 / 900)) < 1e-12 and abs(r[1] + 1 / 30) < 1e-12 and abs(r[0] - (6 / 30 + 3 * 1000 / 900)) < 1e-12


def test_csky3d_rowsum_floor_touches_interior_diagonals_only_and_restores_the_pairing():
    """the reference's pair rule needs a_ii - s_i + a_jj - s_j >= 0; on its own file the printed digits leave 95 % of the interior row sums at
    +2.86e-6 a_ii, at other N they do not (N = 48: 110592 -> 82980 rows).  `rowsum_floor` gives every N the file's margin."""
    from oracle import oracle_py as orc
    from multigridsolver_amd.synthetic import csky3d
    N = 24
    rp, ci, v0 = csky3d(N)
    rp1, ci1, v1 = csky3d(N, rowsum_floor=2.86e-6)
    assert np.array_equal(rp, rp1) and np.array_equal(ci, ci1)
    rows = np.repeat(np.arange(N ** 3), np.diff(rp))
    changed = v0 != v1
    assert np.all(rows[changed] == ci[changed]) and np.all(np.diff(rp)[rows[changed]] == 7)
    assert np.max((v1 - v0)[changed] / v0[changed]) < 2e-5
    rs = np.add.reduceat(v1, rp[:-1]); dg = v1[rows == ci]
    interior = np.diff(rp) == 7
    assert np.all(rs[interior] >= 2.85e-6 * dg[interior])
    sizes = []
    for vals in (v0, v1):
        P = orc.Csr.from_arrays(N ** 3, N ** 3, rp, ci, vals).agmg(10.0, 2, 8.0, strict=False)
        sizes.append(P.shape[1])
    assert sizes[1] * 3.5 < N ** 3, sizes                        # ~4x per level, as on the bundled file (27000 -> 7283)


def test_generators_give_sorted_csr_and_do_not_depend_on_chunking():
    from multigridsolver_amd.synthetic import convdiff3d, csky3d
    for gen in (convdiff3d, csky3d):
        a = gen(12, workers=1)
        b = gen(12, workers=4)
        for x, y in zip(a, b):
            assert np.array_equal(x, y)
        rp, ci, v = a
        assert rp[0] == 0 and rp[-1] == len(ci) == len(v) and rp.dtype == np.int32 and ci.dtype == np.int32
        for i in range(0, 12 ** 3, 97):
            c = ci[rp[i]:rp[i + 1]]
            assert np.all(np.diff(c) > 0) and i in c
