"""Host-side generators of the stand-in operators (multigridsolver_amd/synthetic.py): CSR contract, and how close `csky3d` is to the
reference's bundled CSky3d30 (it is modelled on it, not equal to it)."""
import gzip
import io

import numpy as np


def test_csky3d_is_modelled_on_the_bundled_csky3d30(inputs_gz=None):
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
    # the same sparsity pattern (7-point, sorted columns), every off-diagonal entry negative, the same background stencil
    assert np.array_equal(A.indptr, rp) and np.array_equal(A.indices, ci)
    rows = np.repeat(np.arange(M), np.diff(rp))
    assert np.all(v[rows != ci] < 0) and np.all(v[rows == ci] > 0)
    close = np.abs(A.data - v) <= 2e-5 * np.abs(A.data)          # the file prints six digits
    assert close.mean() >= 0.93, close.mean()                   # the rest: faces of the high-diffusion cubes this model places half a cell off
    # strong convection: upwind neighbour -(D h + v h^2), downwind -D h with v = 1000, h = 1/30, D = 1 in the background
    e = (15 * 30 + 15) * 30 + 15
    r = dict(zip((A.indices[A.indptr[e]:A.indptr[e + 1]] - e).tolist(), v[rp[e]:rp[e + 1]].tolist()))
    assert abs(r[-1] + (1 / 30 + 1000 / 900)) < 1e-12 and abs(r[1] + 1 / 30) < 1e-12 and abs(r[0] - (6 / 30 + 3 * 1000 / 900)) < 1e-12


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
