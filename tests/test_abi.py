"""CPU-only checks of the drop-in boundary: the C-ABI library loads without a GPU, exports
every symbol include/mgs.h declares, its host-side loader matches the oracle/reference, and
compute entry points fail loudly (no CPU fallback) when no device exists."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import REPO


def _declared_symbols():
    src = open(os.path.join(REPO, "include", "mgs.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mgs_[A-Za-z0-9_]+)\s*\(", src)) - {"mgs_halo_fn", "mgs_allreduce_fn"})


def test_library_exports_every_declared_symbol():
    import multigridsolver_amd as mg
    L = C.CDLL(mg.SO_PATH)
    names = _declared_symbols()
    assert len(names) >= 55
    for n in names:
        assert hasattr(L, n), f"libmgs.so lacks {n} declared in include/mgs.h"
    from multigridsolver_amd._lib import PROTOTYPES
    assert set(names) == set(PROTOTYPES), set(names) ^ set(PROTOTYPES)
    assert b"gfx950" in mg.lib().mgs_version()


def test_loader_matches_oracle(orc, inputs, tmp_path):
    import multigridsolver_amd as mg
    for name in ["SmallTestMatrix", "poisson10000promatrix", "CSky3d10", "CSky3d3"]:
        rows, cols, rp, ci, v = mg.read_mtx(inputs[name])
        o = orc.Csr.read(inputs[name])
        assert (rows, cols) == o.shape
        assert np.array_equal(rp, o.rowptr) and np.array_equal(ci, o.col) and np.array_equal(v, o.val)
    # writer: byte-identical to the oracle's restatement of writeMatrix (MatrixIO.cpp:39-57)
    rows, cols, rp, ci, v = mg.read_mtx(inputs["CSky3d3"])
    a, b = str(tmp_path / "a.mtx"), str(tmp_path / "b.mtx")
    mg.write_mtx(a, rows, cols, rp, ci, v)
    orc.Csr.read(inputs["CSky3d3"]).write(b)
    assert open(a, "rb").read() == open(b, "rb").read()
    assert open(a).readline() == "%%MatrixMarket matrix coordinate real general \n"


def test_loader_errors(tmp_path):
    import multigridsolver_amd as mg
    with pytest.raises(mg.MgsError):
        mg.read_mtx(str(tmp_path / "missing.mtx"))
    p = tmp_path / "short.mtx"
    p.write_text("%%MatrixMarket matrix coordinate real general\n3 3 4\n1 1 1.0\n2 2 1.0\n")
    with pytest.raises(mg.MgsError):
        mg.read_mtx(str(p))
    p = tmp_path / "oob.mtx"
    p.write_text("% c\n2 2 1\n3 1 1.0\n")
    with pytest.raises(mg.MgsError):
        mg.read_mtx(str(p))
    # padded header + arbitrary order + comment lines, like matrices/CSky3d3.mtx:2-5
    p = tmp_path / "ok.mtx"
    p.write_text("%%MatrixMarket matrix coordinate real general \n% comment\n   2  3  3      \n2 3 5\n1  2  -1.5\n2 1 4e0\n")
    rows, cols, rp, ci, v = mg.read_mtx(str(p))
    assert (rows, cols) == (2, 3) and rp.tolist() == [0, 1, 3] and ci.tolist() == [1, 0, 2] and v.tolist() == [-1.5, 4.0, 5.0]


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import multigridsolver_amd as mg
    with pytest.raises(mg.MgsError) as e:
        mg.Context(0)
    assert e.value.code == -2


def test_loader_property_random_files(orc, tmp_path):
    """property test of the .mtx contract (MatrixIO.cpp:12-37): any entry order, any whitespace, leading
    comment lines, scientific notation — library loader == oracle loader == scipy's own assembly"""
    import multigridsolver_amd as mg
    import scipy.sparse as sps
    from hypothesis import given, settings, strategies as st

    @settings(max_examples=40, deadline=None)
    @given(st.integers(1, 12), st.integers(1, 12), st.data())
    def run(M, N, data):
        cells = data.draw(st.lists(st.tuples(st.integers(0, M - 1), st.integers(0, N - 1)), unique=True, max_size=40))
        vals = data.draw(st.lists(st.floats(-1e6, 1e6, allow_nan=False, width=64), min_size=len(cells), max_size=len(cells)))
        ncom = data.draw(st.integers(0, 3))
        seps = data.draw(st.lists(st.sampled_from([" ", "  ", "\t", " \t "]), min_size=2, max_size=2))
        p = tmp_path / "r.mtx"
        with open(p, "w") as f:
            for c in range(ncom):
                f.write("%" * (1 + c % 2) + " comment line %d\n" % c)
            f.write(f"  {M}{seps[0]}{N}{seps[1]}{len(cells)}   \n")
            for (i, j), v in zip(cells, vals):
                f.write(f"{i + 1}{seps[0]}{j + 1}{seps[1]}{v!r}\n")
        rows, cols, rp, ci, v = mg.read_mtx(str(p))
        o = orc.Csr.read(str(p))
        assert (rows, cols) == (M, N) == o.shape
        assert np.array_equal(rp, o.rowptr) and np.array_equal(ci, o.col) and np.array_equal(v, o.val)
        ref = sps.csr_matrix((vals, ([c[0] for c in cells], [c[1] for c in cells])), shape=(M, N)) if cells else sps.csr_matrix((M, N))
        ref.sort_indices()
        # scipy drops nothing here (explicit zeros are kept by the constructor only if present in data)
        assert np.array_equal(rp, ref.indptr) and np.array_equal(ci, ref.indices) and np.array_equal(v, ref.data)

    run()
