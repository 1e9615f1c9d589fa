import gzip
import os
import shutil
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLD = os.path.join(REPO, "tests", "golden")
INP = os.path.join(GOLD, "inputs")
# the reference's bundled operators besides CSky3d30 (= config C3): golden vectors in tests/golden/bundled_<name>.npz,
# P = the reference CPU AGMG "10 2 8" (results.txt:22-24), which the oracle's restatement reproduces exactly
BUNDLED = ["CSky2d3", "CSky2d10", "CSky2d20", "CSky2d100", "CSky3d3", "CSky3d10", "CSky3d20"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """A fresh checkout has no built artefacts (they are git-ignored): build them once (hipcc cross-compiles
    gfx950 without a GPU).  The product itself never auto-builds or falls back — see multigridsolver_amd/_lib.py."""
    so = os.path.join(REPO, "multigridsolver_amd", "libmgs.so")
    exe = os.path.join(REPO, "multigridsolver_amd", "cpp", "mgs_bicg")
    if not (os.path.exists(so) and os.path.exists(exe)):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (oracle/mgs_oracle.c) — checker only."""
    from oracle import oracle_py
    oracle_py.lib()
    return oracle_py


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)
    return load


@pytest.fixture(scope="session")
def inputs(tmp_path_factory, orc):
    """Paths of the input matrices: bundled reference data + regenerated poisson10000."""
    d = tmp_path_factory.mktemp("mtx")
    out = {}
    for fn in os.listdir(INP):
        if fn.endswith(".mtx"):
            out[fn[:-4]] = os.path.join(INP, fn)
    for fn in os.listdir(INP):                     # the larger reference data files are committed gzipped
        if fn.endswith(".mtx.gz"):
            p = os.path.join(str(d), fn[:-3])
            with gzip.open(os.path.join(INP, fn), "rb") as f, open(p, "wb") as g:
                shutil.copyfileobj(f, g)
            out[fn[:-7]] = p
    # poisson10000.mtx is not bundled by the reference: regenerate it with the
    # oracle's restatement of src/common/poisson.cpp and write it in that program's format.
    p = os.path.join(str(d), "poisson10000.mtx")
    A = orc.poisson2d(100)
    with open(p, "w") as f:
        f.write("%MatrixMarket matrix coordinate real general\n")
        f.write("%d %d %d\n" % (A.shape[0], A.shape[1], A.nnz))
        rp, col, val = A.rowptr, A.col, A.val
        for i in range(A.shape[0]):
            for k in range(rp[i], rp[i + 1]):
                f.write("%d %d %d\n" % (i + 1, col[k] + 1, int(val[k])))
    out["poisson10000"] = p
    return out
