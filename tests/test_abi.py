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


@pytest.mark.parametrize("threads", ["1", "2", "5", "16"])
def test_loader_parallel_pieces_token_contract(orc, tmp_path, monkeypatch, threads):
    """the loader cuts the file into one piece per thread at whitespace; the contract stays token based (MatrixIO.cpp:23-27 reads
    with `>>`): triples may span lines, any entry order, trailing tokens after the L-th triple are never read.  Library loader ==
    oracle loader for every piece count, values bit for bit (17-digit decimals go through strtod, short ones through the exact
    fast path)."""
    import multigridsolver_amd as mg
    monkeypatch.setenv("MGS_IO_THREADS", threads)
    rng = np.random.default_rng(int(threads))
    M, N = 37, 41
    cells = rng.permutation(M * N)[:600]
    vals = rng.standard_normal(600) * 10.0 ** rng.integers(-30, 30, 600)
    vals[::7] = np.round(vals[::7], 3)
    vals[5] = 0.0; vals[6] = -0.0; vals[8] = 5e-324; vals[9] = 1.7976931348623157e308; vals[10] = 123456789012345678.0
    seps = [" ", "\n", "\t", "  \n ", "\r\n", " \t"]
    p = tmp_path / "t.mtx"
    with open(p, "w") as f:
        f.write("%%MatrixMarket matrix coordinate real general \n% x\n")
        f.write(f"{M} {N}\n{len(cells)}")
        for q, (c, v) in enumerate(zip(cells, vals)):
            txt = repr(float(v)) if q % 3 else ("%.6g" % v)
            f.write(f"{seps[q % 6]}{c // N + 1}{seps[(q + 1) % 6]}{c % N + 1}{seps[(q + 2) % 6]}{txt}")
        f.write("\n9999 9999 not-a-number trailing tokens are never read\n")
    rows, cols, rp, ci, v = mg.read_mtx(str(p))
    o = orc.Csr.read(str(p))
    assert (rows, cols) == (M, N) == o.shape
    assert np.array_equal(rp, o.rowptr) and np.array_equal(ci, o.col)
    assert np.array_equal(v.view(np.int64), o.val.view(np.int64))          # bits, so that -0.0 and the denormal count
    # row-major files (what every writer emits) take the no-scatter path: same arrays
    a = tmp_path / "a.mtx"
    mg.write_mtx(str(a), rows, cols, rp, ci, v)
    r2 = mg.read_mtx(str(a))
    o2 = orc.Csr.read(str(a))
    assert np.array_equal(r2[2], o2.rowptr) and np.array_equal(r2[3], o2.col) and np.array_equal(r2[4].view(np.int64), o2.val.view(np.int64))
    b = tmp_path / "b.mtx"
    o.write(str(b))
    assert open(a, "rb").read() == open(b, "rb").read()


@pytest.mark.parametrize("threads", ["1", "3"])
def test_loader_rejects_what_the_stream_reader_rejects(tmp_path, monkeypatch, threads):
    """`istream >> int/double` fails on these tokens (libstdc++ num_get: no inf/nan/hex, overflow sets failbit); the reference then
    runs on garbage, this loader returns MGS_ERR_IO naming the first offending entry"""
    import multigridsolver_amd as mg
    monkeypatch.setenv("MGS_IO_THREADS", threads)
    body = "".join(f"{i + 1} {i + 1} {i}.5\n" for i in range(50))
    for bad, where in [("3 3 nan\n", "entry 51"), ("3 3 0x10\n", "entry 51"), ("3 3 1e999\n", "entry 51"), ("3 3 1e\n", "entry 51"),
                       ("3.0 3 1\n", "entry 51"), ("0 3 1\n", "entry 51"), ("3 51 1\n", "entry 51"), ("3 3 1,5\n", "entry 51")]:
        p = tmp_path / "bad.mtx"
        p.write_text("%h\n50 50 52\n" + body + bad + "4 5 1.0\n")
        with pytest.raises(mg.MgsError) as e:
            mg.read_mtx(str(p))
        assert where in str(e.value), str(e.value)
    p = tmp_path / "dup.mtx"
    p.write_text("%h\n50 50 52\n" + body + "7 9 1.0\n7 9 2.0\n")
    with pytest.raises(mg.MgsError) as e:
        mg.read_mtx(str(p))
    assert "duplicate entry (7,9)" in str(e.value)
    for hdr in ["50 50\n", "50 x 3\n", "-1 5 0\n", "\n% late comment\n5 5 0\n", ""]:
        p = tmp_path / "hdr.mtx"
        p.write_text("%h\n" + hdr)
        with pytest.raises(mg.MgsError) as e:
            mg.read_mtx(str(p))
        assert "bad size line" in str(e.value)
    p = tmp_path / "empty.mtx"
    p.write_text("%h\n4 6 0\n")
    rows, cols, rp, ci, v = mg.read_mtx(str(p))
    assert (rows, cols) == (4, 6) and rp.tolist() == [0] * 5 and len(ci) == 0 and len(v) == 0


def test_loader_default_threads_on_a_multi_megabyte_file(tmp_path):
    """a file large enough for the loader to cut it into several pieces by itself (no MGS_IO_THREADS): shuffled entries of a 2-D
    five-point operator with random values, written with repr() (17 digits) — arrays equal to scipy's assembly; the writer's output
    read back gives the %g-rounded values and the same pattern"""
    import multigridsolver_amd as mg
    import scipy.sparse as sps
    n = 230
    I = sps.identity(n); T = sps.diags([-1, -1], [-1, 1], shape=(n, n))
    A = (sps.kron(I, sps.diags([4], [0], shape=(n, n)) + T) + sps.kron(T, I)).tocoo()
    rng = np.random.default_rng(5)
    vals = rng.standard_normal(A.nnz) * 10.0 ** rng.integers(-8, 8, A.nnz)
    perm = rng.permutation(A.nnz)
    p = tmp_path / "big.mtx"
    with open(p, "w") as f:
        f.write("%%MatrixMarket matrix coordinate real general\n% shuffled\n")
        f.write(f"{n * n} {n * n} {A.nnz}\n")
        f.write("".join(f"{A.row[k] + 1} {A.col[k] + 1} {float(vals[k])!r}\n" for k in perm))
    assert p.stat().st_size > 4 << 20
    rows, cols, rp, ci, v = mg.read_mtx(str(p))
    ref = sps.csr_matrix((vals, (A.row, A.col)), shape=(n * n, n * n)); ref.sort_indices()
    assert (rows, cols) == ref.shape and np.array_equal(rp, ref.indptr) and np.array_equal(ci, ref.indices) and np.array_equal(v, ref.data)
    q = tmp_path / "out.mtx"
    mg.write_mtx(str(q), rows, cols, rp, ci, v)
    r2 = mg.read_mtx(str(q))
    assert np.array_equal(r2[2], rp) and np.array_equal(r2[3], ci)
    assert np.array_equal(r2[4], np.array([float("%g" % x) for x in v]))


def test_device_allocations_go_through_the_arena_entry_points():
    """every device allocation / release of the library is mgs_hip_malloc / mgs_hip_free (so an arena — mgs_arena_reserve, MGS_ARENA_GB —
    really holds everything); raw hipMalloc / hipFree appear only inside those two functions"""
    import glob
    import re
    pat = re.compile(r"\bhip(Malloc|Free)\(")
    raw = {}
    for f in glob.glob(os.path.join(REPO, "multigridsolver_amd", "csrc", "*.h*")):
        n = sum(1 for line in open(f) if pat.search(line) and not line.lstrip().startswith("//") and "mgs_fail(" not in line)
        if n:
            raw[os.path.basename(f)] = n
    # comm_p2p.hip: the peer-to-peer window is an allocation of its own by necessity (hipIpcGetMemHandle names whole allocations, and the window is
    # uncached memory: hipExtMallocWithFlags) — its one hipFree is the only raw call outside the arena's entry points
    assert raw == {"mgs_api.hip": 3, "comm_p2p.hip": 1}, raw
