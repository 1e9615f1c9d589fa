"""ctypes binding of oracle/libmgs_oracle.so — CPU ORACLE, TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module (see oracle/mgs_oracle.h).  The product package never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class OrcCsr(C.Structure):
    _fields_ = [("rows", C.c_int), ("cols", C.c_int), ("nnz", C.c_int),
                ("rowptr", C.POINTER(C.c_int)), ("col", C.POINTER(C.c_int)), ("val", C.POINTER(C.c_double))]


PRECOND_FN = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double))


def build():
    subprocess.run(["make", "-C", HERE, "oracle"], check=True, capture_output=True)


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(HERE, "libmgs_oracle.so")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(HERE, "mgs_oracle.c")):
            build()
        L = C.CDLL(so)
        dp, ip, cp = C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(OrcCsr)
        L.orc_mtx_read.argtypes = [C.c_char_p, cp]; L.orc_mtx_read.restype = C.c_int
        L.orc_mtx_write.argtypes = [C.c_char_p, cp]; L.orc_mtx_write.restype = C.c_int
        L.orc_csr_free.argtypes = [cp]; L.orc_csr_free.restype = None
        L.orc_csr_from_arrays.argtypes = [C.c_int, C.c_int, C.c_int, ip, ip, dp, cp]
        L.orc_spmv.argtypes = [cp, dp, dp]; L.orc_spmv.restype = None
        L.orc_transpose.argtypes = [cp, cp]
        L.orc_spgemm.argtypes = [cp, cp, cp]
        L.orc_galerkin.argtypes = [cp, cp, cp]
        L.orc_diag_inv.argtypes = [cp, dp]; L.orc_diag_inv.restype = None
        L.orc_residual.argtypes = [cp, dp, dp, dp]; L.orc_residual.restype = None
        L.orc_jacobi.argtypes = [cp, dp, C.c_double, dp, dp, dp]; L.orc_jacobi.restype = None
        L.orc_dot.argtypes = [C.c_int, dp, dp]; L.orc_dot.restype = C.c_double
        L.orc_nrm2.argtypes = [C.c_int, dp]; L.orc_nrm2.restype = C.c_double
        L.orc_twogrid_jacobi.argtypes = [cp, cp, C.c_double, dp, dp]
        L.orc_hier_create.argtypes = [C.c_int, C.POINTER(cp), C.POINTER(cp), C.c_double, C.c_int, C.c_int]
        L.orc_hier_create.restype = C.c_void_p
        L.orc_hier_create_from_P.argtypes = [cp, C.c_int, C.POINTER(cp), C.c_double, C.c_int, C.c_int]
        L.orc_hier_create_from_P.restype = C.c_void_p
        L.orc_hier_destroy.argtypes = [C.c_void_p]; L.orc_hier_destroy.restype = None
        L.orc_hier_nlev.argtypes = [C.c_void_p]
        L.orc_hier_set_smoother.argtypes = [C.c_void_p, C.c_double, C.c_int, C.c_int]; L.orc_hier_set_smoother.restype = None
        L.orc_hier_set_kcycle.argtypes = [C.c_void_p, C.c_int]; L.orc_hier_set_kcycle.restype = None
        L.orc_hier_set_kcycle_energy.argtypes = [C.c_void_p, C.c_int]; L.orc_hier_set_kcycle_energy.restype = None
        L.orc_hier_set_additive.argtypes = [C.c_void_p, C.c_int]; L.orc_hier_set_additive.restype = None
        L.orc_hier_set_correction_scale.argtypes = [C.c_void_p, C.c_double]; L.orc_hier_set_correction_scale.restype = None
        L.orc_hier_A.argtypes = [C.c_void_p, C.c_int]; L.orc_hier_A.restype = cp
        L.orc_vcycle.argtypes = [C.c_void_p, dp, dp, C.c_int]; L.orc_vcycle.restype = None
        L.orc_bicgstab.argtypes = [cp, dp, dp, C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_double)]
        L.orc_rand_rhs.argtypes = [C.c_uint, C.c_int, dp]; L.orc_rand_rhs.restype = None
        L.orc_poisson2d.argtypes = [C.c_int, cp]
        L.orc_poisson3d.argtypes = [C.c_int, cp]
        L.orc_agmg.argtypes = [cp, C.c_double, C.c_int, C.c_double, C.c_int, cp]
        L.orc_set_threads.argtypes = [C.c_int]; L.orc_set_threads.restype = None
        L.orc_get_threads.argtypes = []; L.orc_get_threads.restype = C.c_int
        _LIB = L
    return _LIB


def set_threads(n):
    """host threads of the oracle's row loops (default 1; the bits do not depend on it — rows are independent)"""
    lib().orc_set_threads(int(n))
    return lib().orc_get_threads()


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


class Csr:
    """Owned CSR matrix living in the oracle library (f64 values, i32 indices)."""

    def __init__(self):
        self.c = OrcCsr()
        self._owned = False

    def __del__(self):
        if self._owned and lib is not None:
            try:
                lib().orc_csr_free(C.byref(self.c))
            except Exception:
                pass

    @property
    def shape(self):
        return (self.c.rows, self.c.cols)

    @property
    def nnz(self):
        return self.c.nnz

    @property
    def rowptr(self):
        return np.ctypeslib.as_array(self.c.rowptr, shape=(self.c.rows + 1,)).copy()

    @property
    def col(self):
        return np.ctypeslib.as_array(self.c.col, shape=(max(self.c.nnz, 1),))[: self.c.nnz].copy()

    @property
    def val(self):
        return np.ctypeslib.as_array(self.c.val, shape=(max(self.c.nnz, 1),))[: self.c.nnz].copy()

    def ref(self):
        return C.byref(self.c)

    def ptr(self):
        return C.pointer(self.c)

    @staticmethod
    def read(path):
        m = Csr()
        rc = lib().orc_mtx_read(path.encode(), m.ref())
        if rc:
            raise IOError(f"orc_mtx_read({path}) -> {rc}")
        m._owned = True
        return m

    @staticmethod
    def from_arrays(rows, cols, rowptr, col, val):
        rowptr = np.ascontiguousarray(rowptr, dtype=np.int32)
        col = np.ascontiguousarray(col, dtype=np.int32)
        val = np.ascontiguousarray(val, dtype=np.float64)
        m = Csr()
        rc = lib().orc_csr_from_arrays(rows, cols, len(col), _ip(rowptr), _ip(col), _dp(val), m.ref())
        assert rc == 0
        m._owned = True
        return m

    @staticmethod
    def from_scipy(sp):
        sp = sp.tocsr().sorted_indices()
        return Csr.from_arrays(sp.shape[0], sp.shape[1], sp.indptr, sp.indices, sp.data)

    def to_scipy(self):
        import scipy.sparse as sps
        return sps.csr_matrix((self.val, self.col, self.rowptr), shape=self.shape)

    def write(self, path):
        rc = lib().orc_mtx_write(path.encode(), self.ref())
        if rc:
            raise IOError(rc)

    def spmv(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        assert x.size == self.c.cols
        y = np.empty(self.c.rows)
        lib().orc_spmv(self.ref(), _dp(x), _dp(y))
        return y

    def transpose(self):
        m = Csr(); assert lib().orc_transpose(self.ref(), m.ref()) == 0; m._owned = True
        return m

    def matmul(self, other):
        m = Csr(); assert lib().orc_spgemm(self.ref(), other.ref(), m.ref()) == 0; m._owned = True
        return m

    def galerkin(self, P):
        m = Csr(); assert lib().orc_galerkin(self.ref(), P.ref(), m.ref()) == 0; m._owned = True
        return m

    def diag_inv(self):
        d = np.empty(self.c.rows); lib().orc_diag_inv(self.ref(), _dp(d)); return d

    def residual(self, x, b):
        x = np.ascontiguousarray(x, dtype=np.float64); b = np.ascontiguousarray(b, dtype=np.float64)
        r = np.empty(self.c.rows); lib().orc_residual(self.ref(), _dp(x), _dp(b), _dp(r)); return r

    def jacobi(self, dinv, omega, b, x):
        x = np.ascontiguousarray(x, dtype=np.float64); b = np.ascontiguousarray(b, dtype=np.float64)
        dinv = np.ascontiguousarray(dinv, dtype=np.float64)
        out = np.empty(self.c.rows); lib().orc_jacobi(self.ref(), _dp(dinv), omega, _dp(b), _dp(x), _dp(out)); return out

    def agmg(self, ktg=10.0, npass=2, tou=8.0, max_restriction=0, strict=True):
        """strict=False accepts rc=-2: P is complete, but the reference itself would have stopped
        on its assert(i < j) (AGMG.cpp:156) for this numbering."""
        m = Csr()
        rc = lib().orc_agmg(self.ref(), ktg, npass, tou, max_restriction, m.ref())
        if rc and not (rc == -2 and not strict):
            raise RuntimeError(f"orc_agmg -> {rc}")
        m._owned = True
        return m


def poisson2d(n):
    m = Csr(); assert lib().orc_poisson2d(n, m.ref()) == 0; m._owned = True; return m


def poisson3d(N):
    m = Csr(); assert lib().orc_poisson3d(N, m.ref()) == 0; m._owned = True; return m


def rand_rhs(n, seed=0):
    b = np.empty(n); lib().orc_rand_rhs(seed, n, _dp(b)); return b


def twogrid_jacobi(A, P, omega, v):
    v = np.ascontiguousarray(v, dtype=np.float64)
    x = np.empty(A.c.rows)
    assert lib().orc_twogrid_jacobi(A.ref(), P.ref(), omega, _dp(v), _dp(x)) == 0
    return x


class Hier:
    def __init__(self, A0, Ps, omega=0.5, nu1=1, nu2=1, As=None):
        n = len(Ps)
        self._keep = (A0, Ps, As)
        parr = (C.POINTER(OrcCsr) * max(n, 1))(*[p.ptr() for p in Ps])
        if As is None:
            self.h = lib().orc_hier_create_from_P(A0.ref(), n, parr, omega, nu1, nu2)
        else:
            aarr = (C.POINTER(OrcCsr) * len(As))(*[a.ptr() for a in As])
            self.h = lib().orc_hier_create(len(As), aarr, parr, omega, nu1, nu2)
        if not self.h:
            raise RuntimeError("orc_hier_create failed")
        self.n = A0.c.rows

    def __del__(self):
        try:
            if self.h:
                lib().orc_hier_destroy(self.h)
        except Exception:
            pass

    @property
    def nlev(self):
        return lib().orc_hier_nlev(self.h)

    def set_smoother(self, omega, nu1, nu2):
        lib().orc_hier_set_smoother(self.h, omega, nu1, nu2)
        return self

    def set_kcycle(self, levels):
        lib().orc_hier_set_kcycle(self.h, levels)
        return self

    def set_kcycle_energy(self, on=True):
        """K-cycle coefficients from energy inner products (flexible-CG form, SPD operators) instead of the GCR form"""
        lib().orc_hier_set_kcycle_energy(self.h, int(bool(on)))
        return self

    def set_correction_scale(self, sigma):
        """x ← x + σ·P e_c on every level (derived knob; σ = 1 is the reference's form)"""
        lib().orc_hier_set_correction_scale(self.h, float(sigma))
        return self

    def set_additive(self, on=True):
        """additive form, reference src/common/bicg.cpp:59 (with M2 = ωD⁻¹)"""
        lib().orc_hier_set_additive(self.h, int(bool(on)))
        return self

    def A(self, l):
        c = lib().orc_hier_A(self.h, l).contents
        m = Csr(); m.c = c; m._owned = False
        m._keepalive = self
        return m

    def vcycle(self, b, x=None):
        b = np.ascontiguousarray(b, dtype=np.float64)
        if x is None:
            out = np.empty(self.n); lib().orc_vcycle(self.h, _dp(b), _dp(out), 1)
        else:
            out = np.array(x, dtype=np.float64); lib().orc_vcycle(self.h, _dp(b), _dp(out), 0)
        return out


def bicgstab(A, b, precond=None, max_iter=10000, tol=1e-6, x0=None):
    """precond: None (identity), a Hier (V-cycle), or a python callable v->out."""
    b = np.ascontiguousarray(b, dtype=np.float64)
    x = np.zeros(A.c.rows) if x0 is None else np.array(x0, dtype=np.float64)
    mi = C.c_int(max_iter); t = C.c_double(tol)
    n = A.c.rows
    if precond is None:
        fn, user = None, None
    elif isinstance(precond, Hier):
        fn = C.cast(lib().orc_precond_vcycle, C.c_void_p); user = C.c_void_p(precond.h)
    else:
        def _cb(_u, vp, op):
            v = np.ctypeslib.as_array(vp, shape=(n,))
            o = np.ctypeslib.as_array(op, shape=(n,))
            o[:] = precond(v.copy())
        cb = PRECOND_FN(_cb)
        fn = C.cast(cb, C.c_void_p); user = None
    st = lib().orc_bicgstab(A.ref(), _dp(x), _dp(b), fn, user, C.byref(mi), C.byref(t))
    return st, mi.value, t.value, x
