"""Thin Python handles over the C-ABI (include/mgs.h).  Python here is plumbing for tests,
bench.py and the torch.distributed launcher; the host-side solver logic lives in C++
(multigridsolver_amd/csrc/mgs_api.hip, multigridsolver_amd/cpp/mgs_host.hpp)."""
import ctypes as C

import numpy as np

from ._lib import ALLREDUCE_FN, HALO_FN, HALO_FUSED_FN, MgsError, check, lib

OP_SPMV, OP_RESIDUAL, OP_JACOBI = 0, 1, 2


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class Context:
    def __init__(self, device=0, stream=None):
        h = C.c_void_p()
        check(lib().mgs_ctx_create(device, C.c_void_p(stream) if stream else None, C.byref(h)))
        self.h = h
        self._cbs = []

    def close(self):
        if self.h:
            lib().mgs_ctx_destroy(self.h)
            self.h = None

    def sync(self):
        check(lib().mgs_sync(self.h), self.h)

    def trim(self):
        """release the work vectors and scratch the context keeps between solves (mgs_ctx_trim)"""
        check(lib().mgs_ctx_trim(self.h), self.h)

    @property
    def stream(self):
        return lib().mgs_ctx_stream(self.h)

    def set_option(self, key, value):
        check(lib().mgs_ctx_set_option(self.h, key.encode(), int(value)), self.h)

    def set_allreduce(self, fn):
        """fn(np.ndarray of float64) -> None (in-place sum over ranks)"""
        def _cb(_u, ptr, count):
            try:
                fn(np.ctypeslib.as_array(ptr, shape=(count,)))
                return 0
            except Exception:  # noqa: BLE001
                import traceback; traceback.print_exc()
                return 1
        cb = ALLREDUCE_FN(_cb)
        self._cbs.append(cb)
        check(lib().mgs_ctx_set_allreduce(self.h, cb, None), self.h)

    # ---- factories
    def csr(self, rows, cols, rowptr, col, val):
        return Csr.upload(self, rows, cols, rowptr, col, val)

    def vec(self, n_or_array):
        if isinstance(n_or_array, (int, np.integer)):
            return Vec(self, int(n_or_array))
        a = np.ascontiguousarray(n_or_array, dtype=np.float64)
        v = Vec(self, a.size)
        v.upload(a)
        return v

    def poisson3d(self, N, plane_lo=0, plane_hi=None, local_cols=False):
        h = C.c_void_p()
        check(lib().mgs_csr_poisson3d(self.h, N, plane_lo, N if plane_hi is None else plane_hi, int(local_cols), C.byref(h)), self.h)
        return Csr(self, h)

    def poisson2d(self, n):
        h = C.c_void_p()
        check(lib().mgs_csr_poisson2d(self.h, n, C.byref(h)), self.h)
        return Csr(self, h)


def read_mtx(path):
    """readMatrix (reference src/common/MatrixIO.cpp:12-37) → (rows, cols, rowptr, col, val)"""
    rows, cols, nnz = C.c_int(), C.c_int(), C.c_int()
    rp, ci, v = C.POINTER(C.c_int)(), C.POINTER(C.c_int)(), C.POINTER(C.c_double)()
    check(lib().mgs_mtx_read(path.encode(), C.byref(rows), C.byref(cols), C.byref(nnz), C.byref(rp), C.byref(ci), C.byref(v)))
    try:
        rowptr = np.ctypeslib.as_array(rp, shape=(rows.value + 1,)).copy()
        col = np.ctypeslib.as_array(ci, shape=(max(nnz.value, 1),))[: nnz.value].copy()
        val = np.ctypeslib.as_array(v, shape=(max(nnz.value, 1),))[: nnz.value].copy()
    finally:
        lib().mgs_host_free(rp); lib().mgs_host_free(ci); lib().mgs_host_free(v)
    return rows.value, cols.value, rowptr, col, val


def write_mtx(path, rows, cols, rowptr, col, val):
    """writeMatrix (reference src/common/MatrixIO.cpp:39-57)"""
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int32); col = np.ascontiguousarray(col, dtype=np.int32)
    val = np.ascontiguousarray(val, dtype=np.float64)
    check(lib().mgs_mtx_write(path.encode(), rows, cols, len(col), _ip(rowptr), _ip(col), _dp(val)))


class Csr:
    def __init__(self, ctx, h, owned=True):
        self.ctx, self.h, self.owned = ctx, h, owned

    def __del__(self):
        try:
            if self.owned and self.h and self.ctx.h:
                lib().mgs_csr_destroy(self.h)
        except Exception:  # noqa: BLE001
            pass

    @staticmethod
    def upload(ctx, rows, cols, rowptr, col, val):
        rowptr = np.ascontiguousarray(rowptr, dtype=np.int32); col = np.ascontiguousarray(col, dtype=np.int32)
        val = np.ascontiguousarray(val, dtype=np.float64)
        h = C.c_void_p()
        check(lib().mgs_csr_upload(ctx.h, rows, cols, len(col), _ip(rowptr), _ip(col), _dp(val), C.byref(h)), ctx.h)
        return Csr(ctx, h)

    @staticmethod
    def from_mtx(ctx, path):
        return Csr.upload(ctx, *read_mtx(path))

    @property
    def shape(self):
        r, c, n = C.c_int(), C.c_int(), C.c_int64()
        lib().mgs_csr_shape(self.h, C.byref(r), C.byref(c), C.byref(n))
        return r.value, c.value

    @property
    def nnz(self):
        n = C.c_int64(); lib().mgs_csr_shape(self.h, None, None, C.byref(n)); return n.value

    def plan_info(self):
        out = (C.c_int64 * 8)(); lib().mgs_csr_plan_info(self.h, out)
        keys = ["max_block_nnz", "max_row_len", "far_band", "lds_bytes", "halo_lo_blocks", "halo_hi_blocks", "halo_split_ok", "has_long_row_blocks"]
        return dict(zip(keys, [int(v) for v in out]))

    def origin(self):
        """finest-level row every row descends from (None: identity) — tie-break space of the device matching"""
        out = np.empty(max(self.shape[0], 1), dtype=np.int32)
        rc = lib().mgs_csr_get_origin(self.h, _ip(out))
        if rc == 1:
            return None
        check(rc, self.ctx.h); return out[: self.shape[0]]

    def set_origin(self, origin):
        o = np.ascontiguousarray(origin, dtype=np.int32)
        check(lib().mgs_csr_set_origin(self.h, _ip(o)), self.ctx.h); return self

    def optimize(self):
        """build the pattern code of the column array (mgs_csr_optimize); returns self"""
        check(lib().mgs_csr_optimize(self.h), self.ctx.h); return self

    def rowcode_info(self):
        out = (C.c_int64 * 4)(); lib().mgs_csr_rowcode_info(self.h, out)
        return dict(zip(["coded_blocks", "blocks", "table_ints", "table_budget"], [int(v) for v in out]))

    def download(self):
        rows, _ = self.shape; nnz = self.nnz
        rp = np.empty(rows + 1, dtype=np.int32); ci = np.empty(max(nnz, 1), dtype=np.int32); v = np.empty(max(nnz, 1))
        check(lib().mgs_csr_download(self.h, _ip(rp), _ip(ci), _dp(v)), self.ctx.h)
        return rp, ci[:nnz], v[:nnz]

    def transpose(self):
        h = C.c_void_p(); check(lib().mgs_csr_transpose(self.h, C.byref(h)), self.ctx.h); return Csr(self.ctx, h)

    def galerkin(self, xfer):
        h = C.c_void_p(); check(lib().mgs_csr_galerkin(self.h, xfer.h, C.byref(h)), self.ctx.h); return Csr(self.ctx, h)

    def spmv(self, x, y=None):
        y = y or Vec(self.ctx, self.shape[0])
        check(lib().mgs_spmv(self.h, x.h, y.h), self.ctx.h); return y

    def residual(self, x, b, r=None):
        r = r or Vec(self.ctx, self.shape[0])
        check(lib().mgs_residual(self.h, x.h, b.h, r.h), self.ctx.h); return r

    def diag_inv(self):
        d = Vec(self.ctx, self.shape[0]); check(lib().mgs_diag_inv(self.h, d.h), self.ctx.h); return d

    def jacobi(self, dinv, omega, b, x, out=None):
        out = out or Vec(self.ctx, max(self.shape))
        check(lib().mgs_jacobi(self.h, dinv.h, omega, b.h, x.h, out.h), self.ctx.h); return out

    def time_kernel(self, op, x, b=None, dinv=None, out=None, reps=20):
        out = out or Vec(self.ctx, max(self.shape))
        ms = C.c_double()
        check(lib().mgs_time_kernel(self.h, op, x.h, b.h if b else None, dinv.h if dinv else None, out.h, reps, C.byref(ms)), self.ctx.h)
        return ms.value


class Vec:
    def __init__(self, ctx, n=None, h=None, owned=True):
        self.ctx = ctx
        if h is None:
            h = C.c_void_p(); check(lib().mgs_vec_create(ctx.h, n, C.byref(h)), ctx.h)
        self.h, self.owned = h, owned

    @staticmethod
    def wrap(ctx, device_ptr, n):
        h = C.c_void_p(); check(lib().mgs_vec_wrap(ctx.h, C.c_void_p(device_ptr), n, C.byref(h)), ctx.h)
        return Vec(ctx, h=h)

    def __del__(self):
        try:
            if self.h and self.ctx.h:
                lib().mgs_vec_destroy(self.h)
        except Exception:  # noqa: BLE001
            pass

    def __len__(self):
        return lib().mgs_vec_size(self.h)

    @property
    def ptr(self):
        return lib().mgs_vec_ptr(self.h)

    def upload(self, a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        check(lib().mgs_vec_upload(self.h, _dp(a), a.size), self.ctx.h); return self

    def numpy(self, n=None):
        n = len(self) if n is None else n
        out = np.empty(n); check(lib().mgs_vec_download(self.h, _dp(out), n), self.ctx.h); return out

    def fill(self, v):
        check(lib().mgs_vec_fill(self.h, float(v)), self.ctx.h); return self

    def rand(self, seed=0, offset=0):
        check(lib().mgs_vec_rand(self.h, seed, offset), self.ctx.h); return self

    def copy_from(self, src):
        check(lib().mgs_vec_copy(src.h, self.h), self.ctx.h); return self

    def dot(self, other):
        out = C.c_double(); check(lib().mgs_dot(self.h, other.h, C.byref(out)), self.ctx.h); return out.value

    def nrm2(self):
        out = C.c_double(); check(lib().mgs_nrm2(self.h, C.byref(out)), self.ctx.h); return out.value

    def axpby(self, a, x, b):
        """self = a·x + b·self"""
        check(lib().mgs_axpby(float(a), x.h, float(b), self.h), self.ctx.h); return self

    def axpbypcz(self, a, x, b, y, c):
        """self = a·x + b·y + c·self"""
        check(lib().mgs_axpbypcz(float(a), x.h, float(b), y.h, float(c), self.h), self.ctx.h); return self


class Xfer:
    """Prolongation P (+ restriction Pᵀ) — reference bicg.cpp:32,48."""

    def __init__(self, ctx, h, owned=True):
        self.ctx, self.h, self.owned = ctx, h, owned

    @staticmethod
    def from_csr(P):
        h = C.c_void_p(); check(lib().mgs_xfer_create(P.h, C.byref(h)), P.ctx.h); return Xfer(P.ctx, h)

    def __del__(self):
        try:
            if self.owned and self.h and self.ctx.h:
                lib().mgs_xfer_destroy(self.h)
        except Exception:  # noqa: BLE001
            pass

    @property
    def shape(self):
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        lib().mgs_xfer_shape(self.h, C.byref(a), C.byref(b), C.byref(c)); return a.value, b.value

    @property
    def is_aggregation(self):
        c = C.c_int(); lib().mgs_xfer_shape(self.h, None, None, C.byref(c)); return bool(c.value)

    def agg(self):
        out = np.empty(self.shape[0], dtype=np.int32); check(lib().mgs_xfer_download_agg(self.h, _ip(out)), self.ctx.h); return out

    def restrict(self, r, rc=None):
        rc = rc or Vec(self.ctx, self.shape[1]); check(lib().mgs_restrict(self.h, r.h, rc.h), self.ctx.h); return rc

    def prolong(self, ec, e=None):
        e = e or Vec(self.ctx, self.shape[0]); check(lib().mgs_prolong(self.h, ec.h, e.h), self.ctx.h); return e

    def prolong_add(self, ec, x):
        check(lib().mgs_prolong_add(self.h, ec.h, x.h), self.ctx.h); return x


class Hierarchy:
    """MultiGridPrecond state (reference bicg.cpp:19-62) generalised to a multilevel V-cycle."""

    def __init__(self, A, omega=0.5, nu1=1, nu2=1):
        self.ctx, self.A = A.ctx, A
        h = C.c_void_p(); check(lib().mgs_hier_create(A.ctx.h, A.h, omega, nu1, nu2, C.byref(h)), A.ctx.h)
        self.h = h
        self._cb = None

    def __del__(self):
        try:
            if self.h and self.ctx.h:
                lib().mgs_hier_destroy(self.h)
        except Exception:  # noqa: BLE001
            pass

    def push_P(self, P):
        check(lib().mgs_hier_push_P(self.h, P.h), self.ctx.h); return self

    def coarsen(self, ktg=10.0, npass=2, tou=8.0, coarse_rows=2500, max_levels=32):
        check(lib().mgs_hier_coarsen(self.h, ktg, npass, tou, coarse_rows, max_levels), self.ctx.h); return self

    def finalize(self):
        check(lib().mgs_hier_finalize(self.h), self.ctx.h); return self

    def set_smoother(self, omega, nu1, nu2):
        check(lib().mgs_hier_set_smoother(self.h, omega, nu1, nu2), self.ctx.h); return self

    def set_kcycle(self, levels):
        check(lib().mgs_hier_set_kcycle(self.h, levels), self.ctx.h); return self

    def set_correction_scale(self, sigma):
        """over-correction x ← x + σ·P e_c on every level (σ = 1: the reference's form)"""
        check(lib().mgs_hier_set_correction_scale(self.h, float(sigma)), self.ctx.h); return self

    def set_additive(self, on=True):
        """additive form of solve() (reference bicg.cpp:59)"""
        check(lib().mgs_hier_set_additive(self.h, int(bool(on))), self.ctx.h); return self

    @property
    def nlev(self):
        return lib().mgs_hier_nlev(self.h)

    def level_shape(self, l):
        r, n = C.c_int(), C.c_int64(); check(lib().mgs_hier_level_shape(self.h, l, C.byref(r), C.byref(n)), self.ctx.h)
        return r.value, n.value

    def level_A(self, l):
        return Csr(self.ctx, C.c_void_p(lib().mgs_hier_level_A(self.h, l)), owned=False)

    def level_P(self, l):
        return Xfer(self.ctx, C.c_void_p(lib().mgs_hier_level_P(self.h, l)), owned=False)

    @property
    def vcycle_bytes(self):
        return lib().mgs_hier_vcycle_bytes(self.h)

    def vcycle(self, b, x=None, zero_guess=True):
        x = x or Vec(self.ctx, len(b))
        check(lib().mgs_vcycle(self.h, b.h, x.h, int(zero_guess)), self.ctx.h); return x

    def solve(self, v):
        """MultiGridPrecond::solve (reference bicg.cpp:51-61)"""
        return self.vcycle(v, None, True)

    def fused_info(self, level):
        out = (C.c_int64 * 6)(); check(lib().mgs_hier_fused_info(self.h, level, out), self.ctx.h)
        return dict(zip(["blocks", "has_val_wd", "has_col_agg", "coded_col", "coded_col_halo", "coded_col_agg"], [int(v) for v in out]))

    def group_info(self, level):
        out = (C.c_int64 * 4)(); check(lib().mgs_hier_group_info(self.h, level, out), self.ctx.h)
        return dict(zip(["groups", "paired_groups", "stray_aggregates", "blocks"], [int(v) for v in out]))

    def graph_info(self):
        out = (C.c_int64 * 4)(); check(lib().mgs_hier_graph_info(self.h, out), self.ctx.h)
        return dict(zip(["captured_cycles", "native_transport", "native_capture_failed", "native_eager_runs"], [int(v) for v in out]))

    def time_vcycle(self, b, x, reps=10):
        ms = C.c_double(); check(lib().mgs_time_vcycle(self.h, b.h, x.h, reps, C.byref(ms)), self.ctx.h); return ms.value

    def set_halo_exchange_split(self, begin, end):
        """begin(level, x_dev_ptr) starts the exchange, end(level, x_dev_ptr) waits for it"""
        def mk(fn):
            def _cb(_u, level, ptr):
                try:
                    fn(level, ptr); return 0
                except Exception:  # noqa: BLE001
                    import traceback; traceback.print_exc(); return 1
            return HALO_FN(_cb)
        self._cb2 = (mk(begin), mk(end))
        check(lib().mgs_hier_set_halo_exchange_split(self.h, self._cb2[0], self._cb2[1], None), self.ctx.h)

    def set_halo_exchange_fused(self, fn):
        """fn(level, kind, a_ptr, b_ptr, halo_out_ptr, phase) — payload exchange of the fused passes"""
        def _cb(_u, level, kind, a, b, out, phase):
            try:
                fn(level, kind, a, b, out, phase); return 0
            except Exception:  # noqa: BLE001
                import traceback; traceback.print_exc(); return 1
        self._cb3 = HALO_FUSED_FN(_cb)
        check(lib().mgs_hier_set_halo_exchange_fused(self.h, self._cb3, None), self.ctx.h)

    def set_halo_exchange(self, fn):
        """fn(level:int, x_dev_ptr:int) -> None"""
        def _cb(_u, level, ptr):
            try:
                fn(level, ptr); return 0
            except Exception:  # noqa: BLE001
                import traceback; traceback.print_exc(); return 1
        self._cb = HALO_FN(_cb)
        check(lib().mgs_hier_set_halo_exchange(self.h, self._cb, None), self.ctx.h)


def bicgstab(A, x, b, hier=None, max_iter=10000, tol=1e-6):
    """BiCGSTABiml (reference bicg.cpp:74-136) → (status, iterations, achieved_tol)"""
    mi, t, st = C.c_int(max_iter), C.c_double(tol), C.c_int(-1)
    check(lib().mgs_bicgstab(A.h, x.h, b.h, hier.h if hier else None, C.byref(mi), C.byref(t), C.byref(st)), A.ctx.h)
    return st.value, mi.value, t.value


def fgcr(A, x, b, hier=None, restart=10, max_iter=1000, tol=1e-6):
    """flexible GCR(restart) for the K-cycle preconditioner → (status, iterations, achieved_tol)"""
    mi, t, st = C.c_int(max_iter), C.c_double(tol), C.c_int(-1)
    check(lib().mgs_fgcr(A.h, x.h, b.h, hier.h if hier else None, restart, C.byref(mi), C.byref(t), C.byref(st)), A.ctx.h)
    return st.value, mi.value, t.value
