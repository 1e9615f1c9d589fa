"""ctypes loader of libmgs.so (the C-ABI in include/mgs.h).

Processes that also use PyTorch-ROCm (the multi-GPU launcher does, for torch.distributed) must import torch BEFORE
this library is loaded: torch ships its own HIP runtime and the first runtime loaded serves the whole process.

The library is HIP-only: if it is missing or cannot be loaded this module raises — there is
no Python/NumPy fallback for any compute entry point.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("MGS_LIBMGS") or os.path.join(HERE, "libmgs.so")      # MGS_LIBMGS: another build of the same ABI (A/B runs of tools/ab_lib.sh)

c_int_p = C.POINTER(C.c_int)
c_dbl_p = C.POINTER(C.c_double)
c_i64_p = C.POINTER(C.c_int64)
HALO_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p)
COARSE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p)
HALO_FUSED_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, c_dbl_p, C.c_int)
P2P_HANDLE_BYTES = 96          # MGS_P2P_HANDLE_BYTES

# name -> (restype, argtypes); every symbol declared in include/mgs.h
PROTOTYPES = {
    "mgs_ctx_create": (C.c_int, [C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]),
    "mgs_ctx_destroy": (C.c_int, [C.c_void_p]),
    "mgs_last_error": (C.c_char_p, [C.c_void_p]),
    "mgs_sync": (C.c_int, [C.c_void_p]),
    "mgs_ctx_trim": (C.c_int, [C.c_void_p]),
    "mgs_ctx_stream": (C.c_void_p, [C.c_void_p]),
    "mgs_version": (C.c_char_p, []),
    "mgs_mtx_read": (C.c_int, [C.c_char_p, c_int_p, c_int_p, c_int_p, C.POINTER(c_int_p), C.POINTER(c_int_p), C.POINTER(c_dbl_p)]),
    "mgs_mtx_write": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_int, c_int_p, c_int_p, c_dbl_p]),
    "mgs_host_free": (None, [C.c_void_p]),
    "mgs_csr_upload": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int64, c_int_p, c_int_p, c_dbl_p, C.POINTER(C.c_void_p)]),
    "mgs_csr_download": (C.c_int, [C.c_void_p, c_int_p, c_int_p, c_dbl_p]),
    "mgs_csr_shape": (C.c_int, [C.c_void_p, c_int_p, c_int_p, c_i64_p]),
    "mgs_csr_plan_info": (C.c_int, [C.c_void_p, c_i64_p]),
    "mgs_csr_get_origin": (C.c_int, [C.c_void_p, c_int_p]),
    "mgs_csr_set_origin": (C.c_int, [C.c_void_p, c_int_p]),
    "mgs_comm_unique_id": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p]),
    "mgs_comm_create": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "mgs_comm_destroy": (C.c_int, [C.c_void_p]),
    "mgs_comm_p2p_create": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.POINTER(C.c_void_p)]),
    "mgs_comm_p2p_connect": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mgs_comm_p2p_info": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mgs_comm_p2p_selftest": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_longlong)]),
    "mgs_comm_exchange_raw": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mgs_comm_allgather_raw": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "mgs_comm_allreduce_raw": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "mgs_comm_size": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "mgs_hier_set_native_exchange": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mgs_hier_set_native_tail": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mgs_hier_set_native_tail_halo": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "mgs_hier_native_halo": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "mgs_arena_reserve": (C.c_int, [C.c_size_t]),
    "mgs_arena_info": (C.c_int, [C.c_void_p]),
    "mgs_hier_native_send_segments": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]),
    "mgs_hier_set_native_recv_segments": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "mgs_ctx_set_native_allreduce": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mgs_csr_optimize": (C.c_int, [C.c_void_p]),
    "mgs_hier_fused_info": (C.c_int, [C.c_void_p, C.c_int, c_i64_p]),
    "mgs_csr_rowcode_info": (C.c_int, [C.c_void_p, c_i64_p]),
    "mgs_hier_graph_info": (C.c_int, [C.c_void_p, c_i64_p]),
    "mgs_hier_group_info": (C.c_int, [C.c_void_p, C.c_int, c_i64_p]),
    "mgs_csr_destroy": (C.c_int, [C.c_void_p]),
    "mgs_csr_device_ptrs": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "mgs_csr_poisson3d": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "mgs_csr_poisson2d": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    "mgs_csr_transpose": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "mgs_csr_galerkin": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]),
    "mgs_vec_create": (C.c_int, [C.c_void_p, C.c_int64, C.POINTER(C.c_void_p)]),
    "mgs_vec_wrap": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_void_p)]),
    "mgs_vec_destroy": (C.c_int, [C.c_void_p]),
    "mgs_vec_upload": (C.c_int, [C.c_void_p, c_dbl_p, C.c_int64]),
    "mgs_vec_download": (C.c_int, [C.c_void_p, c_dbl_p, C.c_int64]),
    "mgs_vec_fill": (C.c_int, [C.c_void_p, C.c_double]),
    "mgs_vec_copy": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mgs_vec_size": (C.c_int64, [C.c_void_p]),
    "mgs_vec_ptr": (C.c_void_p, [C.c_void_p]),
    "mgs_vec_rand": (C.c_int, [C.c_void_p, C.c_uint64, C.c_int64]),
    "mgs_spmv": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "mgs_residual": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mgs_diag_inv": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mgs_jacobi": (C.c_int, [C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mgs_xfer_create": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "mgs_xfer_destroy": (C.c_int, [C.c_void_p]),
    "mgs_xfer_shape": (C.c_int, [C.c_void_p, c_int_p, c_int_p, c_int_p]),
    "mgs_xfer_download_agg": (C.c_int, [C.c_void_p, c_int_p]),
    "mgs_restrict": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "mgs_prolong": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "mgs_prolong_add": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "mgs_dot": (C.c_int, [C.c_void_p, C.c_void_p, c_dbl_p]),
    "mgs_nrm2": (C.c_int, [C.c_void_p, c_dbl_p]),
    "mgs_axpby": (C.c_int, [C.c_double, C.c_void_p, C.c_double, C.c_void_p]),
    "mgs_axpbypcz": (C.c_int, [C.c_double, C.c_void_p, C.c_double, C.c_void_p, C.c_double, C.c_void_p]),
    "mgs_hier_create": (C.c_int, [C.c_void_p, C.c_void_p, C.c_double, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "mgs_hier_push_P": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mgs_hier_coarsen": (C.c_int, [C.c_void_p, C.c_double, C.c_int, C.c_double, C.c_int, C.c_int]),
    "mgs_hier_finalize": (C.c_int, [C.c_void_p]),
    "mgs_hier_set_smoother": (C.c_int, [C.c_void_p, C.c_double, C.c_int, C.c_int]),
    "mgs_hier_set_kcycle": (C.c_int, [C.c_void_p, C.c_int]),
    "mgs_hier_set_kcycle_entry": (C.c_int, [C.c_void_p, C.c_int]),
    "mgs_hier_set_additive": (C.c_int, [C.c_void_p, C.c_int]),
    "mgs_hier_set_correction_scale": (C.c_int, [C.c_void_p, C.c_double]),
    "mgs_hier_destroy": (C.c_int, [C.c_void_p]),
    "mgs_hier_nlev": (C.c_int, [C.c_void_p]),
    "mgs_hier_level_shape": (C.c_int, [C.c_void_p, C.c_int, c_int_p, c_i64_p]),
    "mgs_hier_level_A": (C.c_void_p, [C.c_void_p, C.c_int]),
    "mgs_hier_level_P": (C.c_void_p, [C.c_void_p, C.c_int]),
    "mgs_hier_vcycle_bytes": (C.c_int64, [C.c_void_p]),
    "mgs_vcycle": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "mgs_bicgstab": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, c_int_p, c_dbl_p, c_int_p]),
    "mgs_fgcr": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, c_int_p, c_dbl_p, c_int_p]),
    "mgs_halo_pack": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "mgs_hier_set_halo_exchange": (C.c_int, [C.c_void_p, HALO_FN, C.c_void_p]),
    "mgs_aggregate_shard": (C.c_int, [C.c_void_p, C.c_double, C.c_int, C.c_double, C.POINTER(C.c_void_p)]),
    "mgs_aggregate_shard_zoned": (C.c_int, [C.c_void_p, C.c_double, C.c_int, C.c_double, C.c_void_p, C.POINTER(C.c_void_p)]),
    "mgs_galerkin_shard": (C.c_int, [C.c_void_p, C.c_void_p, c_int_p, C.c_int, C.POINTER(C.c_void_p)]),
    "mgs_hier_push_level": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "mgs_xfer_from_agg": (C.c_int, [C.c_void_p, C.c_int, C.c_int, c_int_p, C.POINTER(C.c_void_p)]),
    "mgs_hier_set_coarse_solver": (C.c_int, [C.c_void_p, COARSE_FN, C.c_void_p]),
    "mgs_hier_set_halo_exchange_fused": (C.c_int, [C.c_void_p, HALO_FUSED_FN, C.c_void_p]),
    "mgs_hier_set_halo_exchange_split": (C.c_int, [C.c_void_p, HALO_FN, HALO_FN, C.c_void_p]),
    "mgs_ctx_set_allreduce": (C.c_int, [C.c_void_p, ALLREDUCE_FN, C.c_void_p]),
    "mgs_time_kernel": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, c_dbl_p]),
    "mgs_time_vcycle": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, c_dbl_p]),
    "mgs_ctx_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
}

_LIB = None


class MgsError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libmgs error {code}: {msg}")
        self.code = code


def lib():
    """Load libmgs.so (built by `make -C multigridsolver_amd/csrc` / __graft_entry__.build())."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(SO_PATH):
            raise ImportError(f"{SO_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        L = C.CDLL(SO_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(L, name)  # AttributeError if the .so lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB


def check(rc, ctx=None):
    if rc != 0:
        msg = lib().mgs_last_error(ctx)
        raise MgsError(rc, msg.decode() if msg else "?")
