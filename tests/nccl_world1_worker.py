"""world-1 NCCL (= RCCL) check of the transport pieces the N>1 run relies on and the shared-GPU gloo tests cannot
reach: all_to_all_single / all_gather_into_tensor / all_reduce on LIBRARY-OWNED device memory wrapped zero-copy
(__cuda_array_interface__), blocking and asynchronous, on the context's stream."""
import ctypes as C
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    import faulthandler; faulthandler.enable()
    import torch
    import torch.distributed as dist
    import multigridsolver_amd as mg
    from multigridsolver_amd import dist as mgd
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29911")
    os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
    dist.init_process_group("nccl")
    torch.cuda.set_device(0)
    stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
    ctx = mg.Context(0, stream.cuda_stream)
    comm = mgd.Comm()
    assert comm.nccl and comm.world == 1
    n, k = 100_000, 4096
    src_np = np.random.default_rng(0).standard_normal(n)
    x = ctx.vec(np.concatenate([src_np, np.zeros(k)]))          # owned entries + halo room, library memory
    sh = mgd.ShardedHierarchy.__new__(mgd.ShardedHierarchy)     # only the view helper is needed
    sh.torch, sh.comm, sh._views = torch, comm, {}
    buf = torch.empty(k, dtype=torch.float64, device=comm.device)
    idx = torch.arange(0, 4 * k, 4, dtype=torch.int32, device=comm.device)
    xv = mg.Vec.wrap(ctx, x.ptr, n); halo = mg.Vec.wrap(ctx, x.ptr + 8 * n, k)
    for async_op in (False, True):
        mg.core.check(mg.lib().mgs_vec_fill(halo.h, 0.0), ctx.h)
        mg.core.check(mg.lib().mgs_halo_pack(ctx.h, xv.h, C.c_void_p(idx.data_ptr()), k, C.c_void_p(buf.data_ptr())), ctx.h)
        recv = sh._view(x.ptr + 8 * n, k)
        w = comm.a2a_f64(recv, buf, [k], [k], async_op=async_op)
        if w is not None:
            w.wait()
        ctx.sync(); torch.cuda.synchronize()
        got = x.numpy()[n:]
        assert np.array_equal(got, src_np[0:4 * k:4]), ("a2a into library memory", async_op)
    # all-gather of the coarse-tail right-hand side from library memory, reductions of dots
    out = torch.empty(k, dtype=torch.float64, device=comm.device)
    comm.allgather_padded(out, sh._view(x.ptr, k))
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), src_np[:k])
    a = np.array([1.5, -2.0]); comm.allreduce_host(a); assert a.tolist() == [1.5, -2.0]
    assert comm.allgather_ints(7).tolist() == [7]
    lists = comm.exchange_lists([np.arange(5, dtype=np.int64)]); assert lists[0].tolist() == [0, 1, 2, 3, 4]
    # native RCCL transport inside the C++ cycle (self-exchange of a middle slab: every call the N>1 run makes, on one
    # GPU): must reproduce the torch.distributed callback path bit for bit, cycle and Krylov solve
    N = 48; lo, hi = 16, 32; n2 = N * N; n_loc = (hi - lo) * n2
    A = ctx.poisson3d(N, lo, hi, local_cols=True)
    ids = np.concatenate([np.arange(n_loc - n2, n_loc), np.arange(0, n2)]).astype(np.int32)
    res = {}
    for native in ("rccl", "p2p", False):       # both native transports (real RCCL at world 1; the peer-to-peer windows) and the callbacks
        if native:
            os.environ["MGS_NATIVE_TRANSPORT"] = native
        shh = mgd.ShardedHierarchy(ctx, A, mgd.LevelPlan(n_loc, [ids], [ids]), 0.6, 1, 1, comm)
        shh.overlap_min_rows = 0; ctx.set_option("split_min_rows", 0)
        shh.build(10.0, 2, 8.0, tail_rows=3000, coarse_rows=100, native=bool(native))
        assert shh.native == bool(native) and (not native or shh.native_transport == native) and len(shh.plans) >= 3, (shh.native, shh.native_transport, len(shh.plans))
        b = ctx.vec(n_loc).rand(seed=5); xx = ctx.vec(A.shape[1])
        shh.vcycle(b, xx)
        xs = ctx.vec(A.shape[1]).rand(seed=6); ys = ctx.vec(n_loc); shh.spmv(xs, ys)
        xsol = ctx.vec(A.shape[1]); st, it, tol = shh.bicgstab(xsol, b, 200, 1e-9)
        res[native] = (xx.numpy(n_loc), ys.numpy(), st, it, xsol.numpy(n_loc))
        shh.close(); del shh, b, xx, xs, ys, xsol
    for tr in ("rccl", "p2p"):
        assert np.array_equal(res[tr][0], res[False][0]) and np.array_equal(res[tr][1], res[False][1]), tr
        assert res[tr][2] == 0 and res[tr][2:4] == res[False][2:4] and np.array_equal(res[tr][4], res[False][4]), (tr, res[tr][2:4], res[False][2:4])
    # a rank that cannot resolve RCCL: with the default order the peer-to-peer transport serves; restricted to RCCL every rank falls back
    # to the callback path, and the cycle still runs
    os.environ["MGS_LIBRCCL"] = "/nonexistent/librccl.so"
    os.environ["MGS_NATIVE_TRANSPORT"] = "p2p,rccl"
    shh = mgd.ShardedHierarchy(ctx, A, mgd.LevelPlan(n_loc, [ids], [ids]), 0.6, 1, 1, comm)
    shh.build(10.0, 2, 8.0, tail_rows=3000, coarse_rows=100, native=True)
    assert shh.native is True and shh.native_transport == "p2p"
    shh.close(); del shh
    os.environ["MGS_NATIVE_TRANSPORT"] = "rccl"
    shh = mgd.ShardedHierarchy(ctx, A, mgd.LevelPlan(n_loc, [ids], [ids]), 0.6, 1, 1, comm)
    shh.build(10.0, 2, 8.0, tail_rows=3000, coarse_rows=100, native=True)
    assert shh.native is False
    b = ctx.vec(n_loc).rand(seed=5); xx = ctx.vec(A.shape[1]); shh.vcycle(b, xx)
    assert np.array_equal(xx.numpy(n_loc), res[False][0])
    shh.close(); del shh, b, xx
    del os.environ["MGS_LIBRCCL"]; del os.environ["MGS_NATIVE_TRANSPORT"]
    print("NCCL_W1_OK")
    ctx.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
