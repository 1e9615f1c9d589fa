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
    print("NCCL_W1_OK")
    ctx.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
