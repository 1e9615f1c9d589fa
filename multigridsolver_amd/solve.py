"""python -m multigridsolver_amd.solve <A.mtx> [--P <P.mtx>] [...]   — solve A x = b on one GPU, or on N GPUs when
launched by `python -m torch.distributed.run --nproc-per-node N -m multigridsolver_amd.solve ...` (rows sharded by
contiguous ranges, halo exchange over RCCL).  b is the reference's right-hand side (srand(0); rand()/RAND_MAX,
src/common/bicg.cpp:139,159-162); the two [info] lines of src/common/bicg.cpp:171-176 are printed by rank 0."""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np


def ref_rhs(n):
    libc = C.CDLL(None)
    libc.srand(0)
    libc.rand.restype = C.c_int
    return np.array([libc.rand() / 2147483647.0 for _ in range(n)])


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m multigridsolver_amd.solve")
    ap.add_argument("matrix")
    ap.add_argument("--P", default=None, help="prolongation .mtx from the AGMG setup (single GPU only); default: aggregate on device")
    ap.add_argument("--tol", type=float, default=1e-6)           # bicg.cpp:148
    ap.add_argument("--max-iter", type=int, default=10000)       # bicg.cpp:164
    ap.add_argument("--solver", choices=["bicgstab", "fgcr"], default="bicgstab")
    ap.add_argument("--kcycle", type=int, default=0)
    ap.add_argument("--kcycle-energy", action="store_true", help="K-cycle coefficients from energy inner products (flexible-CG form; symmetric positive definite operators only)")
    ap.add_argument("--omega", type=float, default=0.6)
    ap.add_argument("--nu1", type=int, default=1)
    ap.add_argument("--nu2", type=int, default=1)
    ap.add_argument("--ktg", type=float, default=10.0)
    ap.add_argument("--npass", type=int, default=2)
    ap.add_argument("--tou", type=float, default=8.0)
    ap.add_argument("--dump-x", default=None, help="write the solution (rank order, raw little-endian f64)")
    args = ap.parse_args(argv)
    world = int(os.environ.get("WORLD_SIZE", "1")); rank = int(os.environ.get("RANK", "0"))
    if world > 1:
        import torch  # noqa: F401  — before libmgs.so: both link a HIP runtime, the first one loaded serves the process
    import multigridsolver_amd as mg
    rows, cols, rp, ci, v = mg.read_mtx(args.matrix)
    if rows != cols:
        raise SystemExit("square operator required")
    bg = ref_rhs(rows)
    if world == 1:
        ctx = mg.Context(int(os.environ.get("LOCAL_RANK", "0")))
        if args.kcycle_energy:
            ctx.set_option("kcycle_energy", 1)
        A = ctx.csr(rows, cols, rp, ci, v)
        h = mg.Hierarchy(A, args.omega, args.nu1, args.nu2)
        if args.P:
            h.push_P(mg.Csr.from_mtx(ctx, args.P))
        h.coarsen(args.ktg, args.npass, args.tou).finalize().set_kcycle(args.kcycle)
        x, b = ctx.vec(rows), ctx.vec(bg)
        ctx.sync(); t0 = time.perf_counter()
        st, it, tol = (mg.bicgstab if args.solver == "bicgstab" else lambda *a: mg.fgcr(a[0], a[1], a[2], a[3], 10, a[4], a[5]))(A, x, b, h, args.max_iter, args.tol)
        ctx.sync(); dt = time.perf_counter() - t0
        xs = x.numpy()
    else:
        import torch
        import torch.distributed as dist
        from . import dist as mgd
        if args.P or args.solver != "bicgstab" or args.kcycle:
            raise SystemExit("multi-GPU: device aggregation + BiCGSTAB + V-cycle only")
        dist.init_process_group(backend=os.environ.get("MGS_DIST_BACKEND", "nccl"))
        dev = 0 if os.environ.get("MGS_DIST_SHARE_GPU") else int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(dev)
        stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
        ctx = mg.Context(dev, stream.cuda_stream)
        comm = mgd.Comm()
        lrp, lci, lv, ncols, plan = mgd.shard_from_global(rows, rp, ci, v, world, rank, comm.exchange_lists)
        A = ctx.csr(plan.n_loc, ncols, lrp, lci, lv)
        sh = mgd.ShardedHierarchy(ctx, A, plan, args.omega, args.nu1, args.nu2, comm).build(args.ktg, args.npass, args.tou, tail_rows=max(20000, rows // 50))
        lo, hi = mgd.row_ranges(rows, world)[rank]
        x, b = ctx.vec(ncols), ctx.vec(bg[lo:hi])
        ctx.sync(); comm.barrier(); t0 = time.perf_counter()      # collectives on the side stream (dist.py Comm: the cycle's stream gets captured)
        st, it, tol = sh.bicgstab(x, b, args.max_iter, args.tol)
        ctx.sync(); comm.barrier(); dt = time.perf_counter() - t0
        parts = comm.all_gather_object(x.numpy(plan.n_loc))
        xs = np.concatenate(parts)
    if rank == 0:
        sys.stderr.write("    \033[1;34m[time] \033[0m%-42s : %f.\n" % ("BiCGStab_SolveTimer", dt))
        if st == 0:
            print("    \033[32m\033[1m[info] \033[00m%-42s : %g." % ("Tolerance ", tol))
            print("    \033[32m\033[1m[info] \033[00m%-42s : %d." % ("Number of iterations BICG", it))
        else:
            print("BiCGSTABiml encountered a problem with status code: %d" % st)
        if args.dump_x:
            xs.astype("<f8").tofile(args.dump_x)
    if world > 1:
        import torch.distributed as dist
        dist.barrier(); dist.destroy_process_group()
    return 0 if st == 0 else 3


if __name__ == "__main__":
    sys.exit(main())
