"""Worker of tests/test_gpu_dist.py::test_p2p_*: two PROCESSES on GPU 0 drive the peer-to-peer transport through the raw C-ABI
(no torch, no hierarchy).  argv: <rank> <dir> <mode>.  Handles travel through files in <dir>.
mode "exchange": both ranks exchange, all-gather and all-reduce a few times and check the data;
mode "timeout":  rank 1 connects and never exchanges; rank 0 (MGS_P2P_TIMEOUT_S=1) must get its error word set, and later exchanges return at once."""
import ctypes as C
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    rank, d, mode = int(sys.argv[1]), sys.argv[2], sys.argv[3]
    import multigridsolver_amd as mg
    from multigridsolver_amd._lib import P2P_HANDLE_BYTES, check, lib
    ctx = mg.Context(0)
    n = 5000
    h = C.create_string_buffer(P2P_HANDLE_BYTES); c = C.c_void_p()
    check(lib().mgs_comm_p2p_create(ctx.h, 2, rank, 2 * n, h, C.byref(c)), ctx.h)
    with open(os.path.join(d, f"h{rank}.tmp"), "wb") as f:
        f.write(h.raw)
    os.replace(os.path.join(d, f"h{rank}.tmp"), os.path.join(d, f"h{rank}"))
    t0 = time.time()
    while not os.path.exists(os.path.join(d, f"h{1 - rank}")):
        assert time.time() - t0 < 60
        time.sleep(0.01)
    both = b"".join(open(os.path.join(d, f"h{r}"), "rb").read() for r in (0, 1))
    check(lib().mgs_comm_p2p_connect(c, C.create_string_buffer(both, len(both))), ctx.h)

    def info():
        out = (C.c_longlong * 6)(); check(lib().mgs_comm_p2p_info(c, out), ctx.h); return [int(v) for v in out]

    def exchange(src, dst, cnt):
        peer = (C.c_int * 2)(1 - rank, 1 - rank); counts = (C.c_size_t * 2)(cnt, cnt)
        sp = (C.c_void_p * 2)(src.ptr, None); rp = (C.c_void_p * 2)(None, dst.ptr)
        check(lib().mgs_comm_exchange_raw(c, 2, peer, counts, sp, rp), ctx.h)

    if mode == "timeout":
        if rank == 1:
            time.sleep(4.0)               # connected, silent: the peer's wait must end by itself
            print("P2P_PEER_SILENT", flush=True)
            os._exit(0)
        src = ctx.vec(np.arange(n, dtype=np.float64)); dst = ctx.vec(n)
        t0 = time.time(); exchange(src, dst, n); ctx.sync(); t1 = time.time() - t0
        e = info()
        assert e[4] != 0 and 0.8 <= t1 < 3.5, (e, t1)          # the wait ended after MGS_P2P_TIMEOUT_S = 1 s with the error word set
        t0 = time.time()
        for _ in range(20):
            exchange(src, dst, n)
        ctx.sync(); t2 = time.time() - t0
        assert t2 < 0.5 and info()[4] != 0, t2                  # a dead transport does not wait again
        print("P2P_TIMEOUT_OK", flush=True)
        os._exit(0)
    for it in range(6):
        src = ctx.vec(1000.0 * rank + it + 1e-3 * np.arange(n)); dst = ctx.vec(n).fill(-1.0)
        exchange(src, dst, n - it)                              # odd and even lengths, both window slots
        ctx.sync()
        got = dst.numpy()
        assert np.array_equal(got[:n - it], 1000.0 * (1 - rank) + it + 1e-3 * np.arange(n - it)) and np.all(got[n - it:] == -1.0), it
        allv = ctx.vec(2 * n)
        check(lib().mgs_comm_allgather_raw(c, C.c_void_p(src.ptr), C.c_void_p(allv.ptr), n), ctx.h)
        red = ctx.vec(np.array([rank + 0.25, 2.0 * it, -1.0 - rank, 7.0]))
        check(lib().mgs_comm_allreduce_raw(c, C.c_void_p(red.ptr), 3), ctx.h)
        ctx.sync()
        a = allv.numpy()
        assert np.array_equal(a[:n], 0.0 + it + 1e-3 * np.arange(n)) and np.array_equal(a[n:], 1000.0 + it + 1e-3 * np.arange(n))
        assert np.array_equal(red.numpy(), np.array([1.5, 4.0 * it, -3.0, 7.0]))
    bad = C.c_longlong(-1)
    check(lib().mgs_comm_p2p_selftest(c, 60, C.byref(bad)), ctx.h)
    assert bad.value == 0 and info()[4] == 0 and info()[0] in (1, 2, 3)
    print(f"P2P_RAW_OK rank {rank} window memory kind {info()[0]}", flush=True)
    check(lib().mgs_comm_destroy(c), ctx.h)
    ctx.close()


if __name__ == "__main__":
    main()
