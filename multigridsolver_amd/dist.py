"""Row-sharded hierarchy over several GPUs of one node: one process per GPU, contiguous row
ranges, halo exchange of x before every kernel that gathers off-shard entries.

Design:
  * every level is a row shard  A_loc (n_loc × (n_loc + n_halo))  with LOCAL column numbering:
    owned rows first, then halo slots grouped by owner rank;
  * aggregates never straddle a shard (mgs_aggregate_shard pairs owned rows only), so restriction
    and prolongation need no communication; the only exchange step of the path is the halo of x
    (one pack kernel + one all_to_all over RCCL per SpMV-shaped kernel);
  * below `tail_rows` global rows the level is gathered once at setup and the rest of the
    hierarchy is replicated on every GPU (one small all-gather of the right-hand side per cycle
    instead of ~3 latency-bound exchanges per level);
  * the V-cycle itself stays in the C++ library (mgs_vcycle); this module only supplies the
    exchange / coarse-tail callbacks and the setup handshake.

torch.distributed is plumbing: backend "nccl" (= RCCL over xGMI) moves device buffers directly;
backend "gloo" (CPU tests, or several ranks sharing one GPU) stages through host memory.
"""
import ctypes as C
import os
from dataclasses import dataclass, field
from typing import List

import numpy as np

from . import core
from ._lib import COARSE_FN, check, lib


# ------------------------------------------------------------------ halo plans (pure host logic)
@dataclass
class LevelPlan:
    """send_idx[p]: owned rows this rank sends to peer p; recv_ids[p]: the peer-local row ids this
    rank receives from p, in halo-slot order (slots are grouped by peer rank ascending)."""
    n_loc: int
    send_idx: List[np.ndarray]
    recv_ids: List[np.ndarray]
    dev_send_idx: object = field(default=None, repr=False)   # device copy (torch tensor / Vec-like)

    @property
    def send_counts(self):
        return [int(len(a)) for a in self.send_idx]

    @property
    def recv_counts(self):
        return [int(len(a)) for a in self.recv_ids]

    @property
    def n_halo(self):
        return int(sum(self.recv_counts))


def poisson_plane_plan(N, world, rank):
    """Level-0 plan of the 7-point N^3 operator sharded by contiguous plane ranges, matching the
    local column numbering of mgs_csr_poisson3d(local_cols=1): lower halo plane, then upper."""
    lo, hi = plane_range(N, world, rank)
    n2 = N * N
    n_loc = (hi - lo) * n2
    send = [np.zeros(0, np.int32) for _ in range(world)]
    recv = [np.zeros(0, np.int32) for _ in range(world)]
    if rank > 0:
        plo, phi = plane_range(N, world, rank - 1)
        send[rank - 1] = np.arange(0, n2, dtype=np.int32)                                   # my first plane
        recv[rank - 1] = np.arange(((phi - plo) - 1) * n2, (phi - plo) * n2, dtype=np.int32)  # their last plane
    if rank < world - 1:
        send[rank + 1] = np.arange(n_loc - n2, n_loc, dtype=np.int32)                      # my last plane
        recv[rank + 1] = np.arange(0, n2, dtype=np.int32)                                   # their first plane
    return LevelPlan(n_loc, send, recv)


def plane_range(N, world, rank):
    base, rem = divmod(N, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def row_ranges(n, world):
    """contiguous, near-equal row ranges [lo, hi) per rank"""
    base, rem = divmod(n, world)
    out, lo = [], 0
    for r in range(world):
        hi = lo + base + (1 if r < rem else 0)
        out.append((lo, hi)); lo = hi
    return out


def shard_from_global(n, rowptr, col, val, world, rank, exchange_lists):
    """Row shard of a global CSR operator (any sparsity): rows [lo,hi) of this rank with LOCAL column numbering —
    owned columns first, then one halo slot per distinct off-shard column, grouped by owner rank (ascending) and
    sorted by global column inside a group; every row re-sorted by local column.  One handshake tells each owner
    which of its rows its peers need.  Returns (rowptr_loc, col_loc, val_loc, n_cols_loc, LevelPlan)."""
    import scipy.sparse as sps
    rng = row_ranges(n, world)
    lo, hi = rng[rank]
    n_loc = hi - lo
    starts = np.array([r[0] for r in rng] + [n], dtype=np.int64)
    rp = np.asarray(rowptr[lo:hi + 1], dtype=np.int64) - int(rowptr[lo])
    ci = np.asarray(col[rowptr[lo]:rowptr[hi]], dtype=np.int64)
    v = np.asarray(val[rowptr[lo]:rowptr[hi]], dtype=np.float64)
    off = (ci < lo) | (ci >= hi)
    halo_glob = np.unique(ci[off])                               # sorted ⇒ grouped by owner rank (contiguous ranges)
    owner = np.searchsorted(starts, halo_glob, side="right") - 1
    recv_ids = [(halo_glob[owner == p] - starts[p]).astype(np.int32) for p in range(world)]
    loc = np.where(off, n_loc + np.searchsorted(halo_glob, ci), ci - lo)
    m = sps.csr_matrix((v, loc, rp), shape=(n_loc, n_loc + len(halo_glob)))
    m.sort_indices()
    req = exchange_lists([r.astype(np.int64) for r in recv_ids])     # owners learn what to send
    plan = LevelPlan(n_loc, [np.asarray(a, dtype=np.int32) for a in req], recv_ids)
    return m.indptr.astype(np.int32), m.indices.astype(np.int32), m.data, n_loc + len(halo_glob), plan


def coarse_plan_handshake(plan, agg_loc, nc_loc, exchange_lists, span_slack=2.5):
    """One setup handshake per level (host arrays only).
    agg_loc: aggregate id of every owned row (−1 = none).  exchange_lists(list_per_peer) →
    list_per_peer is the variable-size all-to-all of int arrays.
    Returns (halo_coarse_col [n_halo ints], n_halo_coarse, coarse LevelPlan)."""
    world = len(plan.send_idx)
    # a. tell every peer the aggregate of the rows it sees as halo
    ragg = exchange_lists([agg_loc[idx].astype(np.int64) for idx in plan.send_idx])
    # b. distinct remote aggregates per peer → coarse halo slots (grouped by peer, ascending id).  Where the distinct ids nearly fill
    #    their span (plane shards: the aggregates of a boundary plane are a prefix of the peer's ids, or interleave with those of the
    #    plane behind it) the slots cover the whole span: the peer then sends ONE contiguous range of its vector, straight from memory
    #    (no pack kernel), for at most `span_slack` times the bytes; the extra slots are halo columns no matrix entry refers to.
    halo_cols, uniq, off = [], [], 0
    for p in range(world):
        r = np.asarray(ragg[p], dtype=np.int64)
        assert len(r) == len(plan.recv_ids[p]), "halo handshake: peer sent a wrong-sized list"
        u = np.unique(r[r >= 0])
        if len(u) and span_slack > 1.0:
            # clusters of ids (a gap wider than the whole list is never worth filling: both boundary planes of a peer that is this rank's
            # upper AND lower neighbour — two ranks, or the one-GPU rehearsal — stay two ranges); each cluster is filled if that costs little
            cuts = np.nonzero(np.diff(u) > len(u))[0] + 1
            parts = []
            for c in np.split(u, cuts):
                parts.append(np.arange(c[0], c[-1] + 1, dtype=np.int64) if (c[-1] - c[0] + 1) <= span_slack * len(c) else c)
            u = np.concatenate(parts)
        cols = np.full(len(r), -1, dtype=np.int32)
        ok = r >= 0
        cols[ok] = nc_loc + off + np.searchsorted(u, r[ok])
        halo_cols.append(cols); uniq.append(u.astype(np.int32)); off += len(u)
    # c. ask every peer for exactly those aggregates at the coarse level
    req = exchange_lists([u.astype(np.int64) for u in uniq])
    coarse = LevelPlan(nc_loc, [np.asarray(a, dtype=np.int32) for a in req], uniq)
    for a in coarse.send_idx:
        assert a.size == 0 or (a.min() >= 0 and a.max() < nc_loc), "peer requested an aggregate this rank does not own"
    hc = np.concatenate(halo_cols) if halo_cols else np.zeros(0, np.int32)
    return hc.astype(np.int32), off, coarse


def export_zones(plan):
    """zone id of every owned row for mgs_aggregate_shard_zoned: 0 = interior; the rows sent to peer p get a zone of their own per
    contiguous run of that peer's list (a row sent to several peers keeps the zone of the last one: still never paired with the interior)"""
    zone = np.zeros(plan.n_loc, dtype=np.int32)
    for p, idx in enumerate(plan.send_idx):
        idx = np.asarray(idx, dtype=np.int64)
        if idx.size == 0:
            continue
        run = np.concatenate([[0], np.cumsum(np.diff(idx) != 1)])
        zone[idx] = 1 + 16 * p + np.minimum(run, 15)
    return zone


def shard_to_global(plan, rowptr, col, val, offsets, rank):
    """local column numbering → global (offsets[p] = first global row of rank p); rows sorted."""
    import scipy.sparse as sps
    n_loc = plan.n_loc
    gmap = np.empty(n_loc + plan.n_halo, dtype=np.int64)
    gmap[:n_loc] = offsets[rank] + np.arange(n_loc)
    k = n_loc
    for p, ids in enumerate(plan.recv_ids):
        gmap[k:k + len(ids)] = offsets[p] + ids.astype(np.int64); k += len(ids)
    m = sps.csr_matrix((np.array(val, copy=True), gmap[col], np.array(rowptr, copy=True)), shape=(n_loc, int(offsets[-1])))
    m.sum_duplicates()      # two halo slots name the same global row only in the one-GPU rehearsal (a slab exchanging with itself)
    m.sort_indices()
    return m


# ------------------------------------------------------------------ transport
class Comm:
    """torch.distributed wrapper: device-direct for nccl (RCCL), host-staged for gloo."""

    def __init__(self, device=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.world, self.rank = dist.get_world_size(), dist.get_rank()
        self.nccl = dist.get_backend() == "nccl"
        self.device = device if device is not None else (torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu"))
        self.cdev = self.device if self.nccl else torch.device("cpu")
        # Host-synchronous collectives (setup handshakes, scalar reductions, barriers) run on a stream of their own: torch
        # executes a blocking collective on the CURRENT stream and keeps polling its completion event from a watchdog thread,
        # and an event that lives on the context's stream cannot be queried while that stream captures the cycle's hipGraph.
        self.side = torch.cuda.Stream(self.device) if self.device.type == "cuda" else None

    def host_ops(self):
        import contextlib
        return self.torch.cuda.stream(self.side) if self.side is not None else contextlib.nullcontext()

    def barrier(self):
        with self.host_ops():
            self.dist.barrier()
            if self.side is not None:
                self.side.synchronize()

    def exchange_lists(self, lists):
        """variable-size all-to-all of int64 numpy arrays (setup only)"""
        with self.host_ops():
            return self._exchange_lists(lists)

    def _exchange_lists(self, lists):
        t, d = self.torch, self.dist
        cnt = t.tensor([len(a) for a in lists], dtype=t.int64, device=self.cdev)
        rcnt = t.empty_like(cnt)
        d.all_to_all_single(rcnt, cnt)
        rc = rcnt.cpu().tolist()
        send = t.from_numpy(np.concatenate([np.asarray(a, dtype=np.int64) for a in lists]) if lists else np.zeros(0, np.int64)).to(self.cdev)
        recv = t.empty(int(sum(rc)), dtype=t.int64, device=self.cdev)
        d.all_to_all_single(recv, send, rc, [len(a) for a in lists])
        out, k, r = [], 0, recv.cpu().numpy()
        for c in rc:
            out.append(r[k:k + c].copy()); k += c
        return out

    def allgather_ints(self, v):
        t, d = self.torch, self.dist
        with self.host_ops():
            x = t.tensor([int(v)], dtype=t.int64, device=self.cdev)
            out = t.empty(self.world, dtype=t.int64, device=self.cdev)
            d.all_gather_into_tensor(out, x)
            return out.cpu().numpy()

    def allreduce_host(self, a, op="sum"):
        """in-place reduction of a float64 numpy array over ranks"""
        t, d = self.torch, self.dist
        with self.host_ops():
            x = t.from_numpy(a).to(self.cdev)
            d.all_reduce(x, op={"sum": d.ReduceOp.SUM, "max": d.ReduceOp.MAX, "min": d.ReduceOp.MIN}[op])
            a[:] = x.cpu().numpy()

    def all_gather_object(self, obj):
        out = [None] * self.world
        with self.host_ops():
            self.dist.all_gather_object(out, obj)
        return out

    def broadcast_object(self, obj, src=0):
        box = [obj]
        with self.host_ops():
            self.dist.broadcast_object_list(box, src=src)
        return box[0]

    def a2a_f64(self, recv, send, recv_counts, send_counts, async_op=False):
        """recv/send: 1-D float64 torch tensors on self.device.  async_op: returns a work handle
        (or None when the transfer already completed, i.e. on the host-staged path)."""
        d = self.dist
        if self.nccl or self.device.type == "cpu":
            w = d.all_to_all_single(recv, send, recv_counts, send_counts, async_op=async_op)
            return w if async_op else None
        else:  # gloo with device buffers: stage through the host
            s = send.cpu(); r = self.torch.empty(recv.numel(), dtype=recv.dtype)
            d.all_to_all_single(r, s, recv_counts, send_counts)
            recv.copy_(r)

    def allgather_padded(self, out_flat, send_padded):
        """out_flat: world*len(send_padded) entries, rank-major"""
        d = self.dist
        if self.nccl or self.device.type == "cpu":
            d.all_gather_into_tensor(out_flat, send_padded)
        else:
            r = self.torch.empty(out_flat.numel(), dtype=out_flat.dtype)
            d.all_gather_into_tensor(r, send_padded.cpu())
            out_flat.copy_(r)


class _DevPtr:
    """zero-copy torch view of library-owned device memory (__cuda_array_interface__)"""

    def __init__(self, ptr, n, typestr="<f8"):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": typestr, "data": (int(ptr), False), "version": 3, "strides": None}


# ------------------------------------------------------------------ sharded hierarchy (GPU)
class ShardedHierarchy:
    """Multilevel V-cycle preconditioner over row shards; API mirrors core.Hierarchy."""

    def __init__(self, ctx, A_local, plan0, omega=0.6, nu1=1, nu2=1, comm=None):
        import torch
        self.torch = torch
        self.ctx, self.A, self.comm = ctx, A_local, comm or Comm()
        self.plans = [plan0]
        self.h = core.Hierarchy(A_local, omega, nu1, nu2)
        self.smoother = (omega, nu1, nu2)
        self.tail = None
        self._views, self._bufs, self._keep, self._xc, self._pending, self._xv = {}, {}, [], {}, {}, {}
        self.n_exchanges = 0
        self.overlap_min_rows = 1_000_000

    # ---- device helpers
    def _view(self, ptr, n, typestr="<f8"):
        key = (ptr, n, typestr)
        v = self._views.get(key)
        if v is None:
            v = self.torch.as_tensor(_DevPtr(ptr, n, typestr), device=self.comm.device)
            self._views[key] = v
        return v

    def _prepare_plan(self, level):
        t = self.torch
        plan = self.plans[level]
        rows, cols = (self.A if level == 0 else self.h.level_A(level)).shape
        if plan.n_loc != rows or plan.n_halo != cols - rows or any(len(a) and (a.min() < 0 or a.max() >= rows) for a in plan.send_idx):
            raise ValueError(f"halo plan of level {level} does not fit its shard: plan {plan.n_loc} rows + {plan.n_halo} halo slots, "
                             f"operator {rows} x {cols}")
        idx = np.concatenate(plan.send_idx) if plan.send_idx else np.zeros(0, np.int32)
        plan.dev_send_idx = t.from_numpy(idx.astype(np.int32)).to(self.comm.device)
        self._bufs[level] = t.empty(max(len(idx), 1), dtype=t.float64, device=self.comm.device)

    def _exchange(self, level, x_ptr, async_op=False, sync_pack=False):
        """halo of x on `level`: pack kernel (gather of the owned rows peers need) on the context's
        stream, then one all_to_all straight into x's halo slots.  Everything per (level, pointer)
        is cached: the hot loop does two ctypes calls and one collective."""
        key = (level, x_ptr)
        c = self._xc.get(key)
        if c is None:
            plan = self.plans[level]
            ns, nr = sum(plan.send_counts), plan.n_halo
            c = (ns, nr, core.Vec.wrap(self.ctx, x_ptr, plan.n_loc) if ns else None,
                 C.c_void_p(plan.dev_send_idx.data_ptr()), C.c_void_p(self._bufs[level].data_ptr()),
                 self._view(x_ptr + 8 * plan.n_loc, max(nr, 1))[:nr], self._bufs[level][:ns], plan.recv_counts, plan.send_counts)
            self._xc[key] = c
        ns, nr, xv, idx_p, buf_p, recv, send, rcnt, scnt = c
        if ns == 0 and nr == 0:
            return None
        if ns:
            check(lib().mgs_halo_pack(self.ctx.h, xv.h, idx_p, ns, buf_p), self.ctx.h)
        if sync_pack:          # the collective runs on another stream than the context's (setup cross-check)
            self.ctx.sync()
        self.n_exchanges += 1
        return self.comm.a2a_f64(recv, send, rcnt, scnt, async_op=async_op)

    # split-phase form: the library runs the interior row blocks between begin and end
    def _exchange_begin(self, level, x_ptr):
        # asynchronous work objects cost ~25 us more host time than the blocking form (measured,
        # tools/studies_r1_r3/exchange_overhead.py): worth it only where the interior kernel is long enough to hide it
        big = self.plans[level].n_loc >= self.overlap_min_rows
        self._pending[level] = self._exchange(level, x_ptr, async_op=big)

    def _exchange_end(self, level, x_ptr):
        w = self._pending.pop(level, None)
        if w is not None:
            w.wait()

    def _exchange_fused(self, level, kind, a_ptr, b_ptr, out_ptr, phase):
        """halo values of the fused passes: out_ptr[slot] = the owner's a_ptr[row the slot stands for], on `level`'s plan (kind 2: plain
        values — the pre pass asks for the right-hand side into its payload buffer, the post pass for the coarse level's e_c into that
        vector's own halo room).  phase 0 packs and starts, phase 1 waits."""
        if phase == 1:
            w = self._pending.pop(("f", level), None)
            if w is not None:
                w.wait()
            return
        plan = self.plans[level]
        ns, nr = sum(plan.send_counts), plan.n_halo
        if ns == 0 and nr == 0:
            return
        buf = self._bufs[level]
        if ns:
            xv = self._xv.get((level, a_ptr))
            if xv is None:
                xv = self._xv[(level, a_ptr)] = core.Vec.wrap(self.ctx, a_ptr, plan.n_loc)
            check(lib().mgs_halo_pack(self.ctx.h, xv.h, C.c_void_p(plan.dev_send_idx.data_ptr()), ns, C.c_void_p(buf.data_ptr())), self.ctx.h)
        recv = self._view(out_ptr, max(nr, 1))[:nr]
        big = plan.n_loc >= self.overlap_min_rows
        self.n_exchanges += 1
        self._pending[("f", level)] = self.comm.a2a_f64(recv, buf[:ns], plan.recv_counts, plan.send_counts, async_op=big)

    # ---- setup
    def build(self, ktg=10.0, npass=2, tou=8.0, tail_rows=600_000, coarse_rows=2500, max_levels=32, log=None, overlap=True, fused=True, native=None, zoned=None):
        comm, ctx = self.comm, self.ctx
        if zoned is None:
            zoned = os.environ.get("MGS_SHARD_ZONES", "1") != "0"
        self._prepare_plan(0)
        A = self.A
        while True:
            plan = self.plans[-1]
            n_glob = int(comm.allgather_ints(plan.n_loc).sum())
            if n_glob <= tail_rows or len(self.plans) >= max_levels:
                break
            T = C.c_void_p()
            # zones: the rows a peer sees as halo pair among themselves only (one zone per peer and contiguous run of its list), so that peer's
            # requests at the next level are exactly their aggregates — a contiguous id range on plane shards, level after level
            zone = export_zones(plan) if zoned else None
            check(lib().mgs_aggregate_shard_zoned(A.h, ktg, npass, tou, zone.ctypes.data_as(C.c_void_p) if zone is not None else None, C.byref(T)), ctx.h)
            xf = core.Xfer(ctx, T, owned=True)
            agg = xf.agg(); nc_loc = xf.shape[1]
            if zone is not None and np.any(zone):
                # Zones pay while the exported rows still coarsen among themselves (plane shards: in-plane boxes, ÷4 per level).  Where the
                # operator's strong couplings point out of the exported layer (deeper levels: z), its rows stay singletons, the layer stops
                # shrinking and the coarse levels grow: this level is then aggregated without zones (its successor's lists go through the
                # pack kernel again; those levels are small).  A local decision: peers only ever see aggregate ids.
                ex = agg[zone > 0]
                n_ex, n_exc = int(ex.size), int(np.unique(ex[ex >= 0]).size) + int(np.count_nonzero(ex < 0))
                if n_exc * 2 > n_ex:
                    del xf
                    T = C.c_void_p()
                    check(lib().mgs_aggregate_shard_zoned(A.h, ktg, npass, tou, None, C.byref(T)), ctx.h)
                    xf = core.Xfer(ctx, T, owned=True)
                    agg = xf.agg(); nc_loc = xf.shape[1]
                    zoned = False              # deeper levels stall as well
                    if log:
                        log(f"level {len(self.plans) - 1}: exported rows do not coarsen among themselves ({n_ex} -> {n_exc}): aggregated without zones from here on")
            ncs = comm.allgather_ints(nc_loc)
            if int(ncs.sum()) > 0.9 * n_glob or int(ncs.min()) == 0:   # stalled somewhere: stop sharding here
                del xf
                break
            halo_cols, n_halo_c, cplan = coarse_plan_handshake(plan, agg, nc_loc, comm.exchange_lists)
            Ac = C.c_void_p()
            hc = np.ascontiguousarray(halo_cols, dtype=np.int32)
            check(lib().mgs_galerkin_shard(A.h, xf.h, hc.ctypes.data_as(C.POINTER(C.c_int)), n_halo_c, C.byref(Ac)), ctx.h)
            xf.owned = False                                     # ownership moves into the hierarchy
            check(lib().mgs_hier_push_level(self.h.h, xf.h, Ac), ctx.h)
            self.plans.append(cplan)
            self._prepare_plan(len(self.plans) - 1)
            A = self.h.level_A(len(self.plans) - 1)
            if log:
                log(f"sharded level {len(self.plans) - 1}: local {nc_loc} rows (+{cplan.n_halo} halo), global {int(ncs.sum())}")
        self._build_tail(ktg, npass, tou, coarse_rows, log)
        if overlap:
            self.h.set_halo_exchange_split(self._exchange_begin, self._exchange_end)
        else:
            self.h.set_halo_exchange(self._exchange)
        if fused:
            self.h.set_halo_exchange_fused(self._exchange_fused)
        # native transports, in the order tried (MGS_NATIVE_TRANSPORT, default "p2p,rccl"): "p2p" = the library's own peer-to-peer
        # exchange kernels over IPC-mapped device windows (csrc/comm_p2p.hip; any torch backend serves the setup handshakes), "rccl" =
        # ncclSend/ncclRecv groups (needs the nccl backend, or MGS_NATIVE_RCCL=force with a stand-in library: tests)
        env = os.environ.get("MGS_NATIVE_RCCL", "1")
        order = [t_ for t_ in os.environ.get("MGS_NATIVE_TRANSPORT", "p2p,rccl").split(",") if t_ in ("p2p", "rccl")]
        if env == "force":
            order = [t_ for t_ in order if t_ == "rccl"] if "MGS_NATIVE_TRANSPORT" not in os.environ else order
        elif not comm.nccl:
            order = [t_ for t_ in order if t_ == "p2p"]
        if native is None:
            native = env != "0" and bool(order) and comm.device.type == "cuda"
        self.native, self.native_transport = False, None
        if native:
            for tr in order:
                if self._enable_native(log, transport=tr):
                    self.native, self.native_transport = True, tr
                    break
        # capture the native cycle (RCCL exchanges included) in a hipGraph: MGS_NATIVE_GRAPH=0 keeps eager launches
        ctx.set_option("native_graph", 0 if os.environ.get("MGS_NATIVE_GRAPH", "1") == "0" else 1)
        return self

    # ---- native RCCL transport: the C++ cycle packs, exchanges (ncclSend/ncclRecv) and gathers the tail by itself
    def _enable_native(self, log=None, transport="rccl"):
        """Creates the library's own communicator — transport "rccl": an RCCL communicator (unique id shipped through
        torch.distributed); "p2p": IPC-mapped device windows, one exchange kernel per halo exchange — hands every
        sharded level's plan and the tail to the C++ cycle, and cross-checks native halo exchanges on every level
        against the torch.distributed path bit for bit on every rank.  Any failure on any rank → all ranks drop
        this transport (returns False)."""
        t, comm, ctx = self.torch, self.comm, self.ctx

        def agree(flag):
            """every collective step below is entered by all ranks or by none"""
            f = np.array([1.0 if flag else 0.0]); comm.allreduce_host(f, op="min"); return bool(f[0] > 0.5)

        ok = True
        self._seg_args = []
        tname = "native RCCL transport" if transport == "rccl" else "peer-to-peer transport"
        if transport == "rccl":
            # the RCCL copy this process already uses (torch's); MGS_LIBRCCL overrides the path
            path = os.environ.get("MGS_LIBRCCL", os.path.join(os.path.dirname(t.__file__), "lib", "librccl.so")).encode()
            idbuf = C.create_string_buffer(128)
            try:      # local preflight on every rank: the library resolves RCCL and can mint an id
                check(lib().mgs_comm_unique_id(ctx.h, path, idbuf), ctx.h)
            except Exception as e:  # noqa: BLE001
                ok = False
                if log:
                    log(f"native RCCL transport unavailable on this rank: {e!r}")
            if not agree(ok):
                if log:
                    log("exchange transport: torch.distributed callbacks")
                return False
            idbuf = C.create_string_buffer(comm.broadcast_object(idbuf.raw, src=0), 128)            # rank 0's id
        def fail(stage, e=None):
            if log:
                log(f"{tname}: {stage} failed" + (f" ({e!r})" if e is not None else "") + " -> next transport")
            self._drop_native()
            return False

        c = C.c_void_p()
        if transport == "rccl":
            try:      # collective: every rank is here
                check(lib().mgs_comm_create(ctx.h, path, idbuf, comm.world, comm.rank, C.byref(c)), ctx.h)
                self._ncomm = c
            except Exception as e:  # noqa: BLE001
                ok = False; err = e
            if not agree(ok):
                return fail("communicator creation", locals().get("err"))
        else:
            # window slot = the most doubles one peer ever stores here in one exchange (halo of any level, or its slice of the tail's
            # right-hand side); the same on every rank (windows are symmetric)
            need = np.array([float(max([64, int(self.tail_nlocs.max())] + [max(p.recv_counts + p.send_counts + [0]) for p in self.plans]))])
            comm.allreduce_host(need, op="max")
            from ._lib import P2P_HANDLE_BYTES
            hbuf = C.create_string_buffer(P2P_HANDLE_BYTES)
            try:
                if comm.world > 8:
                    raise RuntimeError("peer-to-peer transport: at most 8 ranks")
                check(lib().mgs_comm_p2p_create(ctx.h, comm.world, comm.rank, int(need[0]), hbuf, C.byref(c)), ctx.h)
                self._ncomm = c
            except Exception as e:  # noqa: BLE001
                ok = False; err = e
            if not agree(ok):
                return fail("window creation", locals().get("err"))
            handles = b"".join(comm.all_gather_object(hbuf.raw))          # rank order
            try:
                check(lib().mgs_comm_p2p_connect(c, C.create_string_buffer(handles, len(handles))), ctx.h)
            except Exception as e:  # noqa: BLE001
                ok = False; err = e
            if not agree(ok):
                return fail("mapping the peers' windows", locals().get("err"))
            # collective self-test before anything relies on the windows: 2400 pattern exchanges with every peer, verified on the device
            # (sizes 512 KiB ... 8 B, both window slots reused throughout; ~40 ms) — this is the first time the transport sees THIS machine's
            # links, and the ordering it ships (stores acknowledged, no cache maintenance: comm_p2p.hip) has only ever been verified between
            # processes on ONE GPU: a flag that overtakes its data on real links shows up here, on the large messages, and ends the generation
            try:
                bad = C.c_longlong(-1)
                check(lib().mgs_comm_p2p_selftest(c, int(os.environ.get("MGS_P2P_SELFTEST_ROUNDS", "2400")), C.byref(bad)), ctx.h)
                ok = bad.value == 0
                if not ok:
                    err = RuntimeError(f"{bad.value} wrong values received")
            except Exception as e:  # noqa: BLE001
                ok = False; err = e
            if not agree(ok):
                return fail("self-test of the windows", locals().get("err"))
        try:      # local: plans and tail into the C++ cycle
            ip = lambda a: np.ascontiguousarray(a, dtype=np.int32).ctypes.data_as(C.c_void_p)  # noqa: E731
            for l, plan in enumerate(self.plans):
                idx = np.concatenate(plan.send_idx).astype(np.int32) if plan.send_idx else np.zeros(0, np.int32)
                sc = np.asarray(plan.send_counts, dtype=np.int32); rc = np.asarray(plan.recv_counts, dtype=np.int32)
                check(lib().mgs_hier_set_native_exchange(self.h.h, l, c, ip(idx) if idx.size else None, ip(sc), ip(rc)), ctx.h)
            nl = np.ascontiguousarray(self.tail_nlocs, dtype=np.int32)
            check(lib().mgs_hier_set_native_tail(self.h.h, c, self.tail.h, nl.ctypes.data_as(C.c_void_p)), ctx.h)
            # global tail row of every halo slot of the last sharded level: the level above reads e_c straight from the tail's solution
            lp = self.plans[-1]
            if lp.n_halo and os.environ.get("MGS_TAIL_HALO", "1") != "0":
                hg = np.concatenate([self.tail_offs[p] + ids.astype(np.int64) for p, ids in enumerate(lp.recv_ids)]).astype(np.int32)
                check(lib().mgs_hier_set_native_tail_halo(self.h.h, hg.ctypes.data_as(C.c_void_p), int(hg.size)), ctx.h)
        except Exception as e:  # noqa: BLE001
            ok = False; err = e
        if not agree(ok):
            return fail("plan hand-over", locals().get("err"))
        # Pack-free exchanges: every rank reads the contiguous ranges it will send per peer from the library, ships their lengths,
        # and installs what it will receive — collective, all levels or none (a rank that cannot keeps every rank on packed sends).
        segs, ok = [], True
        try:
            # default: ranges on the peer-to-peer transport (a range is just another copy of the exchange kernel); packed messages on RCCL, where
            # ranges mean several ncclSend/ncclRecv per peer inside one group — verified on the stand-in only, never on real RCCL (MGS_NATIVE_SEGMENTS=1 opts in)
            if os.environ.get("MGS_NATIVE_SEGMENTS", "1" if transport == "p2p" else "0") == "0":
                raise RuntimeError("packed messages on this transport (MGS_NATIVE_SEGMENTS)")
            for l, plan in enumerate(self.plans):
                nseg = (C.c_int * comm.world)(); cap = sum(plan.send_counts) + comm.world + 1
                lens = (C.c_int * cap)()
                w = lib().mgs_hier_native_send_segments(self.h.h, l, nseg, lens, cap)
                if w < 0:
                    raise RuntimeError(f"mgs_hier_native_send_segments(level {l}) -> {w}")
                out, k = [], 0
                for p in range(comm.world):
                    out.append(np.array(lens[k:k + nseg[p]], dtype=np.int64)); k += nseg[p]
                segs.append(out)
        except Exception as e:  # noqa: BLE001
            ok = False; err = e
        if agree(ok):
            recv = [comm.exchange_lists(out) for out in segs]           # collective, level after level
            try:
                for l, r in enumerate(recv):
                    nseg = np.ascontiguousarray([len(a) for a in r], dtype=np.int32)
                    lens = np.ascontiguousarray(np.concatenate(r) if len(r) else np.zeros(0), dtype=np.int32)
                    if lens.size == 0:
                        lens = np.zeros(1, np.int32)
                    self._seg_args.append((l, nseg, lens))
            except Exception as e:  # noqa: BLE001
                ok = False; err = e
            if agree(ok):
                try:
                    for l, nseg, lens in self._seg_args:                # every rank is here: the switch to ranges happens on all of them
                        check(lib().mgs_hier_set_native_recv_segments(self.h.h, l, nseg.ctypes.data_as(C.c_void_p), lens.ctypes.data_as(C.c_void_p)), ctx.h)
                except Exception as e:  # noqa: BLE001
                    ok = False; err = e
                if not agree(ok):       # some rank sends ranges, another does not: no way back to a consistent native transport
                    return fail("installation of the receive ranges", locals().get("err"))
                self.native_segments = [[len(a) for a in out] for out in segs]
                if log:
                    log("native exchanges send contiguous row ranges straight from the vectors (ranges per peer and level: "
                        + " ".join(str(x) for x in self.native_segments) + ")")
            elif log:
                log(f"native exchanges keep the pack kernel ({locals().get('err')!r})")
        elif log:
            log(f"native exchanges keep the pack kernel ({locals().get('err')!r})")
        # collective: native halo exchanges per level against the torch.distributed exchange, bit for bit (p2p: three rounds with fresh
        # data, so both window slots are written, read and written again)
        for l, plan in enumerate(self.plans):
            for rnd in range(3 if transport == "p2p" else 1):
                try:
                    n_ext = plan.n_loc + plan.n_halo
                    xa = ctx.vec(n_ext).rand(seed=1234 + l + 100 * rnd, offset=comm.rank * 7919); xb = ctx.vec(n_ext)
                    check(lib().mgs_vec_copy(xa.h, xb.h), ctx.h)
                    ctx.sync()
                    with comm.host_ops():                      # the torch collective's events stay off the context's stream
                        self._exchange(l, xa.ptr, sync_pack=True); t.cuda.current_stream().synchronize()
                    check(lib().mgs_hier_native_halo(self.h.h, l, C.c_void_p(xb.ptr)), ctx.h)
                    ctx.sync(); t.cuda.synchronize()
                    same = bool(np.array_equal(xa.numpy(), xb.numpy()))
                except Exception as e:  # noqa: BLE001
                    same = False; err = e
                if not agree(same):
                    return fail(f"cross-check of the level-{l} exchange", locals().get("err"))
        if log:
            log(f"exchange transport: {tname} inside the C++ cycle (verified against torch.distributed)")
        # torch's collective watchdog retires finished work every 100 ms: let it finish before the first cycle is captured
        t.cuda.synchronize(); import time; time.sleep(0.35)
        return True

    def _drop_native(self):
        for l in range(len(self.plans)):
            lib().mgs_hier_set_native_exchange(self.h.h, l, None, None, None, None)
        lib().mgs_hier_set_native_tail(self.h.h, None, None, None)
        lib().mgs_ctx_set_native_allreduce(self.ctx.h, None)
        c = getattr(self, "_ncomm", None)
        if c is not None:
            lib().mgs_comm_destroy(c)
            self._ncomm = None
        self.native = False

    def _build_tail(self, ktg, npass, tou, coarse_rows, log):
        """gather the last sharded level and replicate the rest of the hierarchy on every GPU"""
        import scipy.sparse as sps
        comm, ctx, t = self.comm, self.ctx, self.torch
        L = len(self.plans) - 1
        plan = self.plans[L]
        nlocs = comm.allgather_ints(plan.n_loc)
        offs = np.concatenate([[0], np.cumsum(nlocs)]).astype(np.int64)
        rp, ci, v = self.h.level_A(L).download()
        mine = shard_to_global(plan, rp, ci, v, offs, comm.rank)
        # origins of the rows (tie-break space of the matching) travel with the operator, shifted to global finest-level rows
        org = self.h.level_A(L).origin() if L > 0 else None
        n0 = comm.allgather_ints(self.plans[0].n_loc)
        off0 = int(np.concatenate([[0], np.cumsum(n0)])[comm.rank])
        parts = comm.all_gather_object((mine.indptr, mine.indices, mine.data, None if org is None else org.astype(np.int64) + off0))
        Ag = sps.vstack([sps.csr_matrix((d, i, p), shape=(len(p) - 1, int(offs[-1]))) for (p, i, d, _) in parts]).tocsr()
        Ag.sort_indices()
        n_t = Ag.shape[0]
        self.tail_A = ctx.csr(n_t, n_t, Ag.indptr, Ag.indices, Ag.data)
        if all(q[3] is not None for q in parts) and int(np.sum(n0)) < 2 ** 31:
            self.tail_A.set_origin(np.concatenate([q[3] for q in parts]).astype(np.int32))
        # MGS_TAIL_NPASS: pairwise passes of the replicated tail's own coarsening (default: the hierarchy's; 3 = aggregates of up to 8 → fewer,
        # launch-bound levels per cycle on every rank)
        self.tail = core.Hierarchy(self.tail_A, *self.smoother).coarsen(ktg, int(os.environ.get("MGS_TAIL_NPASS", npass)), tou, coarse_rows, 32).finalize()
        if log:
            log(f"replicated tail from level {L}: {n_t} global rows, {self.tail.nlev} levels")
        self.tail_offs, self.tail_nlocs = offs, nlocs
        maxn = int(nlocs.max())
        self._tb_send = t.zeros(maxn, dtype=t.float64, device=comm.device)
        self._tb_all = t.zeros(comm.world * maxn, dtype=t.float64, device=comm.device)
        idx = np.concatenate([p * maxn + np.arange(nlocs[p]) for p in range(comm.world)]).astype(np.int64)
        self._tb_idx = t.from_numpy(idx).to(comm.device)
        self._tb_b = t.zeros(n_t, dtype=t.float64, device=comm.device)
        self._tb_x = t.zeros(n_t, dtype=t.float64, device=comm.device)
        self._tb_bv = core.Vec.wrap(ctx, self._tb_b.data_ptr(), n_t)
        self._tb_xv = core.Vec.wrap(ctx, self._tb_x.data_ptr(), n_t)
        n_loc, o = plan.n_loc, int(offs[comm.rank])

        def coarse(_u, b_ptr, x_ptr):
            try:
                self._tb_send[:n_loc].copy_(self._view(b_ptr, n_loc))
                comm.allgather_padded(self._tb_all, self._tb_send)
                t.index_select(self._tb_all, 0, self._tb_idx, out=self._tb_b)
                self.tail.vcycle(self._tb_bv, self._tb_xv, True)
                self._view(x_ptr, n_loc).copy_(self._tb_x[o:o + n_loc])
                return 0
            except Exception:  # noqa: BLE001
                import traceback; traceback.print_exc()
                return 1
        self._coarse_cb = COARSE_FN(coarse)
        check(lib().mgs_hier_set_coarse_solver(self.h.h, self._coarse_cb, None), ctx.h)

    # ---- solve-phase API
    @property
    def nlev(self):
        return self.h.nlev + (self.tail.nlev - 1 if self.tail else 0)

    def set_smoother(self, omega, nu1, nu2):
        self.smoother = (omega, nu1, nu2)
        self.h.set_smoother(omega, nu1, nu2)
        if self.tail:
            self.tail.set_smoother(omega, nu1, nu2)
        return self

    def set_kcycle(self, levels):
        """K-cycle (two GCR steps per coarse solve) on the sharded levels 1..levels — inner products summed over the ranks — and, past
        the last sharded level, on the replicated tail's own levels"""
        self.install_allreduce()
        self.h.set_kcycle(levels)
        if self.tail:
            L = len(self.plans) - 1                    # the last sharded level is the tail's level 0
            self.tail.set_kcycle(max(0, levels - L))
            check(lib().mgs_hier_set_kcycle_entry(self.tail.h, 1 if (L >= 1 and levels >= L) else 0), self.ctx.h)
        return self

    def vcycle(self, b, x, zero_guess=True):
        """b: owned entries; x: n_loc + n_halo entries (halo room behind the owned part)"""
        return self.h.vcycle(b, x, zero_guess)

    def spmv(self, x, y):
        if getattr(self, "native", False):
            check(lib().mgs_hier_native_halo(self.h.h, 0, C.c_void_p(x.ptr)), self.ctx.h)
        else:
            self._exchange(0, x.ptr)
        return self.A.spmv(x, y)

    def close(self):
        """drop the native plans and the library's communicator (the hierarchy itself is freed with the object)"""
        if getattr(self, "_ncomm", None) is not None and self.ctx.h:
            self.ctx.sync()
            self._drop_native()

    def install_allreduce(self):
        self.ctx.set_allreduce(lambda a: self.comm.allreduce_host(a))
        if getattr(self, "native", False):
            check(lib().mgs_ctx_set_native_allreduce(self.ctx.h, self._ncomm), self.ctx.h)

    def bicgstab(self, x, b, max_iter=1000, tol=1e-10):
        self.install_allreduce()
        return core.bicgstab(self.A, x, b, self.h, max_iter, tol)


# ------------------------------------------------------------------ bench leg for N > 1
def bench_sharded(args, rank, world, local_rank, log, spmv_bytes, emit_json=None, cpu_baseline=None, pmc_traffic=None):
    """strong scaling: the args.grid^3 problem split by plane ranges over `world` GPUs.  Runs as a rank worker of
    multigridsolver_amd/launch.py (watchdog heartbeats, generation = which transport this attempt uses)."""
    import json
    import time

    import torch
    import torch.distributed as dist

    from . import Context, OP_SPMV, launch
    wd = launch.Watchdog() if os.environ.get("MGS_BENCH_WORKER") == "1" else None
    beat = (lambda ph: wd.beat(ph)) if wd else (lambda ph: None)
    beat("init_process_group")
    if not dist.is_initialized():
        # MGS_DIST_BACKEND=gloo lets several ranks share one GPU (rehearsal on a 1-GPU box); the
        # real multi-GPU run uses nccl (= RCCL over xGMI)
        launch.init_process_group(os.environ.get("MGS_DIST_BACKEND", "nccl" if torch.cuda.is_available() else "gloo"))
    if os.environ.get("MGS_DIST_SHARE_GPU"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    ctx = Context(local_rank, stream.cuda_stream)
    comm = Comm()
    N = args.grid
    lo, hi = plane_range(N, world, rank)
    beat("operator")
    A = ctx.poisson3d(N, lo, hi, local_cols=True)
    n_loc, n_ext = A.shape
    t0 = time.perf_counter()
    sh = ShardedHierarchy(ctx, A, poisson_plane_plan(N, world, rank), args.omega, args.nu1, args.nu2, comm)
    beat("hierarchy + transport")
    sh.build(args.ktg, args.npass, args.tou, coarse_rows=args.coarse_rows, log=log if rank == 0 else None)
    ctx.sync(); comm.barrier()
    t_setup = time.perf_counter() - t0
    b = ctx.vec(n_loc).rand(seed=0, offset=lo * N * N)
    x = ctx.vec(n_ext)
    # local fine-level SpMV kernel rate (HIP events on the kernel's stream), and with the exchange
    xs = ctx.vec(n_ext).rand(seed=1, offset=lo * N * N); y = ctx.vec(n_loc)
    beat("kernel timing")
    ctx.set_option("rowcode", 0)                      # plain CSR kernel beside the shipped pattern-coded one
    A.time_kernel(OP_SPMV, xs, out=y, reps=3)
    ms_k_csr = A.time_kernel(OP_SPMV, xs, out=y, reps=args.kernel_reps)
    ctx.set_option("rowcode", 1)
    A.optimize()
    A.time_kernel(OP_SPMV, xs, out=y, reps=3)
    ms_k = A.time_kernel(OP_SPMV, xs, out=y, reps=args.kernel_reps)
    for _ in range(3):
        sh.spmv(xs, y)
    ctx.sync(); torch.cuda.synchronize(); comm.barrier()
    t0 = time.perf_counter()
    for _ in range(args.kernel_reps):
        sh.spmv(xs, y)
    ctx.sync(); torch.cuda.synchronize()
    ms_x = (time.perf_counter() - t0) / args.kernel_reps * 1e3
    beat("warm-up cycles")
    for _ in range(max(args.warmup, 3)):              # ≥ 3: the native cycle is captured in a hipGraph after two eager runs
        sh.vcycle(b, x)
    ctx.sync(); torch.cuda.synchronize(); comm.barrier()
    beat("timed cycles")
    ex0 = sh.n_exchanges
    t0 = time.perf_counter()
    for _ in range(args.steps):
        sh.vcycle(b, x)
    ctx.sync(); torch.cuda.synchronize(); comm.barrier()
    ex_per_cycle = (sh.n_exchanges - ex0) / max(args.steps, 1)
    el = np.array([time.perf_counter() - t0, ms_k, ms_x])
    comm.allreduce_host(el, op="max")
    elapsed, ms_k, ms_x = float(el[0]), float(el[1]), float(el[2])
    beat("solve check")
    st, it, tol = sh.bicgstab(ctx.vec(n_ext), b, 300, 1e-10)
    # peer-to-peer transport: its device-side waits are bounded (MGS_P2P_TIMEOUT_S); a wait that timed out left an error word — this
    # generation's numbers are void, the worker exits non-zero and the launcher moves on
    p2p_info = None
    if getattr(sh, "native_transport", None) == "p2p" and getattr(sh, "_ncomm", None) is not None:
        inf = (C.c_longlong * 6)()
        check(lib().mgs_comm_p2p_info(sh._ncomm, inf), ctx.h)
        p2p_info = {"window_memory": {1: "uncached", 2: "fine-grained", 3: "coarse-grained"}.get(int(inf[0]), "?"), "window_bytes": int(inf[1]),
                    "slot_doubles": int(inf[2]), "exchange_kernels_launched": int(inf[3]), "error_word": int(inf[4])}
        bad = np.array([float(inf[4] != 0)]); comm.allreduce_host(bad, op="max")
        if bad[0] > 0:
            raise RuntimeError(f"peer-to-peer transport: a wait timed out on some rank (this rank's error word {int(inf[4])})")
    if st != 0:
        raise RuntimeError(f"sharded BiCGSTAB + V-cycle did not converge on the {getattr(sh, 'native_transport', None) or 'callback'} transport: status {st}, {it} iterations, tol {tol:.2e}")
    out = None
    if rank == 0:
        n, nnz = N ** 3, 7 * N ** 3 - 6 * N * N
        loc_bytes = spmv_bytes(n_loc, A.nnz)
        g = loc_bytes / (ms_k * 1e-3) / 1e9
        code = A.rowcode_info()
        streamed = (8 * A.nnz + 17 * n_loc + n_loc // 16 + 4 * code["table_ints"] + 8 * (code["blocks"] + 1)) if code["coded_blocks"] == code["blocks"] else None
        # HBM traffic of the shard's launch: the committed PMC measurement of the full-grid launch (same kernel, same bytes
        # per row) scaled by this shard's share of the rows — counters cannot be read inside a multi-process run
        tr = pmc_traffic("spmv", N) if pmc_traffic else None
        traffic = tr[0] * n_loc / n if tr else None
        ncomm = getattr(sh, "_ncomm", None)
        cw, cr = C.c_int(-1), C.c_int(-1)
        if ncomm is not None:
            lib().mgs_comm_size(ncomm, C.byref(cw), C.byref(cr))
        graph_info = sh.h.graph_info() if hasattr(sh.h, "graph_info") else None
        out = {"metric": "V-cycles/sec + fine-level SpMV HBM GB/s, 512³ 7-pt Poisson, 1/2/4/8 GPU",
               "value": args.steps / elapsed, "unit": "V-cycles/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
               "dtype": "f64", "data": "synthetic",
               "config": {"workload": f"poisson3d_{N}^3_7pt (BASELINE.json configs[4]) row-sharded by plane ranges; V({args.nu1},{args.nu2}) "
                                      f"damped-Jacobi cycle, omega={args.omega}, device-built hierarchy ktg={args.ktg} npass={args.npass} tou={args.tou}",
                          "grid": N, "rows": n, "nnz": nnz, "parallelism": f"row-shard x{world} (" + ({"p2p": "peer-to-peer exchange kernels over IPC-mapped device windows inside the C++ cycle", "rccl": "native RCCL send/recv inside the C++ cycle"}.get(sh.native_transport, "?") if sh.native else "torch.distributed all_to_all callbacks") + ", replicated coarse tail)",
                          "sharded_levels": len(sh.plans), "total_levels": sh.nlev, "setup_seconds": t_setup},
               "transport": {"backend": dist.get_backend(), "world": world, "rank": rank, "native": sh.native_transport if sh.native else None,
                             "native_rccl": bool(sh.native and sh.native_transport == "rccl"), "p2p": p2p_info,
                             "mgs_comm_size": [cw.value, cr.value] if ncomm is not None else None,
                             "cycle_graph": graph_info, "generation": os.environ.get("MGS_BENCH_GEN_NAME"),
                             "exchanges_per_cycle_callbacks": None if sh.native else ex_per_cycle},
               # a run that completed on a later generation of the launcher (slower transport) says so at top level
               "degraded": bool(launch.abandoned_generations()), "abandoned_generations": launch.abandoned_generations(),
               "multi_gpu_note": "N > 1 has NOT run on real links before this line unless a scaling run of the driver exists (DESIGN.md §7): the native transports "
                                 "(peer-to-peer windows, RCCL groups) were verified with several ranks sharing ONE GPU; this run cross-checked every level's exchange "
                                 "against torch.distributed bit for bit at setup and its solve converged; transport.generation names what this line was measured on",
               # whole-job rates of the sharded fine-level SpMV including its halo exchange: bytes that cross HBM (PMC traffic of the
               # full-grid launch) resp. the §8d-d3 CSR byte count, over the slowest rank's time
               "spmv_hbm_gbps": (tr[0] if tr else (streamed or loc_bytes) * n / n_loc) / (ms_x * 1e-3) / 1e9,
               "spmv_effective_csr_gbps": spmv_bytes(n, nnz) / (ms_x * 1e-3) / 1e9,
               # roofline of the dominant kernel on rank 0's shard: bytes that cross HBM / launch time / 8 TB/s (PMC traffic of the committed
               # full-grid profile scaled by this shard's rows — counters cannot be read inside a multi-process run; else the bytes the
               # kernel streams by construction).  effective_csr_* = the SURVEY §8d-d3 CSR byte count over the same time.
               "roofline": {"bound": "hbm", "achieved": (traffic or streamed or loc_bytes) / (ms_k * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                            "frac": (traffic or streamed or loc_bytes) / (ms_k * 1e-3) / 1e9 / 8000.0, "traffic": traffic,
                            "traffic_source": (tr[1] + f" (full-grid launch, scaled by this shard's rows {n_loc}/{n})") if tr else None,
                            "kernel": "csr_rowblock_coded_kernel<SPMV> (rank 0 shard, per-GPU rate; CSR SpMV with pattern-coded column index)",
                            "ms_per_launch": ms_k, "ms_spmv_with_halo_exchange": ms_x,
                            "effective_csr_bytes_per_launch": loc_bytes, "effective_csr_gbps": g, "effective_csr_frac": g / 8000.0,
                            "streamed_bytes_per_launch": streamed, "streamed_gbps": streamed / (ms_k * 1e-3) / 1e9 if streamed else None,
                            "note": "achieved/frac = bytes that cross HBM in one launch / launch time / 8 TB/s; effective_csr_* = SURVEY §8d-d3 CSR bytes of the "
                                    "shard (12·nnz + 20·n + 4) / time — the coded kernel streams 8 B per entry + 1 B per row (DESIGN.md §4), so that figure "
                                    "counts bytes that never move; csr_kernel = plain 12 B/entry CSR kernel on the same shard",
                            "csr_kernel": {"ms": ms_k_csr, "gbps": loc_bytes / (ms_k_csr * 1e-3) / 1e9},
                            "fused_pass_form": sh.h.fused_info(0)},
               "solve_check": {"bicgstab_status": st, "bicgstab_iterations": it, "bicgstab_tol": tol},
               "cpu_baseline": None}
    beat("teardown")
    comm.barrier()
    if rank == 0:
        launch.persist_result(json.dumps(out))   # the finished measurement, before anything is torn down: the supervisor prints it if this worker dies
    launch.mark_done()               # this rank's part is complete: a crash while tearing RCCL down must not restart the generation
    try:
        sh.close()
        del sh, b, x, xs, y, A
        ctx.close()
        dist.destroy_process_group()
    except Exception as e:  # noqa: BLE001
        log("teardown:", repr(e))
    if rank == 0:
        # CPU baseline on rank 0's host cores, after the other ranks are done (they do not wait for it)
        if cpu_baseline is not None:
            beat("cpu baseline")
            wd_limit = getattr(wd, "limit", None)
            if wd is not None:
                wd.limit = max(wd.limit, 900.0)          # a full-size oracle sample: minutes of CPU work without a heartbeat
            try:
                out["cpu_baseline"] = cpu_baseline()
            except Exception as e:  # noqa: BLE001
                log("cpu_baseline failed:", repr(e))
            if wd is not None:
                wd.limit = wd_limit
        (emit_json or (lambda o: print(json.dumps(o), flush=True)))(out)
        launch.mark_printed()
    if wd:
        wd.stop()
