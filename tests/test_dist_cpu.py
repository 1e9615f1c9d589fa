"""world_size-2 gloo tests (CPU) of the multi-GPU host logic in multigridsolver_amd/dist.py:
plane partition, level-0 halo plan, the per-level setup handshake (remote aggregate ids →
coarse halo slots → coarse send lists) and the halo exchange transport.  Kernels are replaced
by numpy stand-ins here (the product kernels need a GPU); the sharded result is checked against
the same computation done globally with the CPU oracle."""
import os
import sys

import numpy as np
import pytest

from conftest import REPO

N = 8


def _local_poisson(orc, N, lo, hi):
    """numpy mirror of mgs_csr_poisson3d(local_cols=1)"""
    full = orc.poisson3d(N).to_scipy().tocsr()
    n2 = N * N
    sub = full[lo * n2: hi * n2].tocsr(); sub.sort_indices()
    nloc = (hi - lo) * n2
    g = sub.indices.astype(np.int64) - lo * n2
    has_lo = lo > 0
    loc = np.where(g < 0, nloc + g + n2, np.where(g >= nloc, nloc + (n2 if has_lo else 0) + (g - nloc), g))
    return sub.indptr.copy(), loc.astype(np.int32), sub.data.copy(), nloc


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, REPO)
    import scipy.sparse as sps
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from multigridsolver_amd import dist as mgd
    from oracle import oracle_py as orc
    try:
        comm = mgd.Comm(device=torch.device("cpu"))
        lo, hi = mgd.plane_range(N, world, rank)
        plan = mgd.poisson_plane_plan(N, world, rank)
        rp, ci, v, nloc = _local_poisson(orc, N, lo, hi)
        assert plan.n_loc == nloc and plan.n_halo == ci.max() + 1 - nloc
        # --- halo exchange transport: x_ext halo == the owner's entries
        n2 = N * N
        xg = np.arange(N ** 3, dtype=np.float64) * 0.5 + 1.0
        x_ext = torch.zeros(nloc + plan.n_halo, dtype=torch.float64)
        x_ext[:nloc] = torch.from_numpy(xg[lo * n2: hi * n2])
        send = torch.cat([x_ext[torch.from_numpy(ix.astype(np.int64))] for ix in plan.send_idx])
        comm.a2a_f64(x_ext[nloc:], send, plan.recv_counts, plan.send_counts)
        A_loc = sps.csr_matrix((v, ci, rp), shape=(nloc, nloc + plan.n_halo))
        y = A_loc @ x_ext.numpy()
        y_ref = (orc.poisson3d(N).to_scipy() @ xg)[lo * n2: hi * n2]
        assert np.array_equal(y, y_ref)
        # --- handshake: stand-in aggregation = consecutive pairs of owned rows, a few G0 rows
        agg = (np.arange(nloc) // 2).astype(np.int32)
        agg[::7] = -1
        _, agg_c = np.unique(agg[agg >= 0], return_inverse=True)
        agg[agg >= 0] = agg_c
        nc = int(agg.max()) + 1
        halo_cols, n_halo_c, cplan = mgd.coarse_plan_handshake(plan, agg, nc, comm.exchange_lists)
        assert halo_cols.shape == (plan.n_halo,) and cplan.n_loc == nc and cplan.n_halo == n_halo_c
        # local Galerkin with the extended column map (numpy stand-in of mgs_galerkin_shard)
        colmap = np.concatenate([agg, halo_cols])
        coo = A_loc.tocoo()
        I, J = agg[coo.row], colmap[coo.col]
        ok = (I >= 0) & (J >= 0)
        Ac_loc = sps.csr_matrix((coo.data[ok], (I[ok], J[ok])), shape=(nc, nc + n_halo_c)); Ac_loc.sum_duplicates(); Ac_loc.sort_indices()
        # assemble both levels globally and compare with the global Galerkin product
        ncs = comm.allgather_ints(nc); offs_c = np.concatenate([[0], np.cumsum(ncs)])
        nls = comm.allgather_ints(nloc); offs_f = np.concatenate([[0], np.cumsum(nls)])
        mine = mgd.shard_to_global(cplan, Ac_loc.indptr, Ac_loc.indices, Ac_loc.data, offs_c, rank)
        parts = [None] * world
        dist.all_gather_object(parts, (mine.indptr, mine.indices, mine.data, agg))
        Acg = sps.vstack([sps.csr_matrix((d, i, p), shape=(len(p) - 1, int(offs_c[-1]))) for (p, i, d, _) in parts]).tocsr()
        gagg = np.concatenate([np.where(a >= 0, a + offs_c[r], -1) for r, (_, _, _, a) in enumerate(parts)])
        rows = np.nonzero(gagg >= 0)[0]
        Pg = sps.csr_matrix((np.ones(rows.size), (rows, gagg[rows])), shape=(N ** 3, int(offs_c[-1])))
        ref = orc.poisson3d(N).galerkin(orc.Csr.from_scipy(Pg)).to_scipy()
        assert abs(Acg - ref).max() <= 1e-13 and (Acg != 0).nnz <= ref.nnz
        # coarse exchange: the coarse plan delivers exactly the requested remote entries
        xc = np.arange(int(offs_c[-1]), dtype=np.float64) + 0.25
        xc_ext = torch.zeros(nc + n_halo_c, dtype=torch.float64)
        xc_ext[:nc] = torch.from_numpy(xc[offs_c[rank]: offs_c[rank] + nc])
        send = torch.cat([xc_ext[torch.from_numpy(ix.astype(np.int64))] for ix in cplan.send_idx] + [torch.zeros(0, dtype=torch.float64)])
        comm.a2a_f64(xc_ext[nc:], send, cplan.recv_counts, cplan.send_counts)
        exp_halo = np.concatenate([xc[offs_c[p] + cplan.recv_ids[p]] for p in range(world)])
        assert np.array_equal(xc_ext[nc:].numpy(), exp_halo), (xc_ext[nc:].numpy()[:6], exp_halo[:6])
        yc = Ac_loc @ xc_ext.numpy()
        want = (ref @ xc)[offs_c[rank]: offs_c[rank] + nc]
        assert np.allclose(yc, want, rtol=0, atol=1e-12), (np.abs(yc - want).max(), np.nonzero(np.abs(yc - want) > 1e-12)[0][:10], nc, n_halo_c, cplan.recv_counts, cplan.send_counts)
        # reductions
        a = np.array([rank + 1.0, 2.0]); comm.allreduce_host(a)
        assert a.tolist() == [world * (world + 1) / 2.0, 2.0 * world]
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_partition_and_plan_shapes():
    sys.path.insert(0, REPO)
    from multigridsolver_amd import dist as mgd
    for Ng, w in [(512, 8), (512, 3), (10, 4), (6, 2)]:
        rs = [mgd.plane_range(Ng, w, r) for r in range(w)]
        assert rs[0][0] == 0 and rs[-1][1] == Ng and all(rs[i][1] == rs[i + 1][0] for i in range(w - 1))
        assert max(h - l for l, h in rs) - min(h - l for l, h in rs) <= 1
        for r in range(w):
            p = mgd.poisson_plane_plan(Ng, w, r)
            assert p.n_loc == (rs[r][1] - rs[r][0]) * Ng * Ng
            assert p.n_halo == Ng * Ng * ((r > 0) + (r < w - 1))
            for q in range(w):
                assert len(p.send_idx[q]) == (Ng * Ng if abs(q - r) == 1 else 0)


@pytest.mark.parametrize("WORLD", [2, 4])
def test_handshake_and_exchange_gloo(WORLD):
    """world 2 (each rank has one neighbour) and world 4 (interior ranks have two, most peer lists empty)"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + WORLD
    procs = [ctx.Process(target=_worker, args=(r, WORLD, port, q)) for r in range(WORLD)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(WORLD)]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in res:
        assert msg == "ok", f"rank {rank}: {msg}"


def test_export_zones_and_range_expansion_of_coarse_halo():
    """Pure host logic of the pack-free exchanges (multigridsolver_amd/dist.py): zones of the exported rows (one per peer and contiguous
    run of its list, 0 = interior), and coarse_plan_handshake's cluster-wise filling of nearly contiguous id lists — a scattered
    prefix + a far block stay two ranges, a cluster is filled only within the slack, −1 (G0) rows get no slot."""
    sys.path.insert(0, REPO)
    from multigridsolver_amd import dist as mgd
    plan = mgd.LevelPlan(100, [np.array([90, 91, 92, 0, 1, 2, 3], np.int32), np.zeros(0, np.int32)], [np.zeros(0, np.int32)] * 2)
    z = mgd.export_zones(plan)
    assert z[:4].tolist() == [2, 2, 2, 2] and z[90:93].tolist() == [1, 1, 1] and z[4:90].sum() == 0 and z[93:].sum() == 0
    # one "peer" = this rank itself (the one-GPU rehearsal): its list is [last rows, first rows]; the aggregates of the first rows
    # are ids 0..3, those of the last rows 40, 42, 43, 45 plus one far straggler 7 → clusters {0..3, 7} and {40..45}
    plan = mgd.LevelPlan(16, [np.arange(16, dtype=np.int32)], [np.arange(16, dtype=np.int32)])
    agg = np.array([40, 42, 43, 45, 40, 42, -1, 45, 0, 1, 2, 3, 0, 7, 2, 3], dtype=np.int32)
    ex = lambda lists: [np.asarray(a, dtype=np.int64).copy() for a in lists]        # world 1: the "peer" gets its own lists back
    cols, n_halo_c, cplan = mgd.coarse_plan_handshake(plan, agg, 50, ex, span_slack=2.5)
    u = cplan.recv_ids[0].tolist()
    assert u == [0, 1, 2, 3, 4, 5, 6, 7, 40, 41, 42, 43, 44, 45], u              # {0..7}: 8 slots for 5 ids (≤ 2.5×); {40..45}: 6 for 4
    assert n_halo_c == len(u) and cols[6] == -1 and cols[0] == 50 + u.index(40) and cols[13] == 50 + u.index(7)
    assert cplan.send_idx[0].tolist() == u
    cols, n_halo_c, cplan = mgd.coarse_plan_handshake(plan, agg, 50, ex, span_slack=1.2)
    assert cplan.recv_ids[0].tolist() == [0, 1, 2, 3, 7, 40, 42, 43, 45]                 # nothing within 1.2×: the lists stay as they are
    cols, n_halo_c, cplan = mgd.coarse_plan_handshake(plan, agg, 50, ex, span_slack=1.0)
    assert cplan.recv_ids[0].tolist() == [0, 1, 2, 3, 7, 40, 42, 43, 45]
