"""Multi-rank GPU test of the sharded path on a single-GPU box: 2 (and 3) ranks share GPU 0,
backend gloo with host-staged halo buffers.  Everything except the RCCL transport is the code
the 8-GPU run executes: plane shards with local column numbering, mgs_aggregate_shard, the
handshake, mgs_galerkin_shard, the C++ V-cycle with halo/coarse-tail callbacks, all-reduced dots."""
import os
import subprocess
import sys

import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world,N,tail,overlap,fused", [(2, 20, 1500, 1, 1), (3, 18, 800, 1, 1), (2, 40, 3000, 1, 1), (2, 20, 1500, 0, 1),
                                                        (2, 20, 1500, 1, 0), (3, 18, 800, 0, 0), (4, 24, 1200, 1, 1)])
def test_sharded_vcycle_matches_oracle(world, N, tail, overlap, fused):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    port = 29600 + (os.getpid() % 1000) + world
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(REPO, "tests", "dist_gpu_worker.py"), str(N), str(tail), str(overlap), str(fused)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=REPO)
    assert r.returncode == 0 and "DIST_OK" in r.stdout, r.stdout[-3000:] + r.stderr[-6000:]


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_general_operator_matches_oracle(world, inputs):
    """a bundled, nonsymmetric operator (CSky3d30) sharded by contiguous row ranges with the generic halo plan
    (shard_from_global): sharded cycle vs the oracle on the globally assembled hierarchy, sharded solve to 1e-10"""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    port = 29700 + (os.getpid() % 1000) + world
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(REPO, "tests", "dist_gpu_worker.py"), "0", "3000", "1", "1", inputs["CSky3d30"]]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=REPO)
    assert r.returncode == 0 and "DIST_OK" in r.stdout, r.stdout[-3000:] + r.stderr[-6000:]
