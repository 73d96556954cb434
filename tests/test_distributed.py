"""N > 1 path on CPU: two processes (gloo process group for the launcher-level barrier, as bench.py uses
torch.distributed), each solving its contiguous shard with the emulator build, then the final gather to
the root through cvxpygen_amd.sharding.HostGather -- the only exchange of the design."""
import os
import socket
import sys

import numpy as np
import pytest


def _worker(rank, world, port, sim_lib, q, conic=False):
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from cvxpygen_amd import families
    from cvxpygen_amd.runtime import BatchSolver
    from cvxpygen_amd.conic_runtime import ConicBatchSolver
    from cvxpygen_amd.sharding import HostGather, solve_sharded
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    rng = np.random.default_rng(0)
    tv = rng.standard_normal((5, 3))
    if conic:                       # the same family through the interior-point path
        bs = ConicBatchSolver(families.nonneg_ls(solver='CLARABEL'), lib_path=sim_lib)
    else:
        bs = BatchSolver(families.nonneg_ls(), lib_path=sim_lib)
        bs.set_launch(1, 1, 0)
    bs.set_updated(['b'])
    g = HostGather(rank, world, key=f't{port}')
    out = solve_sharded(bs, tv, rank, world, g)
    assert (out is None) == (rank != 0)
    if rank == 0:
        q.put({k: v for k, v in out.items()})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('conic', [False, True])
def test_two_rank_shard_and_gather(sim_lib, oracle_lib, conic):
    import torch.multiprocessing as mp
    from cvxpygen_amd import families
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, sim_lib, q, conic)) for r in range(2)]
    for p in procs:
        p.start()
    out = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rng = np.random.default_rng(0)
    tv = rng.standard_normal((5, 3))
    if conic:
        from oracle import clarabel_numpy as cl
        d = families.nonneg_ls(solver='CLARABEL')
        th = np.tile(d.theta0, (5, 1)); th[:, 3:6] = tv
        o = cl.cpg_solve_batch(d, th)
        assert out['iter'].tolist() == o['iter'].tolist()
        assert np.allclose(out['prim'], o['sol_x'][:, d.variables[0].indices], atol=1e-9)
        return
    d = families.nonneg_ls()
    th = np.tile(d.theta0, (5, 1)); th[:, 3:6] = tv
    o = oracle_lib.cpg_solve_batch(d, th, ['b'])
    assert out['iter'].tolist() == o['iter'].tolist()
    assert np.allclose(out['prim'], o['sol_x'][:, d.variables[0].indices], atol=1e-10)
    assert out['prim'].shape == (5, 2) and out['status'].shape == (5,)


def test_shard_bounds_cover_the_batch():
    from cvxpygen_amd.sharding import shard_bounds
    for B in (0, 1, 7, 8, 100000, 1000003):
        for world in (1, 2, 3, 8):
            cuts = [shard_bounds(B, r, world) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == B
            assert all(cuts[r][1] == cuts[r + 1][0] for r in range(world - 1))
            sizes = [hi - lo for lo, hi in cuts]
            assert max(sizes) - min(sizes) <= 1


def test_package_does_not_import_torch():
    """north_star: host code calls HIP through ctypes, no PyTorch -- the package (incl. the multi-GPU gather) is torch-free"""
    pkg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'cvxpygen_amd')
    for dp, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith('.py'):
                txt = open(os.path.join(dp, fn)).read()
                assert 'import torch' not in txt and 'from torch' not in txt, fn
