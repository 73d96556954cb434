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


def _fake_rccl():
    """builds tests/sim/fake_rccl/libfake_rccl.so (recording stand-in for librccl.so, transfers through /dev/shm files)"""
    import subprocess
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'sim', 'fake_rccl')
    so, src = os.path.join(d, 'libfake_rccl.so'), os.path.join(d, 'fake_rccl.c')
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(['gcc', '-O1', '-shared', '-fPIC', '-w', '-o', so, src])
    return so


def _bench_two_ranks(sim_lib, workload, B, steps, rccl_lib, log=None, extra=()):
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT', 'MASTER_ADDR')}
    if log is not None:
        env['CPG_FAKE_RCCL_LOG'] = str(log)
    p = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', str(steps), '--warmup', '1',
                        '--workload', workload, '--batch', str(B), '--lib', sim_lib, '--rccl-lib', rccl_lib, '--generic', *extra],
                       capture_output=True, text=True, env=env, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1                                   # rank 0 prints ONE JSON line
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['config']['instances_per_gpu'] == B and out['scaling'] == 'weak'
    assert out['config']['solved'] == 2 * B                  # the root saw both shards' statuses through the gather
    assert out['value'] > 0 and abs(out['ms_per_step'] * steps * out['value'] / 1e3 - 2 * B * steps) < 1e-6 * B
    # first-contact insurance for the 8-GPU run (round-5 review, item 6): the line says what travelled, how the ranks' clocks
    # compare, and that the root checked every rank's rows after the last step
    cfg = out['config']
    assert cfg['ranks_delivered'] == 2
    assert cfg['gather_bytes_per_step'] > 0 and cfg['gather_bytes_per_step'] % B == 0
    rm = cfg['rank_ms_per_step']
    assert len(rm['per_rank']) == 2 and 0 < rm['min'] <= rm['max'] <= out['ms_per_step'] * 1.5 + 1.0
    return out, p.stderr


@pytest.mark.parametrize('workload,B,n_user', [('mpc6', 6, (96, 96)), ('portfolio', 2, (210, 112))])
def test_bench_gpus_2_starts_two_ranks_and_gathers_over_one_rccl(sim_lib, tmp_path, workload, B, n_user):
    """`python bench.py --gpus 2` (no launcher around it, as the driver calls --gpus 1) must start two ranks by
    itself, report n_gpus 2, and move every rank's results to the root through ONE communicator: checked on the
    emulator library with the recording RCCL stand-in (send / receive schedule of RcclGather.enqueue) -- for the
    headline workload's family and for config 3's (per-instance factor kernel in front of the gather)."""
    log = tmp_path / 'rccl.log'
    steps = 2 if workload == 'mpc6' else 1
    out, _ = _bench_two_ranks(sim_lib, workload, B, steps, _fake_rccl(), log)
    assert out['config']['gather'].startswith('rccl')
    txt = log.read_text().splitlines()
    assert sum(ln.startswith('getuid') for ln in txt) == 1 and sum(ln.startswith('init') for ln in txt) == 2   # one id, one communicator per rank
    # per step one group per rank: rank 1 sends its 7 result arrays to rank 0, rank 0 posts the 7 matching receives
    row_bytes = sorted([8 * n_user[0], 8 * n_user[1], 8, 4, 4, 8, 8])      # prim, dual (user entries), obj, iter, status, pri, dua
    sends = [ln for ln in txt if ln.startswith('send')]
    recvs = [ln for ln in txt if ln.startswith('recv')]
    assert len(sends) == len(recvs) == 7 * (steps + 1)
    assert all('rank=1 peer=0' in ln for ln in sends) and all('rank=0 peer=1' in ln for ln in recvs)
    first = sorted(int(ln.rsplit('bytes=', 1)[1]) for ln in sends[:7])
    assert first == [B * rb for rb in row_bytes]
    assert sorted(int(ln.rsplit('bytes=', 1)[1]) for ln in recvs[:7]) == first
    assert out['config']['gather_bytes_per_step'] == sum(first)           # one non-root rank's rows


def test_bench_gpus_2_falls_back_to_the_host_gather_when_rccl_cannot_start(sim_lib, tmp_path):
    """no usable RCCL (library missing; communicator creation failing on one rank only): every rank takes the
    shared-memory transport, the JSON line says so under config.gather, the results still reach the root"""
    out, err = _bench_two_ranks(sim_lib, 'mpc6', 4, 1, str(tmp_path / 'no_such_librccl.so'))
    assert out['config']['gather'].startswith('host (rccl init failed: OSError')
    assert 'RCCL gather not available' in err
    # one rank's ncclCommInitRank fails (stand-in: CPG_FAKE_RCCL_FAIL_RANK), the other's succeeds: both must agree on the host transport
    os.environ['CPG_FAKE_RCCL_FAIL_RANK'] = '1'
    try:
        out, err = _bench_two_ranks(sim_lib, 'mpc6', 4, 1, _fake_rccl(), tmp_path / 'r.log')
    finally:
        del os.environ['CPG_FAKE_RCCL_FAIL_RANK']
    assert out['config']['gather'].startswith('host (rccl init failed')


def test_host_gather_ignores_the_segment_of_a_killed_run():
    """a segment AND its ready file left under the same key by a killed run: a rank that attaches to them before the
    root replaced them must deliver again into the root's segment (token echo), not into the orphan"""
    import threading
    from multiprocessing import shared_memory
    from cvxpygen_amd.sharding import HostGather
    key = f'stale{os.getpid()}'
    name = f'cpg_{key}_0'
    S = HostGather.SLOT
    hdr = S * 2 + S
    stale = shared_memory.SharedMemory(name=name, create=True, size=hdr + 6 * 8)
    stale.buf[:hdr] = bytes(hdr)
    stale.buf[2 * S:2 * S + 16] = b'0123456789abcdef'
    with open(f'/dev/shm/{name}.ready', 'wb') as f:
        f.write(b'0123456789abcdef')
    stale.close()
    got = {}

    def rank1():
        got[1] = HostGather(1, 2, key, timeout=30).gather_rows(np.array([4.0, 5.0, 6.0]), 6)
    th = threading.Thread(target=rank1)
    th.start()
    import time
    time.sleep(0.3)                                          # rank 1 sits in the orphan by now
    got[0] = HostGather(0, 2, key, timeout=30).gather_rows(np.array([1.0, 2.0, 3.0]), 6)
    th.join(30)
    assert not th.is_alive() and got[1] is None
    assert got[0].tolist() == [1.0, 2.0, 3.0, 4.0, 5.0, 6.0]
    assert not os.path.exists(f'/dev/shm/{name}.ready') and not os.path.exists(f'/dev/shm/{name}')


def test_bench_single_rank_json_line_carries_the_contract(sim_lib):
    """`python bench.py` at N = 1 (emulator library, tiny batch): ONE JSON line with every field of the driver's contract,
    the `roofline` and `cpu_baseline` objects of the tier, both kernels of the default-mode step under `phases`, and the
    fixed-rho fork beside the headline value"""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT', 'MASTER_ADDR')}
    B, steps = 6, 2
    p = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--steps', str(steps), '--warmup', '1', '--workload', 'mpc6',
                        '--batch', str(B), '--lib', sim_lib, '--generic', '--cpu-seconds', '1'],
                       capture_output=True, text=True, env=env, timeout=1200)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1
    out = json.loads(lines[0])
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
                'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert key in out, key
    assert out['n_gpus'] == 1 and out['steps'] == steps and out['warmup'] == 1 and out['higher_is_better'] is True
    assert out['dtype'] == 'f64' and out['scaling'] == 'weak' and out['vs_baseline'] is None and 'synthetic' in out['data']
    assert 'workload' in out['config'] and 'model' not in out['config'] and out['config']['solved'] == B
    assert abs(out['value'] * out['ms_per_step'] / 1e3 - B) < 1e-6 * B            # value = instances / step time
    r = out['roofline']
    assert r['bound'] == 'hbm' and r['unit'] == 'GB/s' and r['peak'] == 8000.0 and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-12
    assert 'traffic' in r and r['kernel'] and r['kernel_ms'] <= out['ms_per_step'] + 1e-9
    # ONE accounting: the step's two kernels are the launch the roofline object prices
    assert r['kernel'] == 'osqp_shared_kernel + ' + out['phases']['per_instance_factor']['kernel']
    assert r['algorithmic_bytes_per_launch'] == r['algorithmic_bytes_per_instance'] * r['units_per_launch'] and r['units_per_launch'] == B
    last = out['phases']['shared_factor']['ms'] + out['phases']['per_instance_factor']['ms']          # (the split of the LAST step)
    assert 0.3 * last < r['kernel_ms'] < 3.0 * last
    assert abs(r['achieved'] - r['algorithmic_bytes_per_launch'] / (r['kernel_ms'] * 1e-3) / 1e9) < 1e-9 * max(1.0, r['achieved']) and 'frac_step' not in r
    c = out['cpu_baseline']
    assert c['kind'] == 'restatement' and c['kind_contract'] == 'port' and c['value'] > 0 and c['cores'] >= 1 and c['unit'] == out['unit'] and c['sample']
    assert set(out['phases']) == {'shared_factor', 'per_instance_factor'}            # default mode: two kernels per step
    assert out['phases']['shared_factor']['instances'] == B
    assert out['fixed_rho']['value'] > 0
