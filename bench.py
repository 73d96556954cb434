#!/usr/bin/env python
"""
bench.py -- QP instances solved per second by the MI355X batched-solve backend on the workload
BASELINE.json's metric is quoted on: config 2, the MPC QP of examples/MPC.ipynb at
n_x = 12, n_u = 4, horizon 10, OSQP backend, 100 000 instances per GPU, x_init varying per
instance (tests/test_E2E_QP.py:146: x_init = -2 + 4 * rand), cold start, default OSQP settings
(cvxpygen/solvers/osqp.py:102-115) on top of the OSQP >= 1.0 library defaults that every cpg_solve of the
reference restores (osqp_set_default_settings, osqp.py:100-101): rho adapted every 50 iterations, duality-gap
test.  `value` is measured in THAT mode (two kernels per step: shared factor until an instance's rho changes,
per-instance factor behind it); the fixed-rho fork of the same workload is printed beside it as `fixed_rho`.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" = one pass of the hot path over one batch (theta already resident in HBM): canonicalise ->
update -> ADMM solve -> retrieve, for every instance of the rank's shard.  Weak scaling: every rank
solves its own 100 000 instances; no data-path collective (instances are independent); rank 0
prints ONE JSON line with the whole-job throughput, the roofline object for the solve kernel and
the CPU baseline (the scalar-C oracle restatement, timed on the host cores of this box).
"""

import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from cvxpygen_amd import families                      # noqa: E402
from cvxpygen_amd.runtime import BatchSolver, DeviceBatch, BUILD_OPTIONS_FIXED_RHO   # noqa: E402

def C_float():
    return ctypes.c_float(0)


HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8 TB/s peak


def source_fingerprint() -> str:
    """sha256 over the kernel sources and the generators of the family libraries: what a recorded PMC
    measurement must have been taken on to be replayed next to a live timing"""
    import hashlib
    h = hashlib.sha256()
    base = os.path.join(ROOT, 'cvxpygen_amd')
    for f in sorted(os.listdir(os.path.join(base, 'csrc'))):
        if f.endswith(('.h', '.cpp')):
            h.update(open(os.path.join(base, 'csrc', f), 'rb').read())
    for f in ('codegen.py', 'solve_program.py', 'refactor_plan.py', 'slot_layout.py'):
        h.update(open(os.path.join(base, f), 'rb').read())
    return h.hexdigest()[:16]


def make_workload(name: str):
    if name == 'mpc12':
        desc = families.mpc(12, 4, 10)
        label = 'MPC QP n_x=12 n_u=4 H=10 (examples/MPC.ipynb shape), OSQP, x_init varies'
    elif name == 'mpc6':
        desc = families.mpc(6, 3, 10)
        label = 'MPC QP n=6 m=3 H=10 (examples/MPC.ipynb), OSQP, x_init varies'
    elif name == 'portfolio':
        desc = families.portfolio(100, 10)
        label = 'portfolio QP n=100 m=10 (examples/portfolio.ipynb), OSQP, a/F/Sig_f_sqrt/d_sqrt/w_prev per instance'
    elif name == 'adp':
        desc = families.adp()
        label = 'ADP SOCP (tests/test_E2E_SOCP.py:15-64, norm form), conic interior point (Clarabel path), f/G per instance'
    else:
        raise ValueError(name)
    return desc, label


def adp_params(B: int, seed: int):
    """per-instance values of tests/test_E2E_SOCP.py:38-62: state = -2 + 4 rand(6), f = A(state) state,
    G = B(state); Rsqrt stays sqrt(0.1) I"""
    st = -2.0 + 4.0 * np.random.default_rng(seed).random((B, 6))
    f = np.empty((B, 6)); G = np.zeros((B, 6, 3))
    f[:, :3] = st[:, :3] + 0.1 * st[:, 3:]
    f[:, 3:] = st[:, 3:] * (1.0 - 0.1 * st[:, 3:])
    for k in range(3):
        G[:, 3 + k, k] = 0.1 * st[:, 3 + k]
    return {'f': f, 'G': G}


BINDING = {
    # (workload, all parameters per instance, fixed-rho fork) -> what binds the dominant kernel, from the PMC passes under profiles/
    # (busy fractions: counter x 4 / (1024 SIMDs x kernel cycles) for the quad-cycle counters, / (256 CUs x cycles) for the LDS ones)
    ('mpc12', False, False): 'LDS throughput, then latency: the shared-factor kernel (11.8 of the 21.9 ms) keeps the LDS pipe 80 % busy, 18 % of '
                             'it bank conflicts, VALU 50 %; the per-instance factor kernel behind it (10.1 ms) keeps no unit busy -- VALU 47 %, '
                             'LDS pipe 54 %, 51 % of the wave cycles waiting at two wavefronts per SIMD and 256 VGPRs (profiles/r6_final4_pmc_config2.txt)',
    ('mpc12', False, True): 'LDS throughput: SQ_LDS_IDX_ACTIVE 84 % of the CU cycles, 24 % of it bank conflicts; VALU 50 % (profiles/r2_final6_pmc_config2.txt)',
    ('mpc6', False, False): 'as mpc12: an LDS-bound shared-factor kernel in front of a latency-bound per-instance factor kernel',
    ('portfolio', False, False): 'instruction rate of ONE wavefront per SIMD (three of the four SIMDs of a CU): VALU 24 % of all SIMD cycles, LDS pipe '
                                 '25 %, 40 % of the wave cycles waiting; HBM-side traffic 82 GB per launch = 1.66 TB/s = 21 % of the peak '
                                 '(profiles/r6_final4_pmc_config3.txt, r4_s13_probe_resident.txt)',
    ('portfolio', False, True): 'latency (as the default mode; fewer termination tests and no refactorisations)',
    ('mpc12', True, False): 'latency at ONE instance per CU (team kernel, four wavefronts per instance): 17 dependent phases per ADMM iteration (one '
                            'barrier each, 4.3 us per iteration) and a factorisation table walk of 2 472 step slots at the instruction rate of one wavefront (~97 cycles per slot); '
                            'nothing streamed per iteration (DESIGN.md 4.7; the streaming kernel it replaces: 484 phases per iteration each waiting '
                            'for HBM, 342 GB of fetches per launch, profiles/r4_final_pmc_allparams.txt)',
    ('adp', False, False): 'VALU issue: SQ_ACTIVE_INST_VALU 91 % of the SIMD cycles at four wavefronts per SIMD, 41.0 k vector instructions per instance (profiles/r6_final4_pmc_config4.txt)',
}


def team_kernel_in_use(solver) -> bool:
    """does the per-instance factor handle of this solver run the team kernel (one workgroup of W wavefronts per instance)?"""
    import ctypes as _C
    h = getattr(solver, 'h_ref', None)
    if h is None or not h.value:
        return False
    v = _C.c_double(0)
    solver.lib.L.cpg_hip_get_setting(h, b'team_executor', _C.byref(v))
    return v.value > 0.0               # (the team's width W)


def squad_kernel_in_use(solver) -> bool:
    """does the shared-factor handle of this solver run the squad kernel (the family's solve program in registers, W instances per
    workgroup of W wavefronts: cpg_osqp_squad.h)?"""
    import ctypes as _C
    h = getattr(solver, 'h_shared', None)
    if h is None or not h.value:
        return False
    v = _C.c_double(0)
    solver.lib.L.cpg_hip_get_setting(h, b'squad_executor', _C.byref(v))
    return v.value == 1.0


def resident_kernel_in_use(solver) -> bool:
    """does the per-instance factor handle of this solver run the resident kernel (family library with this family's resident executor)?"""
    import ctypes as _C
    h = getattr(solver, 'h_ref', None)
    if h is None or not h.value:
        return False
    v = _C.c_double(0)
    solver.lib.L.cpg_hip_get_setting(h, b'resident_executor', _C.byref(v))
    return v.value == 1.0


def portfolio_params(desc, B: int, seed: int):
    """per-instance values of examples/portfolio.ipynb cell 7"""
    rng = np.random.default_rng(seed)
    n, m = 100, 10
    sig = np.zeros((B, m, m))
    sig[:, np.arange(m), np.arange(m)] = rng.random((B, m))
    return {'a': rng.standard_normal((B, n)), 'F': np.round(rng.standard_normal((B, n, m))),
            'Sig_f_sqrt': sig, 'd_sqrt': rng.random((B, n)), 'w_prev': np.zeros((B, n))}


def make_theta(desc, B: int, seed: int) -> np.ndarray:
    p = desc.param('x_init')
    rng = np.random.default_rng(seed)
    return -2.0 + 4.0 * rng.random((B, p.size))


def effective_cpus():
    """CPUs this process may really use: the scheduler affinity mask and the cgroup CPU quota (a lease on a 256-thread host
    is often worth a dozen cores: omp_get_max_threads() says 256, the quota decides).  Returns (count, how it was found)."""
    n_aff = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    quota = None
    try:                                   # cgroup v2
        q, per = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if q != 'max':
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:                               # cgroup v1
            q = float(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
            per = float(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    n = n_aff if quota is None else max(1, min(n_aff, int(np.ceil(quota))))
    return n, f'sched_getaffinity {n_aff}' + ('' if quota is None else f', cgroup cpu quota {quota:.1f}')


def _cpu_sweep(run, unit_per_probe: int, target_seconds: float, cap: int, unit: str, what: str):
    """The oracle timed on the box's host cores: single thread first, then a sweep over thread counts (1, 8, 32, the effective
    CPU count, everything omp sees) on a sample sized for ~target_seconds of the best setting; reports the best rate with
    the thread count that produced it, the single-thread rate, and the whole sweep -- `cores` is the count that was USED."""
    from oracle import binding as ob
    omp_max = int(ob.lib().oracle_num_threads())
    eff, how = effective_cpus()
    t1 = run(unit_per_probe, 1, 1)                              # one thread
    rate1 = unit_per_probe / t1
    counts = sorted({c for c in (8, 32, eff, omp_max) if 1 < c <= omp_max})
    budget = max(2.0, 0.6 * target_seconds / max(1, len(counts)))
    sweep = {1: rate1}
    for c in counts:
        Bc = int(max(2 * c, min(cap, budget * rate1 * min(c, eff))))
        sweep[c] = Bc / run(Bc, 2, c)
    best = max(sweep, key=sweep.get)
    Bf = int(max(2 * best, min(cap, 0.4 * target_seconds * sweep[best])))
    tf = run(Bf, 3, best)
    return {'value': Bf / tf, 'unit': unit, 'cores': int(best), 'kind': 'restatement', 'kind_contract': 'port',      # ("restatement" in every report; the bench contract's enum calls it a port)
            'single_thread': rate1, 'effective_cpus': int(eff), 'effective_cpus_from': how, 'omp_max_threads': omp_max,
            'thread_sweep': {str(k): round(v, 1) for k, v in sorted(sweep.items())},
            'sample': f'{Bf} instances of the same workload ({what}), OpenMP static over instances on {best} threads, {tf:.1f} s wall; '
                      f'thread sweep on smaller samples beside it'}


def cpu_baseline(desc, target_seconds: float = 12.0, **mode):
    """Oracle (scalar-C restatement of the generated solver, oracle/osqp_oracle.c) on the host cores;
    bounded sample of the same workload, same mode (mode: oracle settings, e.g. adaptive_rho=0)."""
    from oracle import binding as ob
    ob.build()
    p = desc.param('x_init')

    def run(B, seed, threads):
        th = np.tile(desc.theta0, (B, 1))
        th[:, p.col:p.col + p.size] = make_theta(desc, B, seed)
        t0 = time.time()
        ob.cpg_solve_batch(desc, th, ['x_init'], nthreads=threads, **mode)
        return time.time() - t0
    return _cpu_sweep(run, 64, target_seconds, 200000, 'QP instances/s', 'same settings, cold start')


def cpu_baseline_portfolio(desc, target_seconds: float = 12.0, **mode):
    """config 3 on the host: the C oracle with per-instance osqp_update_data_mat"""
    from oracle import binding as ob
    ob.build()

    def run(B, seed, threads):
        pv = portfolio_params(desc, B, seed)
        th = np.tile(desc.theta0, (B, 1))
        for nm, v in pv.items():
            p = desc.param(nm)
            for k in range(B):
                th[k, p.col:p.col + p.size] = desc.flatten_param(nm, v[k])
        t0 = time.time()
        ob.cpg_solve_batch(desc, th, list(pv.keys()), nthreads=threads, **mode)
        return time.time() - t0
    return _cpu_sweep(run, 8, target_seconds, 20000, 'QP instances/s', 'per-instance osqp_update_data_mat')


def cpu_baseline_adp(desc, target_seconds: float = 10.0):
    """config 4 on the host: the interior-point restatement in C (oracle/clarabel_oracle.c, the algorithm of
    oracle/clarabel_numpy.py statement for statement), OpenMP over the instances -- a new solver per instance as the
    reference's generated code builds one (solvers/clarabel.py:201-204)"""
    from oracle import binding as ob
    ob.build()

    def run(B, seed, threads):
        pv = adp_params(B, seed)
        th = np.stack([desc.theta_from_values({k: v[i] for k, v in pv.items()}) for i in range(B)])
        t0 = time.time()
        r = ob.clarabel_solve_batch(desc, th, nthreads=threads)
        dt = time.time() - t0
        if not (r['status'] == 1).all():
            raise RuntimeError('cpu baseline: the conic oracle did not solve its sample')
        return dt
    out = _cpu_sweep(run, 256, target_seconds, 400000, 'SOCP instances/s', 'a new solver per instance, default settings')
    out['sample'] = out['sample'].replace('OpenMP static', 'C restatement of the interior-point method, OpenMP dynamic')
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch', type=int, default=100000, help='instances per GPU')
    ap.add_argument('--workload', default='mpc12')
    ap.add_argument('--lib', default=None, help='alternative libcpg_hip build (experiments)')
    ap.add_argument('--generic', action='store_true', help='table-driven kernels instead of the family-specialised library')
    ap.add_argument('--waves', type=int, default=0)
    ap.add_argument('--ipw', type=int, default=0, help='instances per wave')
    ap.add_argument('--blocks-per-cu', type=int, default=0)
    ap.add_argument('--placement', type=int, default=-1, help='solve program: -1 auto, 0 L2/HBM stream, 1 LDS resident')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-seconds', type=float, default=12.0)
    ap.add_argument('--check', action='store_true', help='compare a sample with the oracle')
    ap.add_argument('--max-iter', type=int, default=0, help='experiments: cap the iteration count')
    ap.add_argument('--check-termination', type=int, default=0, help='experiments: termination check interval (reference default 25)')
    ap.add_argument('--eps', type=float, default=0.0, help='eps_abs = eps_rel (tight run of SURVEY.md 8(d): 1e-6)')
    ap.add_argument('--adjoint', action='store_true', help='config 5: also time the batched QP adjoint (gradient=True path)')
    ap.add_argument('--osqp1', action='store_true', help='(default now; kept for old command lines) OSQP >= 1.0 library defaults')
    ap.add_argument('--fixed-rho', action='store_true', help='the other fork: a solver that never adapts rho, no duality-gap test (OSQP 0.6-style codegen)')
    ap.add_argument('--no-fixed-rho-leg', action='store_true', help='skip the short fixed-rho measurement printed beside the default mode')
    ap.add_argument('--no-gather', action='store_true', help='multi-GPU: leave the final gather out of the timed steps')
    ap.add_argument('--no-wall', action='store_true', help='skip the PCIe-inclusive pipelined measurement')
    ap.add_argument('--all-params', action='store_true', help='every parameter varies per instance (matrix parameters: per-instance refactorisation path)')
    ap.add_argument('--instance-executor', choices=['auto', 'stream', 'generated'], default='auto',
                    help='experiments: executor of the per-instance factor kernel behind the shared-factor one (family libraries)')
    ap.add_argument('--debug-stage', type=int, default=0, help='experiments: the per-instance factor kernel stops after stage k of its set-up (results are garbage)')
    ap.add_argument('--rccl-timeout', type=float, default=60.0, help='seconds ncclCommInitRank may take before the gather falls back to the host transport')
    ap.add_argument('--rccl-lib', default='librccl.so', help='RCCL library of the result gather (CPU tier: the recording stand-in of tests/sim/fake_rccl)')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # started as plain `python bench.py --gpus N`: become the launcher -- one rank per GPU through the same
        # torch.distributed.run command line the driver uses; rank 0's JSON line is this process's output
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0)); port = sk.getsockname()[1]
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
               '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
        sys.exit(subprocess.call(cmd, env=env))

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    dist = None
    if world > 1:
        # torch.distributed only as the launcher-level plumbing the driver's contract asks for (rendezvous,
        # barrier, max over ranks of the elapsed time) -- on its CPU backend, so that the process holds exactly ONE
        # RCCL runtime and ONE communicator: the one of the data path, the final gather of the results
        # (cvxpygen_amd.sharding.RcclGather: librccl through ctypes on the solver's own stream)
        import torch
        import torch.distributed as dist
        os.environ.setdefault('GLOO_SOCKET_IFNAME', 'lo')
        if torch.cuda.is_available():
            torch.cuda.set_device(local_rank)
        dist.init_process_group('gloo')

    desc, label = make_workload(args.workload)
    lib_path = args.lib
    plan = None
    gen = os.path.join(ROOT, 'cvxpygen_amd', 'generated', args.workload, f'libcpg_{args.workload}.so')
    if lib_path is None and not args.generic and os.path.exists(gen) and desc.solver == 'OSQP':
        # what generate_code() builds: executor specialised for this family.  The library is tied to the family's solve
        # program by a fingerprint; build_family_library is a no-op when the prebuilt one matches the plan and
        # recompiles it (hipcc, ~1 min) when it does not -- never a stale library, never a silent fallback
        from cvxpygen_amd import codegen
        from cvxpygen_amd.runtime import build_family_plan
        plan = build_family_plan(desc)
        lib_path = codegen.build_family_library(plan, os.path.dirname(gen), args.workload) if rank == 0 else gen
        if dist is not None:
            dist.barrier()              # the other ranks load what rank 0 has verified / rebuilt
    # default: OSQP >= 1.0 library defaults (rho adaptation every 50 iterations, tolerance 5, duality-gap test)
    build_options = dict(BUILD_OPTIONS_FIXED_RHO) if args.fixed_rho else {}
    args.osqp1 = not args.fixed_rho
    oracle_mode = dict(adaptive_rho=0, check_dualgap=0) if args.fixed_rho else {}
    if desc.solver == 'CLARABEL':
        from cvxpygen_amd.conic_runtime import ConicBatchSolver
        cplan = None
        if lib_path is None and not args.generic:
            # what generate_code(solver='CLARABEL') builds: substitution-program executor generated for this family
            # (verified against the plan's fingerprint / rebuilt, as above)
            from cvxpygen_amd import codegen
            from cvxpygen_amd.conic_plan import build_conic_plan
            cplan = build_conic_plan(desc)
            lib_path = codegen.build_conic_library(cplan, os.path.dirname(gen), args.workload) if rank == 0 else gen
            if dist is not None:
                dist.barrier()
        solver = ConicBatchSolver(desc, device=local_rank, lib_path=lib_path, plan=cplan)
    else:
        solver = BatchSolver(desc, device=local_rank, lib_path=lib_path, build_options=build_options, plan=plan)
    solver.set_launch(args.waves, args.ipw, args.blocks_per_cu)
    solver.set_program_placement(args.placement)
    B = args.batch
    if args.workload in ('portfolio', 'adp'):
        pv = portfolio_params(desc, B, 1000 + rank) if args.workload == 'portfolio' else adp_params(B, 1000 + rank)
        solver.set_updated(list(pv.keys()))
        theta = solver.theta_var(pv)
    elif args.all_params:
        solver.set_updated(None)
        rng = np.random.default_rng(1000 + rank)
        full = np.tile(desc.theta0[:-1], (B, 1)) * (1 + 0.05 * rng.standard_normal((B, desc.NP)))
        p = desc.param('x_init')
        full[:, p.col:p.col + p.size] = make_theta(desc, B, seed=1000 + rank)
        theta = full[:, solver._var_cols]
    else:
        solver.set_updated(['x_init'])
        theta = make_theta(desc, B, seed=1000 + rank)
    if args.instance_executor == 'stream' and desc.solver == 'OSQP' and getattr(solver, 'h_rs', None) is not None and solver.h_rs.value:
        solver.lib.check(solver.lib.L.cpg_hip_set_program_placement(solver.h_rs, 0), 'set_program_placement')
    stg = {}
    if args.max_iter:
        stg['max_iter'] = args.max_iter
    if args.eps:
        stg['eps_abs'] = stg['eps_rel'] = args.eps
    if args.check_termination:
        stg['check_termination'] = args.check_termination
    if args.debug_stage:
        stg['debug_stage'] = args.debug_stage
    solver.apply_settings(**stg)                 # reference defaults unless an experiment overrides them
    dev = DeviceBatch(solver, B)
    dev.upload(theta)

    # multi-GPU: the job's only exchange is the final gather of every rank's results to the root's device
    # memory, point-to-point over xGMI (RcclGather); it is part of every timed step
    gather, gather_kind, gather_ms = None, None, 0.0
    Btot = world * B
    if world > 1:
        from cvxpygen_amd.sharding import job_key, make_gather, result_spec
        def bcast_uid(raw):        # the RCCL unique id rides on the launcher's process group: no rendezvous file to go stale
            box = [raw]
            dist.broadcast_object_list(box, src=0)
            return box[0]
        def all_ranks(ok):         # every rank has its communicator, or every rank takes the host transport
            import torch
            t = torch.tensor([1 if ok else 0], dtype=torch.int32)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return bool(t.item())
        gather, kind, note = make_gather(solver, rank, world, key=job_key(), lib=args.rccl_lib, uid_exchange=bcast_uid,
                                         agree=all_ranks, init_timeout=args.rccl_timeout)
        if kind == 'rccl':
            gather_kind = 'rccl ncclSend/ncclRecv to rank 0 (device memory), on the solve stream'
        else:
            gather_kind = f'host (rccl init failed: {note}): D2H per rank + shared-memory slices read by rank 0'
            print(f'[bench rank {rank}] RCCL gather not available ({note}); results travel through host shared memory', file=sys.stderr)
        spec = result_spec(dev)
        arrays = [(k, dev._ptrs[k], B, rb) for k, (rb, dt, tail, nm) in spec.items()]

    def barrier():
        solver.synchronize()
        if dist is not None:
            import torch
            if torch.cuda.is_available():
                torch.cuda.synchronize()
            dist.barrier()

    def step():
        solver.solve_device(dev)
        if gather is not None and not args.no_gather:
            gather.enqueue(arrays, Btot)
        solver.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    kernel_ms = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ts = time.perf_counter()
        step()
        kernel_ms.append(solver.last_kernel_ms())
        gather_ms += 1e3 * (time.perf_counter() - ts) - kernel_ms[-1]
    barrier()
    elapsed = time.perf_counter() - t0
    rank_ms = None
    if dist is not None:
        import torch
        mine_ms = 1e3 * elapsed / max(1, args.steps)
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # every rank's own clock around the same steps: the spread says whether a rank (a GPU, its xGMI link to the root) lags
        tl = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(tl, torch.tensor([mine_ms], dtype=torch.float64))
        rank_ms = [float(v.item()) for v in tl]

    hybrid = desc.solver == 'OSQP' and bool(getattr(solver, '_hybrid', False)) and solver.h is solver.h_shared
    phase = solver.last_phase_ms() if hybrid else None      # split of the last step: (ms shared, ms per-instance, handed over)
    res = dev.download()
    iters = res.iter
    solved = int((res.status == 1).sum())
    stats = {'mean_iter': float(iters.mean()), 'max_iter': int(iters.max()), 'solved': solved,
             'not_solved': int(B - solved)}
    if gather is not None and not args.no_gather:
        # what the root holds after the last step: every rank's iteration counts and statuses
        alli = gather.fetch('iter', dev._ptrs['iter'], 4, Btot, np.int32)
        alls = gather.fetch('status', dev._ptrs['status'], 4, Btot, np.int32)
        if rank == 0:
            # the root must hold EVERY rank's rows after the last step: a shard the gather did not deliver would still carry the
            # fill pattern of the root's buffers (iteration count 0 / status 0 are not results of any solve)
            per_rank_ok = [bool((alli[r * B:(r + 1) * B] > 0).all() and np.isin(alls[r * B:(r + 1) * B], (1, 2, 3, 4, 5, 6, 7, 9)).all()) for r in range(world)]
            if not all(per_rank_ok):
                raise RuntimeError(f'gather incomplete: ranks whose rows are missing at the root: {[r for r, ok in enumerate(per_rank_ok) if not ok]}')
            stats = {'mean_iter': float(alli.mean()), 'max_iter': int(alli.max()),
                     'solved': int((alls == 1).sum()), 'not_solved': int((alls != 1).sum()),
                     'ranks_delivered': int(sum(per_rank_ok))}

    # whole path from host memory (H2D of theta + kernel + D2H of the results), N = 1 only:
    # page-locked buffers, transfers of batch i +- 1 hidden behind the kernel of batch i
    wall = None
    if world == 1 and not args.no_wall and desc.solver == 'OSQP':
        from cvxpygen_amd.runtime import PinnedStream
        nb = 8                                     # fill and drain of the pipeline (first H2D, last D2H) over eight batches
        ps = PinnedStream(solver, B, nb)
        for k in range(nb):
            ps.theta[k * B:(k + 1) * B] = theta
        ps.run()                                   # warm-up (first-touch of the pinned pages, buffer allocation)
        tw = time.perf_counter()
        ps.run()
        tw = time.perf_counter() - tw
        rs = ps.result()
        same = bool(np.array_equal(rs.iter[:B], res.iter) and np.array_equal(rs.prim_flat[-B:], res.prim_flat))
        wall = {'value': nb * B / tw, 'unit': 'QP instances/s', 'ms_per_batch': 1e3 * tw / nb, 'batches': nb,
                'bytes_h2d_per_batch': int(theta.nbytes),
                'bytes_d2h_per_batch': int(B * (8 * (solver.n_out_prim + solver.n_out_dual) + 32)),
                'same_results_as_resident_path': same,
                'what': 'theta in page-locked host memory -> results in page-locked host memory; H2D of batch i+1, '
                        'kernel of batch i, D2H of batch i-1 on three HIP streams (cpg_hip_solve_batches_pipelined)'}
        ps.free()

    if rank == 0:
        if stats['not_solved'] and args.workload in ('mpc12', 'mpc6', 'adp') and not args.all_params:
            print(f"WARNING: {stats['not_solved']} instances of a feasible workload were not solved", file=sys.stderr)
        n_prim, n_dual = len(solver.plan.prim_idx), len(solver.plan.dual_idx)
        bytes_per_inst = 8 * (solver.np_var + n_prim + n_dual) + 32
        if args.workload == 'portfolio':
            bytes_per_inst = 15416                                      # config 3 figure of SURVEY.md 8(d)
        if args.workload == 'adp':
            bytes_per_inst = 8 * (27 + 6 + 2) + 32                      # config 4: theta 27, u 6, dual 2, info
        k_ms = float(np.mean(kernel_ms))
        traffic = None          # recorded PMC measurement of the same command (profiles/), not re-measured live
        # what binds the dominant kernel, from the PMC passes under profiles/ (a property of the kernel design, reported next to
        # the contractual HBM figure; the traffic record below is replayed only on matching sources, this is not tied to it)
        binding = BINDING.get((args.workload, bool(args.all_params), bool(args.fixed_rho)))
        traffic_src = None
        traffic_stale = None
        binding_num = None      # per kernel: busy fractions of LDS / VALU / HBM and the share of wave cycles waiting (scripts/record_traffic.py)
        try:
            # records are stamped with the fingerprint of the kernel sources and name their kernel: a record taken on
            # other kernels is refused instead of silently replayed (scripts/record_traffic.py writes them)
            tj = json.load(open(os.path.join(ROOT, 'profiles', 'hbm_traffic.json')))
            key = args.workload + ('_all_params' if args.all_params else '') + ('_fixed_rho' if args.fixed_rho else '')
            rec = tj.get(key)
            if rec and rec['instances'] == B:
                if rec.get('source_fingerprint') == source_fingerprint():
                    traffic = rec['fetch_bytes'] + rec['write_bytes']
                    binding = rec.get('binding_resource') or binding
                    binding_num = rec.get('binding')
                    traffic_src = f"{rec.get('source')}; recorded {rec.get('recorded', 'date not recorded')}; source fingerprint {rec.get('source_fingerprint')}; kernel {rec.get('kernel')}"
                else:
                    traffic_stale = f"record {key} of profiles/hbm_traffic.json was taken on other kernel sources ({rec.get('source_fingerprint')}): refused"
        except (OSError, ValueError, KeyError):
            pass
        rnote = ('compulsory traffic only (theta in, solution out); the iteration state never leaves '
                 'registers/LDS, so this path is latency / LDS bound, not HBM bound (DESIGN.md section 6)')
        per_instance_kernel = desc.solver == 'OSQP' and solver.h is solver.h_ref
        rpl = (getattr(solver, '_rplan_s', None) if hybrid else getattr(solver, '_rplan', None)) if desc.solver == 'OSQP' else None
        shared_kernel = 'osqp_squad_kernel' if (desc.solver == 'OSQP' and squad_kernel_in_use(solver)) else 'osqp_shared_kernel'
        kernel_name = ('clarabel_kernel' if args.workload == 'adp' else 'osqp_refactor_kernel' if per_instance_kernel else shared_kernel)
        units = B
        stream = None
        out_plan_extra = {}
        sv = 8 * int(rpl.stats['sol_stream_entries']) if rpl is not None else 0
        phases = None
        if hybrid:
            # two kernels per step; the roofline object describes the one that takes longer
            ms1, ms2, n_ho = phase
            it = res.iter.astype(np.int64)
            from cvxpygen_amd.runtime import BUILD_OPTION_DEFAULTS
            ad = int({**BUILD_OPTION_DEFAULTS, **build_options}['adaptive_rho_interval'])
            # ESTIMATE of the iteration split (the kernels record only totals): an instance that runs past the first adaptation point is
            # counted as handed over there.  Instances whose estimate stays inside the tolerance band hand over later or never -- the
            # measured number of hand-overs (n_ho) and the two kernel times are what is exact.
            it1 = np.minimum(it, ad)
            ho = it > ad
            import ctypes as _C
            gv = _C.c_double(0)
            solver.lib.L.cpg_hip_get_setting(solver.h_rs, b'generated_instance_executor', _C.byref(gv))
            inst_kernel = 'osqp_instance_kernel' if gv.value else 'osqp_refactor_kernel'
            phases = {'shared_factor': {'kernel': shared_kernel, 'ms': ms1, 'instances': B,
                                        'iterations_estimate': int(it1.sum())},
                      'per_instance_factor': {'kernel': inst_kernel, 'ms': ms2, 'instances': n_ho,
                                              'iterations_estimate': int((it - it1)[ho].sum()), 'handed_over_by_iteration_count': int(ho.sum()),
                                              'note': 'instances handed over after a rho change: numeric LDL\' for the new rho, then ADMM with their own factor'}}
            # ONE accounting for the two-kernel step (round-5 review, item 5): a "launch" is the step's pair of kernels on one stream --
            # every instance's theta goes in and every instance's solution comes out exactly once across the two --, so the algorithmic
            # bytes are SURVEY 8(d)'s per-instance figure x all instances and the time is the sum of the two kernel times (HIP events:
            # ev0 .. ev_mid .. ev1); the split per kernel is in `phases`
            units = B
            algorithmic_bytes_launch = bytes_per_inst * B
            kernel_name = f'{shared_kernel} + {inst_kernel}'
            # (k_ms stays the mean over the timed steps of the span ev0 .. ev1 of both kernels; `phases` is the split of the LAST step)
            it2 = float((it - it1)[ho].mean()) if ho.any() else 0.0
            stream = None if gv.value else {'bytes_per_instance': int(it2 * sv), 'what': f'{it2:.1f} iterations x {sv} B of per-instance substitution coefficients',
                      'achieved': it2 * sv * n_ho / (ms2 * 1e-3) / 1e9}
            rnote = ('two kernels per step (shared factor until an instance\'s rho changes, per-instance factor behind it): `achieved` = SURVEY.md 8(d) bytes per '
                     'instance x all instances / (sum of the two kernel times); compulsory traffic only -- the iteration state never leaves registers / LDS, '
                     'the path is LDS / issue / latency bound, not HBM bound (DESIGN.md sections 4.5, 6)')
        elif per_instance_kernel and team_kernel_in_use(solver):
            # team per-instance factor kernel (cpg_osqp_team.h): one workgroup of W wavefronts per instance, coefficients, operand
            # offsets and output slots of a wavefront's steps in its registers -- no coefficient stream
            kernel_name = 'osqp_team_kernel'
            rpl_res = getattr(solver, '_rplan_res', None)
            stream = None
            rnote = ('team per-instance factor kernel: one workgroup of W wavefronts per instance, numeric LDL\' + block inverses in LDS, the merged '
                     'substitution program split over the wavefronts with its coefficients / offsets / slots in their registers, one barrier per '
                     'phase; `achieved` prices SURVEY.md 8(d) bytes per instance; the kernel is a chain of latencies, not HBM bound (DESIGN.md 4.7)')
            if rpl_res is not None:
                out_plan_extra = {'team_wavefronts': int(rpl_res.team), 'team_phases': int(rpl_res.sol.n_phases),
                                  'team_steps': int(rpl_res.stats.get('sol_steps', 0)), 'merged_groups': int(rpl_res.stats.get('merged', 0))}
            if binding_num is None:
                binding = ('latency: a chain of dependent phases per ADMM iteration (one barrier each) and of dependent levels per '
                           'factorisation at one instance per CU; nothing streamed per iteration (DESIGN.md 4.7)')
        elif per_instance_kernel and resident_kernel_in_use(solver):
            # resident per-instance factor kernel (cpg_osqp_resident.h): factor in LDS, substitution coefficients in registers --
            # no coefficient stream; what it reads from memory per instance beyond SURVEY.md 8(d)'s bytes is in DESIGN.md 4.6
            kernel_name = 'osqp_resident_kernel'
            rpl_res = getattr(solver, '_rplan_res', None)
            stream = None
            rnote = ('resident per-instance factor kernel: numeric LDL\' + block inverses in LDS, merged substitution program with its '
                     'coefficients in registers (one wavefront per SIMD, 512 registers); `achieved` prices SURVEY.md 8(d) bytes per instance. '
                     'Per instance it moves ~100 KB of set-up state once, 74 KB of coefficients per termination test (reloaded into registers) and '
                     '~60 KB of program-order matrix copies per product of the test -- not the 73 KB PER ITERATION of the streaming kernel '
                     '(DESIGN.md section 4.6); the kernel is latency bound, not HBM bound')
            if rpl_res is not None:
                out_plan_extra = {'resident_phases': int(rpl_res.sol.n_phases), 'resident_steps': int(rpl_res.stats.get('sol_steps', 0)),
                                  'merged_groups': int(rpl_res.stats.get('merged', 0))}
        elif per_instance_kernel:
            # per-instance factor: every ADMM iteration streams the substitution coefficients of the
            # instance (8 bytes per entry of the streaming layout) from its buffer in HBM
            stream = {'bytes_per_instance': int(stats['mean_iter'] * sv), 'what': f"{stats['mean_iter']:.1f} iterations x {sv} B of per-instance substitution coefficients",
                      'achieved': stats['mean_iter'] * sv * B / (k_ms * 1e-3) / 1e9}
            rnote = ('`achieved` prices SURVEY.md 8(d) bytes per instance; `stream` the per-instance substitution coefficients read from HBM in every ADMM '
                     'iteration (DESIGN.md section 4.2); the shared index tables stay in L2')
            if args.workload == 'portfolio' and traffic_src is None:
                binding = ('HBM stream + latency: streaming per-instance factor kernel (this library carries no resident executor for the family): '
                           'substitution coefficients re-read from HBM in every ADMM iteration (profiles/r3_final7_pmc_config3.txt)')
        if not hybrid:
            algorithmic_bytes_launch = bytes_per_inst * units
        achieved = algorithmic_bytes_launch / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        value = world * B * args.steps / elapsed
        out = {
            # (BASELINE.json's metric; the driver's contract quotes `value` with the inputs already resident in HBM -- said in the string,
            # the PCIe-inclusive whole-path rate of SURVEY.md 8(d) is `wall_pcie.value` beside it)
            'metric': ('SOCP instances solved/sec (batched ADP, conic interior point; inputs resident in HBM)' if args.workload == 'adp'
                       else 'QP instances solved/sec (batched MPC QP, OSQP; inputs resident in HBM)'),
            'value': value, 'unit': ('SOCP instances/s' if args.workload == 'adp' else 'QP instances/s'), 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64',
            'data': ('synthetic (examples/portfolio.ipynb cell 7 draws, default_rng(1000+rank))'
                     if args.workload == 'portfolio' else
                     'synthetic (state = -2 + 4*U(0,1), default_rng(1000+rank); tests/test_E2E_SOCP.py:38-62)'
                     if args.workload == 'adp' else
                     'synthetic (x_init = -2 + 4*U(0,1), default_rng(1000+rank); family parameters of '
                     'examples/MPC.ipynb cell 3 extended to 12/4)'),
            'config': {'workload': label, 'instances_per_gpu': B, 'kkt_dim': desc.n_var + desc.m,
                       'n_var': desc.n_var, 'n_constr': desc.m, 'varying_params': (['a', 'F', 'Sig_f_sqrt', 'd_sqrt', 'w_prev'] if args.workload == 'portfolio' else
                                          ['f', 'G'] if args.workload == 'adp' else
                                          'all (matrix parameters: per-instance refactorisation)' if args.all_params else ['x_init']),
                       'settings': ('Clarabel defaults of the generated solver (cvxpygen/solvers/clarabel.py:63-119): '
                                    'tol_gap/feas 1e-8, max_iter 200, new solver per instance' if args.workload == 'adp' else
                                    'OSQP defaults of the generated solver: eps_abs=eps_rel=1e-3, max_iter=4000, '
                                    'check_termination=25, cold start; OSQP >= 1.0 library defaults restored by every cpg_solve: rho adapted '
                                    'every 50 iterations (tolerance 5), duality-gap test on' if args.osqp1 else
                                    'OSQP defaults of the generated solver: eps_abs=eps_rel=1e-3, '
                                    'max_iter=4000, check_termination=25, rho=0.1 fixed, cold start'),
                       'parallelism': f'shard{world}', **stats,
                       'library': os.path.relpath(solver.lib.path, ROOT),
                       'plan': {**{k: (round(v, 3) if isinstance(v, float) else v)
                                   for k, v in solver.plan.stats.items()}, **out_plan_extra}},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic,
                         'traffic_note': (f'REPLAYED, not measured in this run: bytes per step (all kernels of the step) from the rocprofv3 PMC '
                                          f'passes recorded in profiles/hbm_traffic.json ({traffic_src})') if traffic else traffic_stale,
                         'binding_resource': binding,
                         # busy fractions per kernel of the step (LDS array, VALU, HBM side, share of wave cycles waiting) from the stamped PMC record
                         'binding': [dict(v, kernel=k) for k, v in (binding_num or {}).items() if any(nm in k for nm in kernel_name.split(' + '))] or None,
                         'kernel': kernel_name, 'kernel_ms': k_ms, 'units_per_launch': int(units), 'algorithmic_bytes_per_launch': int(algorithmic_bytes_launch),
                         'algorithmic_bytes_per_instance': bytes_per_inst, 'stream': stream,
                         'note': rnote},
        }
        out['config']['value_is'] = ('inputs resident in HBM when the timed region starts, results left in HBM (driver contract); '
                                     'the PCIe-inclusive whole-path rate of SURVEY.md 8(d) is reported beside it in "wall_pcie"')
        if phases is not None:
            out['phases'] = phases
        if wall is not None:
            out['wall_pcie'] = wall
        if world > 1:
            out['config']['gather'] = ('left out of the timed steps (--no-gather)' if args.no_gather else gather_kind)
            out['config']['gather_ms_per_step_rank0'] = gather_ms / max(1, args.steps)
            # what travels per step: every non-root rank's result rows (solutions + the five info scalars), point to point to the root
            out['config']['gather_bytes_per_step'] = int(sum(rb for (rb, dt, tail, nm) in spec.values()) * B * (world - 1))
            if rank_ms is not None:
                out['config']['rank_ms_per_step'] = {'min': min(rank_ms), 'max': max(rank_ms), 'per_rank': [round(v, 3) for v in rank_ms]}
        if world == 1 and not args.no_cpu_baseline and not args.all_params:
            if args.workload == 'portfolio':
                out['cpu_baseline'] = cpu_baseline_portfolio(desc, args.cpu_seconds, **oracle_mode)
            elif args.workload == 'adp':
                out['cpu_baseline'] = cpu_baseline_adp(desc, min(args.cpu_seconds, 10.0))
            else:
                out['cpu_baseline'] = cpu_baseline(desc, args.cpu_seconds, **oracle_mode)
            if desc.solver == 'OSQP':
                out['cpu_baseline']['mode'] = 'fixed rho, no duality-gap test' if args.fixed_rho else 'OSQP >= 1.0 defaults (same mode as value)'
        if (world == 1 and desc.solver == 'OSQP' and not args.fixed_rho and not args.no_fixed_rho_leg and not args.all_params
                and args.workload in ('mpc12', 'mpc6')):
            # the other fork of the reference's default, same workload and batch: one shared-factor kernel
            fs = BatchSolver(desc, device=local_rank, lib_path=lib_path, build_options=dict(BUILD_OPTIONS_FIXED_RHO), plan=plan)
            fs.set_launch(args.waves, args.ipw, args.blocks_per_cu)
            fs.set_program_placement(args.placement)
            fs.set_updated(['x_init'])
            fs.apply_settings(**stg)
            fdev = DeviceBatch(fs, B)
            fdev.upload(theta)
            for _ in range(2):
                fs.solve_device(fdev); fs.synchronize()
            nf = max(3, min(args.steps, 10))
            tf = time.perf_counter()
            for _ in range(nf):
                fs.solve_device(fdev); fs.synchronize()
            tf = time.perf_counter() - tf
            fr = fdev.download()
            out['fixed_rho'] = {'value': B * nf / tf, 'unit': 'QP instances/s', 'steps': nf, 'ms_per_step': 1e3 * tf / nf,
                                'kernel_ms': fs.last_kernel_ms(), 'mean_iter': float(fr.iter.mean()), 'solved': int((fr.status == 1).sum()),
                                'what': 'same workload with build options adaptive_rho=0, check_dualgap=0 (a solver whose OSQP never adapts rho): '
                                        'one shared-factor kernel; NOT the mode `value` is quoted in'}
            fdev.free(); fs.close()
        if args.adjoint and desc.solver == 'OSQP':
            # config 5: forward as above, then the adjoint with upstream dX = dU = 0.1
            gs = BatchSolver(desc, device=local_rank, lib_path=lib_path, full_output=True, build_options=build_options, plan=plan)
            Bg = min(B, 20000)
            x0 = make_theta(desc, Bg, seed=77)
            fw = gs.solve({'x_init': x0}, updated_params=['x_init'])
            dv = {v.name: np.full((Bg,) + tuple(v.shape), 0.1) for v in desc.variables}
            t0 = time.perf_counter()
            gs.gradient({'x_init': x0}, fw.sol_x, fw.sol_y, dv, updated_params=['x_init'])
            tg = time.perf_counter() - t0
            t0 = time.perf_counter()
            g = gs.gradient({'x_init': x0}, fw.sol_x, fw.sol_y, dv, updated_params=['x_init'])
            tg = min(tg, time.perf_counter() - t0)
            ms = C_float()
            gs.lib.L.cpg_hip_last_kernel_ms(gs.h_grad, ms)
            out['adjoint'] = {'instances': Bg, 'wall_ms_incl_pcie': 1e3 * tg, 'kernel_ms': float(ms.value),
                              'adjoints_per_s_kernel': Bg / (ms.value * 1e-3),
                              'dtheta_shape': list(g['_flat'].shape), 'kernel': 'osqp_gradient_kernel'}
            gs.close()
        if args.check:
            from oracle import binding as ob
            nchk = min(B, 64 if args.workload == 'portfolio' else 256)
            th = np.tile(desc.theta0, (nchk, 1))
            if args.workload == 'portfolio':
                upd = list(pv.keys())
                th[:, solver._var_cols] = theta[:nchk]            # theta_var columns of the varying parameters
            elif args.all_params:
                nchk = min(nchk, 64)                              # (a refactorisation per instance on the host)
                th = th[:nchk]
                upd = [q_.name for q_ in desc.params]
                th[:, solver._var_cols] = theta[:nchk]
            else:
                upd = ['x_init']
                p = desc.param('x_init')
                th[:, p.col:p.col + p.size] = theta[:nchk]
            stg_o = {k_: v_ for k_, v_ in stg.items() if k_ != 'debug_stage'}
            o = ob.cpg_solve_batch(desc, th, upd, **oracle_mode, **stg_o)
            po = np.concatenate([o['sol_x'][:, v.indices] for v in desc.variables], axis=1)
            do = np.concatenate([o['sol_y'][:, d.indices] for d in desc.duals], axis=1)
            out['check'] = {
                'n': nchk, 'iter_mismatch': int((o['iter'] != res.iter[:nchk]).sum()),
                'prim_relerr': float(np.abs(po - res.prim_flat[:nchk]).max() / np.abs(po).max()),
                'dual_relerr': float(np.abs(do - res.dual_flat[:nchk]).max() / np.abs(do).max())}
        print(json.dumps(out), flush=True)
    dev.free()
    if gather is not None:
        gather.close()
    solver.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
