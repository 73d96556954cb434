"""
TEST INFRASTRUCTURE ONLY -- ctypes binding of oracle/liboracle.so (the scalar-C restatement in
oracle/osqp_oracle.c).  Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg; never by cvxpygen_amd/.
"""

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

SETTING_ORDER = ['rho', 'sigma', 'alpha', 'scaling', 'max_iter', 'eps_abs', 'eps_rel',
                 'eps_prim_inf', 'eps_dual_inf', 'scaled_termination', 'check_termination',
                 'warm_starting', 'adaptive_rho', 'adaptive_rho_interval',
                 'adaptive_rho_tolerance', 'check_dualgap']

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def build(force=False):
    so = os.path.join(_HERE, 'liboracle.so')
    srcs = [os.path.join(_HERE, f) for f in ('osqp_oracle.c', 'clarabel_oracle.c', 'Makefile')]
    def stale():
        return force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(f) for f in srcs)
    if stale():
        # several ranks / pytest workers may import this at once: one of them builds, the others wait for the lock and find
        # the library up to date
        import fcntl
        with open(os.path.join(_HERE, '.build.lock'), 'w') as lk:
            fcntl.flock(lk, fcntl.LOCK_EX)
            try:
                if stale():
                    subprocess.check_call(['make', '-C', _HERE, '-B', 'liboracle.so'], stdout=subprocess.DEVNULL)
                    force = False
            finally:
                fcntl.flock(lk, fcntl.LOCK_UN)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, 'liboracle.so')
        build()
        L = C.CDLL(so)
        L.oracle_setup.restype = C.c_void_p
        L.oracle_setup.argtypes = [C.c_int, C.c_int, _ip, _ip, _dp, _dp, _ip, _ip, _dp, _dp, _dp, _dp]
        L.oracle_clone.restype = C.c_void_p
        L.oracle_clone.argtypes = [C.c_void_p]
        L.oracle_free.argtypes = [C.c_void_p]
        L.oracle_set_settings.argtypes = [C.c_void_p, _dp]
        L.oracle_default_settings.argtypes = [_dp]
        L.oracle_dims.argtypes = [C.c_void_p, _ip]
        L.oracle_get_scaling.argtypes = [C.c_void_p, _dp, _dp, _dp]
        L.oracle_update_vec.argtypes = [C.c_void_p, _dp, _dp, _dp]
        L.oracle_update_mat.argtypes = [C.c_void_p, _dp, _dp]
        L.oracle_solve.argtypes = [C.c_void_p, _dp, _dp]
        L.oracle_info.argtypes = [C.c_void_p, _dp]
        L.oracle_kkt_solve.argtypes = [C.c_void_p, _dp, _dp]
        L.oracle_warm_start.argtypes = [C.c_void_p, _dp, _dp]
        L.oracle_cpg_solve_batch.restype = C.c_int
        L.oracle_cpg_solve_batch.argtypes = [
            C.c_void_p, C.c_int, _ip, C.POINTER(_ip), C.POINTER(_ip), C.POINTER(_dp), _ip, C.c_int,
            C.c_int, C.c_long, _dp, _dp, _dp, _dp, C.c_int]
        L.oracle_num_threads.restype = C.c_int
        L.clarabel_oracle_solve_batch.restype = C.c_int
        L.clarabel_oracle_solve_batch.argtypes = [
            C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _ip, _ip, _ip, _ip, _ip,
            _ip, C.POINTER(_ip), C.POINTER(_ip), C.POINTER(_dp), C.c_int, C.c_int, C.c_int, C.c_long, _dp, _dp,
            _dp, _dp, _dp, C.c_int]
        _LIB = L
    return _LIB


def _d(a):
    return None if a is None else a.ctypes.data_as(_dp)


def _i(a):
    return a.ctypes.data_as(_ip)


def settings_vector(**kw):
    L = lib()
    s = np.zeros(len(SETTING_ORDER))
    L.oracle_default_settings(_d(s))
    for k, v in kw.items():
        s[SETTING_ORDER.index(k)] = float(v)
    return s


class Oracle:
    """One embedded-OSQP workspace, set up at (P, q, A, l, u) like `osqp.OSQP().setup()`."""

    def __init__(self, P, q, A, l, u, **settings):
        L = lib()
        P, A = P.tocsc(), A.tocsc()
        self.n, self.m = P.shape[0], A.shape[0]
        self._keep = [np.ascontiguousarray(P.indptr, dtype=np.int32),
                      np.ascontiguousarray(P.indices, dtype=np.int32),
                      np.ascontiguousarray(P.data, dtype=np.float64),
                      np.ascontiguousarray(q, dtype=np.float64),
                      np.ascontiguousarray(A.indptr, dtype=np.int32),
                      np.ascontiguousarray(A.indices, dtype=np.int32),
                      np.ascontiguousarray(A.data, dtype=np.float64),
                      np.ascontiguousarray(l, dtype=np.float64),
                      np.ascontiguousarray(u, dtype=np.float64)]
        k = self._keep
        self.stg = settings_vector(**settings)
        self.h = L.oracle_setup(self.n, self.m, _i(k[0]), _i(k[1]), _d(k[2]), _d(k[3]), _i(k[4]),
                                _i(k[5]), _d(k[6]), _d(k[7]), _d(k[8]), _d(self.stg))
        if not self.h:
            raise RuntimeError('oracle_setup failed')

    def __del__(self):
        if getattr(self, 'h', None):
            lib().oracle_free(self.h)
            self.h = None

    def set(self, **kw):
        for k, v in kw.items():
            self.stg[SETTING_ORDER.index(k)] = float(v)
        lib().oracle_set_settings(self.h, _d(self.stg))

    def dims(self):
        out = np.zeros(5, dtype=np.int32)
        lib().oracle_dims(self.h, _i(out))
        return dict(n=out[0], m=out[1], nnzL=out[2], nnzK=out[3], n_refactor=out[4])

    def scaling(self):
        D, E, c = np.zeros(self.n), np.zeros(self.m), np.zeros(1)
        lib().oracle_get_scaling(self.h, _d(D), _d(E), _d(c))
        return D, E, c[0]

    def update_vec(self, q=None, l=None, u=None):
        f = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.float64)
        q, l, u = f(q), f(l), f(u)
        return lib().oracle_update_vec(self.h, _d(q), _d(l), _d(u))

    def update_mat(self, Px=None, Ax=None):
        f = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.float64)
        Px, Ax = f(Px), f(Ax)
        return lib().oracle_update_mat(self.h, _d(Px), _d(Ax))

    def kkt_solve(self, b):
        b = np.ascontiguousarray(b, dtype=np.float64)
        out = np.zeros_like(b)
        lib().oracle_kkt_solve(self.h, _d(b), _d(out))
        return out

    def solve(self, warm=False):
        self.set(warm_starting=1 if warm else 0)
        x, y, info = np.zeros(self.n), np.zeros(self.m), np.zeros(6)
        rc = lib().oracle_solve(self.h, _d(x), _d(y))
        lib().oracle_info(self.h, _d(info))
        return dict(x=x, y=y, obj_val=info[0], iter=int(info[1]), status=int(info[2]),
                    prim_res=info[3], dual_res=info[4], rho=info[5], rc=rc)


def cpg_solve_batch(desc, theta, updated_params=None, nthreads=0, oracle=None, **settings):
    """Reference-semantics batch: per instance canonicalise -> update -> cold-start solve ->
    retrieve.  theta: (B, NP) or (B, NP+1).  Returns dict with sol_x, sol_y, obj_val, iter,
    status, pri_res, dua_res and user-facing prim / dual arrays."""
    from cvxpygen_amd.canon_builder import canon_lu  # data model only (no solver code)
    L = lib()
    theta = np.asarray(theta, dtype=np.float64)
    B = theta.shape[0]
    if theta.shape[1] == desc.NP:
        theta = np.concatenate([theta, np.ones((B, 1))], axis=1)
    theta = np.ascontiguousarray(theta)
    if oracle is None:
        c0 = desc.default_canon()
        l0, u0 = canon_lu(desc, c0)
        oracle = Oracle(desc.P, c0['q'], desc.A, l0, u0, **settings)
    else:
        oracle.set(**settings)
    dep = desc.user_p_name_to_canon_outdated()
    if updated_params is None:
        updated_params = desc.param_names
    outd = set()
    for p in updated_params:
        outd.update(dep[desc.param(p).name])
    ids = ['P', 'q', 'd', 'A', 'l', 'u']
    outdated = np.array([1 if (i in outd and desc.changes[i]) else 0 for i in ids], dtype=np.int32)
    rows = np.zeros(6, dtype=np.int32)
    ps, is_, xs, keep = (_ip * 6)(), (_ip * 6)(), (_dp * 6)(), []
    for k, pid in enumerate(ids):
        Cm = desc.maps[pid].tocsr()
        a = np.ascontiguousarray(Cm.indptr, dtype=np.int32)
        b = np.ascontiguousarray(Cm.indices, dtype=np.int32)
        c = np.ascontiguousarray(Cm.data, dtype=np.float64)
        keep += [a, b, c]
        rows[k] = Cm.shape[0]
        ps[k], is_[k], xs[k] = _i(a), _i(b), _d(c)
    if not desc.nonzero_d:
        rows[2] = 0
    n, m = desc.n_var, desc.m
    sol_x, sol_y, info = np.zeros((B, n)), np.zeros((B, m)), np.zeros((B, 5))
    rc = L.oracle_cpg_solve_batch(oracle.h, desc.n_eq, _i(rows), ps, is_, xs, _i(outdated),
                                  int(desc.is_maximization), desc.NP + 1, B, _d(theta), _d(sol_x),
                                  _d(sol_y), _d(info), nthreads)
    out = dict(sol_x=sol_x, sol_y=sol_y, obj_val=info[:, 0], iter=info[:, 1].astype(np.int32),
               status=info[:, 2].astype(np.int32), pri_res=info[:, 3], dua_res=info[:, 4], rc=rc)
    out['prim'] = {v.name: sol_x[:, v.indices] for v in desc.variables}
    out['dual'] = {d.name: sol_y[:, d.indices] for d in desc.duals}
    return out


def qp_adjoint(desc, canon, x, y, dx):
    """C restatement of cpg_osqp_gradient for one instance; canon = desc.canon_at(theta).
    Returns dict(r, dq, dl, du, dP (nnzP), dA (nnzA), dtheta (NP))."""
    import scipy.sparse as sp
    L = lib()
    L.oracle_qp_adjoint.restype = C.c_int
    L.oracle_qp_adjoint.argtypes = [C.c_int, C.c_int, _ip, _ip, _dp, _ip, _ip, _dp, _dp, _dp, _dp,
                                    _dp, _dp, _dp, _dp, _dp, _dp]
    n, m = desc.n_var, desc.m
    Pp = np.ascontiguousarray(desc.P.indptr, dtype=np.int32); Pi = np.ascontiguousarray(desc.P.indices, dtype=np.int32)
    Ap = np.ascontiguousarray(desc.A.indptr, dtype=np.int32); Ai = np.ascontiguousarray(desc.A.indices, dtype=np.int32)
    Px = np.ascontiguousarray(canon['P'], dtype=np.float64); Ax = np.ascontiguousarray(canon['A'], dtype=np.float64)
    f = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    x, y, dx = f(x), f(y), f(dx)
    r, dq, dl, du = np.zeros(n + m), np.zeros(n), np.zeros(m), np.zeros(m)
    dP, dA = np.zeros(len(Px)), np.zeros(len(Ax))
    rc = L.oracle_qp_adjoint(n, m, _i(Pp), _i(Pi), _d(Px), _i(Ap), _i(Ai), _d(Ax), _d(x), _d(y), _d(dx),
                             _d(r), _d(dq), _d(dl), _d(du), _d(dP), _d(dA))
    if rc:
        raise RuntimeError('oracle_qp_adjoint failed')
    dth = np.zeros(desc.NP + 1)
    for pid, vec in (('q', dq), ('l', dl[:desc.n_eq]), ('u', du), ('P', dP), ('A', dA)):
        if desc.changes.get(pid, False):
            dth += sp.csr_matrix(desc.maps[pid]).T @ vec
    return dict(r=r, dq=dq, dl=dl, du=du, dP=dP, dA=dA, dtheta=dth[:desc.NP])


class CpgSession:
    """What ONE process of the reference does over successive `cpg_solve()` calls: a static workspace
    (cvxpygen/utils.py:470-689) -- cpg_params_vec, Canon_Outdated flags (all raised at start,
    utils.py:559-562), and the OSQP workspace with its scaling, factor, iterates and rho.
    `solve(values)` = cpg_update_<param> for the listed parameters, canonicalise what is outdated,
    osqp_update_data_mat then _vec (solvers/osqp.py:20-59), osqp_solve, cpg_retrieve_info."""

    def __init__(self, desc, **build_settings):
        from cvxpygen_amd.canon_builder import canon_lu
        self._canon_lu = canon_lu
        self.desc = desc
        c0 = desc.default_canon()
        l0, u0 = canon_lu(desc, c0)
        self.build = dict(build_settings)
        self.oracle = Oracle(desc.P, c0['q'], desc.A, l0, u0, **build_settings)
        self.theta = np.array(desc.theta0, dtype=np.float64)
        self.outdated = {pid for pid in desc.maps if desc.changes.get(pid, False)}

    def solve(self, values=None, warm=True, **settings):
        desc = self.desc
        dep = desc.user_p_name_to_canon_outdated()
        for name, v in (values or {}).items():
            p = desc.param(name)
            self.theta[p.col:p.col + p.size] = desc.flatten_param(name, v)
            self.outdated.update(pid for pid in dep[name] if desc.changes.get(pid, False))
        canon = desc.canon_at(self.theta)
        # settings: defaults of the generated solver, then the call's (templates/cpg_solver.py.jinja2:55-60)
        stg = dict(max_iter=4000, eps_abs=1e-3, eps_rel=1e-3, eps_prim_inf=1e-4, eps_dual_inf=1e-4,
                   scaled_termination=0, check_termination=25)
        stg.update(settings)
        self.oracle.set(**stg)
        od = self.outdated
        if od & {'P', 'A'}:
            self.oracle.update_mat(canon['P'] if 'P' in od else None, canon['A'] if 'A' in od else None)
        if od & {'q', 'l', 'u'}:
            l, u = self._canon_lu(desc, canon)
            self.oracle.update_vec(canon['q'] if 'q' in od else None, l if 'l' in od else None, u if 'u' in od else None)
        self.outdated = set()
        r = self.oracle.solve(warm=warm)
        d = float(np.atleast_1d(canon['d'])[0]) if desc.nonzero_d else 0.0
        ov = r['obj_val'] + d
        r['obj_val'] = -ov if desc.is_maximization else ov
        return r


CLARABEL_SETTING_ORDER = [
    'max_iter', 'max_step_fraction', 'tol_gap_abs', 'tol_gap_rel', 'tol_feas', 'tol_infeas_abs', 'tol_infeas_rel', 'tol_ktratio',
    'reduced_tol_gap_abs', 'reduced_tol_gap_rel', 'reduced_tol_feas', 'reduced_tol_infeas_abs', 'reduced_tol_infeas_rel',
    'reduced_tol_ktratio', 'equilibrate_enable', 'equilibrate_max_iter', 'equilibrate_min_scaling', 'equilibrate_max_scaling',
    'linesearch_backtrack_step', 'min_switch_step_length', 'min_terminate_step_length', 'static_regularization_enable',
    'static_regularization_constant', 'static_regularization_proportional', 'dynamic_regularization_enable',
    'dynamic_regularization_eps', 'dynamic_regularization_delta', 'iterative_refinement_enable', 'iterative_refinement_reltol',
    'iterative_refinement_abstol', 'iterative_refinement_max_iter', 'iterative_refinement_stop_ratio']


def clarabel_solve_batch(desc, theta, nthreads=0, **settings):
    """The conic restatement in C (oracle/clarabel_oracle.c: the algorithm of oracle/clarabel_numpy.py statement for
    statement) on a batch of one conic family: per instance canonicalise, NEW solver, solve, retrieve.  Same result dict as
    clarabel_numpy.cpg_solve_batch.  Settings not given take clarabel_numpy.DEFAULTS."""
    from oracle import clarabel_numpy as cl
    L = lib()
    if list(cl.DEFAULTS) != CLARABEL_SETTING_ORDER:
        raise RuntimeError('clarabel_numpy.DEFAULTS and the C settings vector disagree')
    stg = dict(cl.DEFAULTS)
    for k, v in settings.items():
        if k not in stg:
            raise KeyError(k)
        stg[k] = v
    sv = np.array([float(stg[k]) for k in CLARABEL_SETTING_ORDER])
    theta = np.asarray(theta, dtype=np.float64)
    B = theta.shape[0]
    if theta.shape[1] == desc.NP:
        theta = np.concatenate([theta, np.ones((B, 1))], axis=1)
    theta = np.ascontiguousarray(theta)
    n, m = desc.n_var, desc.m
    soc = np.ascontiguousarray(desc.cones['soc'], dtype=np.int32)
    P, A = desc.P.tocsc(), desc.A.tocsc()
    keep = [np.ascontiguousarray(P.indptr, dtype=np.int32), np.ascontiguousarray(P.indices, dtype=np.int32),
            np.ascontiguousarray(A.indptr, dtype=np.int32), np.ascontiguousarray(A.indices, dtype=np.int32)]
    ids = ['P', 'q', 'd', 'A', 'b']
    rows = np.zeros(5, dtype=np.int32)
    ps, is_, xs = (_ip * 5)(), (_ip * 5)(), (_dp * 5)()
    for k, pid in enumerate(ids):
        Cm = desc.maps[pid].tocsr()
        a = np.ascontiguousarray(Cm.indptr, dtype=np.int32)
        b = np.ascontiguousarray(Cm.indices, dtype=np.int32)
        c = np.ascontiguousarray(Cm.data, dtype=np.float64)
        keep += [a, b, c]
        rows[k] = Cm.shape[0]
        ps[k], is_[k], xs[k] = _i(a), _i(b), _d(c)
    sol_x, sol_z, info = np.zeros((B, n)), np.zeros((B, m)), np.zeros((B, 5))
    rc = L.clarabel_oracle_solve_batch(n, m, int(desc.cones['zero']), int(desc.cones['nonneg']), len(soc), _i(soc) if len(soc) else None,
                                       _i(keep[0]), _i(keep[1]), _i(keep[2]), _i(keep[3]), _i(rows), ps, is_, xs,
                                       int(desc.is_maximization), int(bool(desc.nonzero_d)), desc.NP + 1, B, _d(theta), _d(sv),
                                       _d(sol_x), _d(sol_z), _d(info), int(nthreads))
    if rc:
        raise RuntimeError('clarabel_oracle_solve_batch: cone dimensions do not add up to m')
    out = dict(sol_x=sol_x, sol_z=sol_z, obj_val=info[:, 0].copy(), iter=info[:, 1].astype(np.int32),
               status=info[:, 2].astype(np.int32), pri_res=info[:, 3].copy(), dua_res=info[:, 4].copy())
    out['prim'] = {v.name: sol_x[:, v.indices] for v in desc.variables}
    out['dual'] = {d.name: sol_z[:, d.indices] for d in desc.duals}
    return out
