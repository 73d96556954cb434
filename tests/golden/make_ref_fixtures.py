"""
Generates tests/golden/ref_canon_fixtures.npz + ref_emitted_c.json: outputs of the REFERENCE's own
code for the canonicalisation conventions of the hot path.

Runs ONLY in the build container (needs /root/reference, gcc); the fixtures it writes are plain data
(inputs + what the reference computed from them) and are the only thing that travels.  No reference
source is copied: the modules are loaded from where they lie,

    cvxpygen/mappings.py              dataclasses (AffineMap, ParameterCanon, ParameterInfo, ConstraintInfo)
    cvxpygen/solvers/_interface.py    QPCanonMixin.get_affine_map / augment_vector_parameter (:39-79),
                                      SolverInterface.get_affine_map (:132-173, conic ids)
    cvxpygen/utils.py                 replace_inf (:213-228), write_canonicalize (:279-294),
                                      write_vec_def / write_mat_def (:87-131)

by file path (they import numpy / scipy / jinja2 only; NO stand-ins for cvxpy, osqp or any other
third-party package are created -- only an empty `cvxpygen` namespace package so that
`from cvxpygen.mappings import ...` inside _interface.py resolves to the reference's own mappings.py).
`cvxpygen/canonicalizer.py` cannot be imported (top-level `import cvxpy`); its cvxpy-free helpers

    Canonicalizer._update_to_dense_mapping (:425-433)   _update_adjacency_matrix (:436-446)
    Canonicalizer._set_default_values (:448-486)        the ConstraintInfo arithmetic (:270-281)

are compiled from the file's own text (ast) and driven in the order of
`_process_canonical_parameters` (:283-332), restated below.

Inputs: the arrays `problem.get_problem_data()` would hand over (reduced_P / reduced_A maps, q map,
problem_data_index), rebuilt from this repository's hand-canonicalised families with
`cvxpygen_amd.canonicalizer.reduced_from_descriptor` -- so what is pinned is "given cvxpy's data, the
per-canonical-parameter maps, defaults, change flags and adjacency are the reference's", not cvxpy's
own canonicalisation (cvxpy is not installable here).

The second fixture holds numbers produced by C code the reference's emitters wrote
(`write_canonicalize` over maps emitted by `write_mat_def`, theta by `write_vec_def`), compiled with
gcc: the canonical parameters p = C_p [theta; 1] exactly as a generated `cpg_canonicalize_<p>` computes them.
"""
import ast
import importlib.util
import json
import os
import subprocess
import sys
import tempfile
import types

import numpy as np
import scipy.sparse as sp
from scipy import sparse

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, '..', '..'))
REF = '/root/reference/cvxpygen'
sys.path.insert(0, ROOT)

from cvxpygen_amd import families                                     # noqa: E402
from cvxpygen_amd.canonicalizer import reduced_from_descriptor        # noqa: E402


def _load_reference():
    pkg = types.ModuleType('cvxpygen')
    pkg.__path__ = []                       # empty namespace: nothing of cvxpygen/__init__.py runs
    sys.modules['cvxpygen'] = pkg

    def load(name, rel):
        spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[name] = mod
        spec.loader.exec_module(mod)
        return mod
    mappings = load('cvxpygen.mappings', 'mappings.py')
    interface = load('cvxpygen.solvers._interface', os.path.join('solvers', '_interface.py'))
    utils = load('cvxpygen.utils', 'utils.py')
    # cvxpy-free helpers of canonicalizer.py, compiled from the file's own text
    tree = ast.parse(open(os.path.join(REF, 'canonicalizer.py')).read())
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == 'Canonicalizer')
    want = ('_update_to_dense_mapping', '_update_adjacency_matrix', '_set_default_values')
    fns = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name in want]
    for f in fns:
        f.decorator_list = []
    ns = {'np': np, 'sparse': sparse}
    exec(compile(ast.Module(body=fns, type_ignores=[]), 'canonicalizer.py (helpers)', 'exec'), ns)
    return mappings, interface, utils, ns


class _Reduced:
    def __init__(self, mat, index):
        self.reduced_mat = mat
        self.problem_data_index = index


class _ParamProb:
    """the three attributes of cvxpy's ParamQuadProg / ParamConeProg the reference reads"""
    def __init__(self, red_P, P_index, q_map, red_A, A_index):
        self.reduced_P = _Reduced(red_P, P_index)
        self.reduced_A = _Reduced(red_A, A_index)
        self.q = q_map


def reference_canon(desc, mappings, interface, helpers):
    """`_process_canonical_parameters` (canonicalizer.py:283-332) on the reduced data of `desc`,
    using the reference's get_affine_map and helper functions."""
    red_P, P_index, q_map, red_A, A_index = reduced_from_descriptor(desc)
    pp = _ParamProb(None if red_P is None else sp.csr_matrix(red_P), P_index, sp.csr_matrix(q_map),
                    sp.csr_matrix(red_A), A_index)
    conic = desc.solver != 'OSQP'

    if conic:
        class Iface(interface.SolverInterface):                       # base get_affine_map, conic ids
            canon_p_ids = ['P', 'q', 'd', 'A', 'b']
            canon_p_ids_constr_vec = ['b']
            stgs = {}
            dual_var_split = False
            solver_type = 'conic'

            def generate_code(self, *a, **k):
                pass
        # ClarabelInterface.__init__ (solvers/clarabel.py:121-123): every cone row counts as an "equality" row
        si = Iface('CLARABEL', desc.n_var, desc.m, 0, pp, {}, [])
    else:
        class Iface(interface.QPCanonMixin, interface.SolverInterface):
            solver_name = 'OSQP'
            stgs = {}

            def generate_code(self, *a, **k):
                pass
        si = Iface({'n_var': desc.n_var, 'n_eq': desc.n_eq, 'n_ineq': desc.n_ineq}, pp, [])

    # ConstraintInfo (canonicalizer.py:270-281)
    n_data_constr = len(si.indices_constr)
    n_data_constr_vec = si.indptr_constr[-1] - si.indptr_constr[-2]
    n_data_constr_mat = n_data_constr - n_data_constr_vec
    rows_eq = np.nonzero(si.indices_constr < si.n_eq)[0]
    rows_ineq = np.nonzero(si.indices_constr >= si.n_eq)[0]
    ci = mappings.ConstraintInfo(n_data_constr, n_data_constr_mat, rows_eq, rows_ineq)

    # ParameterInfo: column layout of the user parameters (canonicalizer.py:226-271)
    ids = list(range(len(desc.params))) + [len(desc.params)]
    id_to_col = {k: p.col for k, p in enumerate(desc.params)}
    id_to_col[len(desc.params)] = desc.NP
    pinfo = types.SimpleNamespace(num=len(desc.params), ids=ids, id_to_col=id_to_col,
                                  flat_usp=np.asarray(desc.theta0, dtype=float))
    pcanon = mappings.ParameterCanon()
    ids_used = [p for p in si.canon_p_ids if not (p == 'P' and si.indices_obj is None)]
    adjacency = np.zeros((len(si.canon_p_ids), pinfo.num), dtype=bool)
    out = {}
    for i, p_id in enumerate(si.canon_p_ids):
        am = si.get_affine_map(p_id, pp, ci)
        if not am:
            out[p_id] = None
            continue
        if p_id in si.canon_p_ids_constr_vec:
            am = helpers['_update_to_dense_mapping'](am, pp)
        if len(am.mapping.shape) < 2:
            am.mapping = am.mapping.reshape(1, -1)
        am.mapping = am.mapping.tocsr()
        nonzero_d = bool(am.mapping.nnz > 0) if p_id == 'd' else None
        adjacency = helpers['_update_adjacency_matrix'](adjacency, i, pinfo, am.mapping)
        am.mapping = sparse.csc_matrix(am.mapping.toarray() * am.sign)
        am, pcanon = helpers['_set_default_values'](None, am, p_id, pcanon, pinfo, si)
        M = am.mapping.tocsr()
        M.sort_indices()
        out[p_id] = dict(mapping=M, changes=bool(am.mapping[:, :-1].nnz > 0), size=int(am.mapping.shape[0]),
                         nonzero_d=nonzero_d, default=pcanon.p[p_id])
    return out, adjacency, list(si.canon_p_ids)


def emitted_c_canonicalize(utils, mapping, theta, name):
    """compiles what the reference's emitters write for `cpg_canonicalize_<name>` and runs it"""
    import io
    buf = io.StringIO()
    buf.write('#include <stdio.h>\ntypedef double cpg_float;\ntypedef int cpg_int;\n'
              'typedef struct { cpg_int *p; cpg_int *i; cpg_float *x; cpg_int nnz; } cpg_csc;\n')   # utils.py:718-724
    utils.write_mat_def(buf, sp.csr_matrix(mapping), f'canon_{name}_map')
    utils.write_vec_def(buf, np.asarray(theta), 'cpg_params_vec', 'cpg_float')
    rows = mapping.shape[0]
    s = '' if name != 'd' else ''
    if name == 'd':
        buf.write('struct { cpg_float d; } Canon_Params;\n')
    else:
        buf.write(f'cpg_float out_{name}[{rows}];\nstruct {{ cpg_float *{name}; }} Canon_Params = {{ out_{name} }};\n')
    buf.write('int main(void) {\n  cpg_int i, j;\n')
    utils.write_canonicalize(buf, name, s, sp.csr_matrix(mapping), '')
    if name == 'd':
        buf.write('  printf("%.17g\\n", Canon_Params.d);\n')
    else:
        buf.write(f'  for (i = 0; i < {rows}; i++) printf("%.17g\\n", Canon_Params.{name}[i]);\n')
    buf.write('  return 0;\n}\n')
    with tempfile.TemporaryDirectory() as td:
        src, exe = os.path.join(td, 'c.c'), os.path.join(td, 'c')
        open(src, 'w').write(buf.getvalue())
        subprocess.check_call(['gcc', '-O0', '-o', exe, src])
        txt = subprocess.check_output([exe]).decode().split()
    return [float(t) for t in txt]


def main():
    mappings, interface, utils, helpers = _load_reference()
    fams = {'nonneg_ls': families.nonneg_ls(), 'mpc_6_3_10': families.mpc(6, 3, 10), 'toy_box': families.toy_box(),
            'actuator': families.actuator(), 'portfolio_8_3': families.portfolio(8, 3), 'adp_conic': families.adp(), 'toy_lp': families.toy_lp(),
            'nonneg_ls_conic': families.nonneg_ls(solver='CLARABEL')}
    arrays, meta, emitted = {}, {}, {}
    for tag, desc in fams.items():
        out, adjacency, ids = reference_canon(desc, mappings, interface, helpers)
        meta[tag] = {'ids': ids, 'solver': desc.solver, 'changes': {}, 'size': {}, 'nonzero_d': None, 'none': []}
        arrays[f'{tag}/adjacency'] = adjacency
        for p_id, rec in out.items():
            if rec is None:
                meta[tag]['none'].append(p_id)
                continue
            M = rec['mapping']
            arrays[f'{tag}/{p_id}/map_indptr'] = M.indptr.astype(np.int64)
            arrays[f'{tag}/{p_id}/map_indices'] = M.indices.astype(np.int64)
            arrays[f'{tag}/{p_id}/map_data'] = M.data.astype(np.float64)
            arrays[f'{tag}/{p_id}/map_shape'] = np.asarray(M.shape, dtype=np.int64)
            meta[tag]['changes'][p_id] = rec['changes']
            meta[tag]['size'][p_id] = rec['size']
            if rec['nonzero_d'] is not None:
                meta[tag]['nonzero_d'] = rec['nonzero_d']
            dflt = rec['default']
            if sp.issparse(dflt):
                dflt = utils.replace_inf(sp.csc_matrix(dflt))
                arrays[f'{tag}/{p_id}/default_indptr'] = dflt.indptr.astype(np.int64)
                arrays[f'{tag}/{p_id}/default_indices'] = dflt.indices.astype(np.int64)
                arrays[f'{tag}/{p_id}/default_data'] = dflt.data.astype(np.float64)
            else:
                arrays[f'{tag}/{p_id}/default'] = utils.replace_inf(np.array(dflt, dtype=np.float64))
        # what generated C computes for the vector parameters at a second, seeded theta
        rng = np.random.default_rng(11)
        theta = np.asarray(desc.theta0, dtype=float).copy()
        theta[:-1] = theta[:-1] * (1 + 0.3 * rng.standard_normal(desc.NP)) + 0.1 * rng.standard_normal(desc.NP)
        emitted[tag] = {'theta': [float(v) for v in theta]}
        for p_id in ('q', 'd', 'l', 'u', 'b'):
            if out.get(p_id):
                emitted[tag][p_id] = emitted_c_canonicalize(utils, out[p_id]['mapping'], theta, p_id)
    arrays['meta'] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, 'ref_canon_fixtures.npz'), **arrays)
    with open(os.path.join(HERE, 'ref_emitted_c.json'), 'w') as f:
        json.dump(emitted, f)
    # replace_inf known answers (utils.py:213-228)
    v = np.array([1.0, -np.inf, np.inf, 0.0, -3.5])
    ri = {'in': ['1.0', '-inf', 'inf', '0.0', '-3.5'], 'out': [float(t) for t in utils.replace_inf(v.copy())]}
    with open(os.path.join(HERE, 'ref_replace_inf.json'), 'w') as f:
        json.dump(ri, f)
    print('wrote', len(arrays), 'arrays for', list(fams))


if __name__ == '__main__':
    main()
