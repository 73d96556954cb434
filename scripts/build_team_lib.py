"""Experiment helper: a minimal family library with the team kernel (and the streaming per-instance kernel beside it) under
cvxpygen_amd/generated/variants/<tag>/ for `bench.py --lib` / scripts/gpu_probe_team.py.
    CPG_TEAM_WAVES=4 [CPG_TEAM_MAX_GROUP_ROWS=64] python scripts/build_team_lib.py mpc12|portfolio|mpc6 <tag> [-DFLAG ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cvxpygen_amd import codegen, families
from cvxpygen_amd.runtime import build_family_plan

FAMS = {'mpc12': lambda: families.mpc(12, 4, 10), 'mpc6': lambda: families.mpc(6, 3, 10), 'portfolio': lambda: families.portfolio(100, 10)}
fam, tag = sys.argv[1], sys.argv[2]
d = FAMS[fam]()
plan = build_family_plan(d)
out = os.path.join(ROOT, 'cvxpygen_amd', 'generated', 'variants', tag)
os.makedirs(out, exist_ok=True)
th = codegen.team_header(plan, out, fam)
assert th, 'no team header: CPG_TEAM_WAVES unset or the program does not fit'
nsx, nsz = -(-d.n_var // 64), -(-d.m // 64)
defs = ['-DCPG_KERNELS(X)=', '-DCPG_KERNELS_LDS(Y)=', f'-DCPG_KERNELS_REFACTOR(Z)=Z({nsx}, {nsz})', f'-DCPG_GENT_HEADER="{th}"',
        '-DCPG_REFACTOR_WAVES_PER_SIMD=' + ('4' if nsx + nsz <= 14 else '2')] + sys.argv[3:]
lib = os.path.join(out, f'libcpg_{fam}.so')
src, deps = codegen.source_files()
print(codegen.compile_if_stale(codegen._hipcc_cmd(src, defs, [], lib), lib, [th] + deps, verbose=False))
