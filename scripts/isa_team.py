#!/usr/bin/env python
"""Static check of the team per-instance factor kernel of a family (no GPU): compiles cpg_hip.cpp for gfx950 to assembly
with ONLY that family's team executor configured and prints the resources of the kernel and of its stage functions, and the
instruction mix of the ADMM loop of team_iterate (scratch accesses, LDS operations, barriers).

    CPG_TEAM_WAVES=4 python scripts/isa_team.py mpc12|portfolio|mpc6 [out.s] [extra hipcc flags]
"""
import os, re, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cvxpygen_amd import codegen, families          # noqa: E402
from cvxpygen_amd.runtime import build_family_plan  # noqa: E402

fam = sys.argv[1] if len(sys.argv) > 1 else 'mpc12'
out = sys.argv[2] if len(sys.argv) > 2 else f'/tmp/team_{fam}.s'
d = {'portfolio': lambda: families.portfolio(100, 10), 'mpc12': lambda: families.mpc(12, 4, 10), 'mpc6': lambda: families.mpc(6, 3, 10)}[fam]()
plan = build_family_plan(d)
gen = os.path.join('/tmp', f'isa_team_{fam}')
os.makedirs(gen, exist_ok=True)
th = codegen.team_header(plan, gen, fam)
assert th, 'no team header (CPG_TEAM_WAVES unset, or the program does not fit)'
print(open(th).read(400).split('#pragma')[0])
nsx, nsz = -(-d.n_var // 64), -(-d.m // 64)
defs = ['-DCPG_KERNELS(X)=', '-DCPG_KERNELS_LDS(Y)=', f'-DCPG_KERNELS_REFACTOR(Z)=Z({nsx}, {nsz})', f'-DCPG_GENT_HEADER="{th}"',
        '-DCPG_REFACTOR_WAVES_PER_SIMD=2']
src, _ = codegen.source_files()
t = time.time()
log = subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-S', '--cuda-device-only', '-Wno-unused-value',
                      '-Rpass-analysis=kernel-resource-usage', src, *defs, *sys.argv[3:], '-o', out], capture_output=True, text=True)
print(f'hipcc {time.time() - t:.0f} s, rc {log.returncode}')
if log.returncode:
    print(log.stderr[-3000:]); sys.exit(1)
lines = log.stderr.splitlines()
for i, l in enumerate(lines):
    if 'Function Name' in l and 'team' in l:
        print(l.split('Function Name:')[1].split('[')[0].strip()[:70])
        print('   ', '; '.join(x.split('remark:')[1].split('[-R')[0].strip() for x in lines[i:i + 14]
                               if any(k in x for k in ('VGPRs', 'AGPRs', 'Scratch', 'Spill', 'Occupancy', 'LDS'))))
lines = open(out).read().split('\n')
for fn in ('team_iterate', 'team_check', 'team_setup', 'team_factorise', 'team_store_coefficients'):
    try:
        a = next(i for i, l in enumerate(lines) if re.match(r'^_Z\w*' + fn + r'\w*:', l))
    except StopIteration:
        continue
    b = next(i for i in range(a, len(lines)) if lines[i].startswith('.Lfunc_end'))
    body = [x.strip().split()[0] for x in lines[a:b] if x.startswith('\t') and not x.strip().startswith(('.', ';'))]
    print(f'{fn}: {len(body)} instructions, scratch loads / stores {sum(1 for x in body if x.startswith("scratch_load"))} / '
          f'{sum(1 for x in body if x.startswith("scratch_store"))}, barriers {sum(1 for x in body if x == "s_barrier")}')
    if fn != 'team_iterate':
        continue
    labels = {m.group(1): i for i in range(a, b) for m in [re.match(r'^(\.LBB\d+_\d+):', lines[i])] if m}
    best = None
    for i in range(a, b):
        m = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', lines[i])
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            lb = [x.strip().split()[0] for x in lines[labels[m.group(1)]:i + 1] if x.startswith('\t') and not x.strip().startswith(('.', ';'))]
            f64 = sum(1 for x in lb if x.startswith(('v_fma_f64', 'v_fmac_f64', 'v_mul_f64', 'v_add_f64')))
            if f64 >= 50 and (best is None or len(lb) > best[0]):
                best = (len(lb), sum(1 for x in lb if x.startswith('scratch_load')), sum(1 for x in lb if x.startswith('scratch_store')),
                        sum(1 for x in lb if x == 'v_accvgpr_read_b32'), sum(1 for x in lb if x.startswith('ds_')), sum(1 for x in lb if x == 's_barrier'),
                        sum(1 for x in lb if x.startswith('s_waitcnt')))
    if best:
        print('  ADMM loop (all wavefronts\' code): %d instructions, scratch loads / stores %d / %d, AGPR reads %d, LDS operations %d, barriers %d, waits %d' % best)
