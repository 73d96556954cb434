#!/usr/bin/env python
"""Static check of the resident per-instance factor kernel of a family (no GPU): compiles cpg_hip.cpp for gfx950 to
assembly with ONLY that family's resident executor configured (no shared-factor kernels: ~1 min instead of ~3) and
prints the kernel's resources and the statistics of its loops (scripts/isa_loops.py): scratch accesses and AGPR reads
inside the ADMM loop are what to look at.

    python scripts/isa_resident.py portfolio [out.s]
"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cvxpygen_amd import codegen, families          # noqa: E402
from cvxpygen_amd.runtime import build_family_plan  # noqa: E402

fam = sys.argv[1] if len(sys.argv) > 1 else 'portfolio'
out = sys.argv[2] if len(sys.argv) > 2 else f'/tmp/resident_{fam}.s'
d = {'portfolio': lambda: families.portfolio(100, 10)}[fam]()
plan = build_family_plan(d)
gen = os.path.join(ROOT, 'cvxpygen_amd', 'generated', fam)
rh = codegen.resident_header(plan, gen, fam)
nsx, nsz = -(-d.n_var // 64), -(-d.m // 64)
defs = ['-DCPG_KERNELS(X)=', '-DCPG_KERNELS_LDS(Y)=', f'-DCPG_KERNELS_REFACTOR(Z)=Z({nsx}, {nsz})', f'-DCPG_GENR_HEADER="{rh}"',
        '-DCPG_REFACTOR_WAVES_PER_SIMD=2']
src, _ = codegen.source_files()
t = time.time()
log = subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-S', '--cuda-device-only', '-Wno-unused-value',
                      '-Rpass-analysis=kernel-resource-usage', src, *defs, *codegen.resident_compiler_flags(defs), *sys.argv[3:], '-o', out], capture_output=True, text=True)
print(f'hipcc {time.time() - t:.0f} s, rc {log.returncode}')
if log.returncode:
    print(log.stderr[-3000:]); sys.exit(1)
lines = log.stderr.splitlines()
for i, l in enumerate(lines):
    if 'Function Name' in l and 'resident' in l:
        print(l.split('Function Name:')[1].split('[')[0].strip()[:60])
        print('   ', '; '.join(x.split('remark:')[1].split('[-R')[0].strip() for x in lines[i:i + 14]
                               if any(k in x for k in ('VGPRs', 'AGPRs', 'Scratch', 'Spill', 'Occupancy'))))
r = subprocess.run([sys.executable, os.path.join(ROOT, 'scripts', 'isa_loops.py'), out, '150'], capture_output=True, text=True).stdout
rows = [l for l in r.splitlines() if 'resident' in l]
seen = set()
for l in rows:
    key = l.split(':', 1)[1]
    if key in seen:
        continue
    seen.add(key)
    print(l)

# the function that runs the ADMM iterations: its loop must hold (next to) no scratch access
import re
lines = open(out).read().split('\n')
a = next(i for i, l in enumerate(lines) if re.match(r'^_Z\w*resident_iterate\w*:', l))
b = next(i for i in range(a, len(lines)) if lines[i].startswith('.Lfunc_end'))
labels = {m.group(1): i for i in range(a, b) for m in [re.match(r'^(\.LBB\d+_\d+):', lines[i])] if m}
best = None
for i in range(a, b):
    m = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', lines[i])
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        body = [x.strip().split()[0] for x in lines[labels[m.group(1)]:i + 1] if x.startswith('\t') and not x.strip().startswith(('.', ';'))]
        f64 = sum(1 for x in body if x.startswith(('v_fma_f64', 'v_fmac_f64', 'v_mul_f64', 'v_add_f64')))
        if f64 >= 100 and (best is None or len(body) < best[0]):
            best = (len(body), sum(1 for x in body if x.startswith('scratch_load')), sum(1 for x in body if x.startswith('scratch_store')),
                    sum(1 for x in body if x == 'v_accvgpr_read_b32'), sum(1 for x in body if x.startswith('ds_')))
print('ADMM loop of resident_iterate: %d instructions, scratch loads / stores %d / %d, AGPR reads %d, LDS operations %d' % best)
