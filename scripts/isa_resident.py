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

# compiler-allocated AGPRs (outside the kernel's inline asm) per function: only a0 - a31 may appear in the functions that
# hold named coefficients (a32 - a255) live without a call boundary in between (cpg_wave_gfx950.h)
import re
asm = open(out).read().split('\n')
func, inasm, cnt, low = '', False, {}, {}
for l in asm:
    m = re.match(r'^(_Z\w+):', l)
    if m:
        func = m.group(1)
    if '#ASMSTART' in l:
        inasm = True
    elif '#ASMEND' in l:
        inasm = False
    elif not inasm and 'resident' in func and not l.strip().startswith(';') and re.search(r'[ ,]a\[?\d+', l):
        hi = max(int(x) for x in re.findall(r'[ ,:]a?\[?(\d+)\]?', ' ' + ' '.join(re.findall(r'a\[\d+:\d+\]|a\d+', l))) or [0])
        if hi < 32:
            low[func] = low.get(func, 0) + 1          # (caller-saved a0 - a31: no coefficient lives there)
            continue
        cnt[func] = cnt.get(func, 0) + 1
print('compiler-allocated AGPR operands outside inline asm:')
bad = False
for f in sorted(set(x for x in [re.match(r'^(_Z\w+):', l).group(1) for l in asm if re.match(r'^(_Z\w+):', l)] if 'resident' in x)):
    c = cnt.get(f, 0)
    must = any(k in f for k in ('resident_iterate', 'resident_store_coefficients', 'osqp_resident_kernel'))
    print(f'    {f[:70]:70s} {c}' + (f' (+ {low[f]} on a0 - a31)' if f in low else '') + ('   <-- MUST BE 0' if must and c else ''))
    bad |= bool(must and c)
sys.exit(1 if bad else 0)
