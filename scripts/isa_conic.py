#!/usr/bin/env python
"""Static look at the conic interior-point kernel of a family library (no GPU): compiles cpg_hip.cpp for gfx950 to assembly with
line tables, prints the resources of the specialised kernel (`clarabel_kernel<true, true>`) and its instructions by kind and by
source file (the kernel is one inlined function: the line tables say which header an instruction came from).

    python scripts/isa_conic.py [out.s] [extra hipcc flags, e.g. -DCPG_CONIC_TWICE=1]
"""
import collections, os, re, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cvxpygen_amd import codegen, families              # noqa: E402
from cvxpygen_amd.conic_plan import build_conic_plan    # noqa: E402

args = sys.argv[1:]
out = args.pop(0) if args and not args[0].startswith('-') else '/tmp/conic_adp.s'
cp = build_conic_plan(families.adp())
gen = tempfile.mkdtemp(prefix='isa_conic_')
hdr = codegen.conic_header(cp, gen, 'adp')
src, _ = codegen.source_files()
defs = ['-DCPG_KERNELS(X)=X(1, 1, 1, 1)', '-DCPG_KERNELS_LDS(Y)=Y(1, 1, 1, 1, 8)', '-DCPG_KERNELS_REFACTOR(Z)=Z(1, 1)', f'-DCPG_GENC_HEADER="{hdr}"']
t = time.time()
log = subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-S', '--cuda-device-only', '-gline-tables-only',
                      '-Wno-unused-value', '-Rpass-analysis=kernel-resource-usage', src, *defs, *args, '-o', out], capture_output=True, text=True)
print(f'hipcc {time.time() - t:.0f} s, rc {log.returncode}')
if log.returncode:
    print(log.stderr[-3000:]); sys.exit(1)
blk = log.stderr.split('Function Name: _Z15clarabel_kernelILb1ELb1E')[1].split('Function Name:')[0]
print('clarabel_kernel<true, true>:', ', '.join(m.group(1) for m in re.finditer(r'remark:\s+((?:VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill): \d+)', blk)))
s = open(out).read().split('\n')
files = {}
for l in s:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m:
        files[int(m.group(1))] = (m.group(3) or m.group(2)).split('/')[-1]
st = next(i for i, l in enumerate(s) if l.startswith('_Z15clarabel_kernelILb1ELb1E'))
en = next(i for i in range(st, len(s)) if s[i].strip().startswith('.Lfunc_end'))
kinds, byfile, cur = collections.Counter(), collections.Counter(), '?'
for l in s[st:en]:
    tk = l.strip()
    m = re.match(r'\.loc\s+(\d+)\s+(\d+)', tk)
    if m:
        cur = files.get(int(m.group(1)), m.group(1)); continue
    if not tk or tk.startswith(('.', ';')) or tk.endswith(':'):
        continue
    op = tk.split()[0]
    k = ('LDS' if op.startswith('ds_') else 'memory' if op.startswith(('global_', 'scratch_', 'buffer_', 'flat_')) else
         'lane moves (v_readlane / v_writelane: SGPR spills and reductions)' if ('readlane' in op or 'writelane' in op) else
         'DPP' if 'dpp' in op else 'branch' if 'branch' in op else 'scalar' if op.startswith('s_') else 'vector f64' if 'f64' in op else 'vector')
    kinds[k] += 1; byfile[cur] += 1
print('static instructions:', sum(kinds.values()))
for k, v in kinds.most_common():
    print(f'  {k:70s} {v:6d}')
print('by source file:', ', '.join(f'{k} {v}' for k, v in byfile.most_common(8)))
