#!/usr/bin/env python
"""Static check of the compiled per-instance kernels of a family library (no GPU needed): compiles cpg_hip.cpp for
gfx950 to assembly with the family's definitions and reports, per kernel,

  * registers / scratch / spills (clang's kernel-resource-usage remarks),
  * scratch accesses INSIDE the ADMM loop of the generated instance kernel (must be 0: the wavefronts' scratch does not
    fit the L2, every reload there is a memory-latency stall in a loop that is a chain of latencies already) and, for
    the streaming kernels, inside the stream loop and in the rest of an iteration,
  * "address-reload sequences": a 64-bit address reloaded from scratch right in front of the global access that uses it
    -- the signature of instance-invariant addresses computed outside the persistent instance loop (HISTORY.md 4.5).

This is how round 3's register-sharing, coefficient-load and termination-test changes were judged before they went to
the GPU (each costs ~2.5 min of hipcc here instead of a GPU session).

    python scripts/isa_hot_loops.py mpc12 | mpc6 | portfolio [--keep out.s]
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from cvxpygen_amd import codegen, families          # noqa: E402
from cvxpygen_amd.runtime import build_family_plan  # noqa: E402

FAMILIES = {'mpc12': lambda: families.mpc(12, 4, 10), 'mpc6': lambda: families.mpc(6, 3, 10),
            'portfolio': lambda: families.portfolio(100, 10)}


def kernels(asm):
    for m in re.finditer(r'^(_Z\d+\w*kernel\w*):', asm, re.M):
        e = asm.find('.end_amdhsa_kernel', m.start())
        if e > 0:
            yield m.group(1), asm[m.start():e].splitlines()


def loops_of(lines):
    labels = {}
    for i, l in enumerate(lines):
        mm = re.match(r'^(\.LBB\d+_\d+):', l)
        if mm:
            labels[mm.group(1)] = i
    out = set()
    for i, l in enumerate(lines):
        mm = re.search(r's_cbranch_\w+\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)', l)
        if mm:
            t = mm.group(1) or mm.group(2)
            if t in labels and labels[t] < i:
                out.add((labels[t], i))
    return out


def stats(lines, a, b):
    seg = [l.strip() for l in lines[a:b + 1]]
    c = lambda p: sum(1 for l in seg if l.startswith(p))
    return dict(n=b - a + 1, fma=c('v_fma_f64') + c('v_fmac_f64'), mul=c('v_mul_f64'), ds_read=c('ds_read'),
                ds_write=c('ds_write'), scratch_load=c('scratch_load'), scratch_store=c('scratch_store'),
                global_load=c('global_load'), dpp=c('v_mov_b32_dpp'))


def address_reloads(lines):
    code = [l.strip() for l in lines if l.strip() and not l.strip().startswith(';')]
    hits = 0
    for i, l in enumerate(code):
        mm = re.match(r'scratch_load_dwordx2 (v\[\d+:\d+\])', l)
        if mm:
            reg = mm.group(1)
            if any(code[j].startswith('global_') and (', ' + reg + ', off') in code[j] for j in range(i + 1, min(i + 8, len(code)))):
                hits += 1
    return hits, sum(1 for l in code if l.startswith('scratch_load')), len(code)


def main():
    fam = sys.argv[1] if len(sys.argv) > 1 else 'mpc12'
    keep = sys.argv[sys.argv.index('--keep') + 1] if '--keep' in sys.argv else None
    plan = build_family_plan(FAMILIES[fam]())
    tmp = tempfile.mkdtemp(prefix='cpg_isa_')
    _, defs = codegen.family_library_defs(plan, tmp, fam)
    out = keep or os.path.join(tmp, fam + '.s')
    src, _ = codegen.source_files()
    r = subprocess.run(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-Wno-unused-value', *defs,
                        '-Rpass-analysis=kernel-resource-usage', '-S', '--cuda-device-only', src, '-o', out],
                       capture_output=True, text=True)
    if r.returncode:
        sys.exit(r.stderr[-3000:])
    for blk in r.stderr.split('remark: Function Name: ')[1:]:
        g = lambda k: re.search(k + r': (\d+)', blk).group(1)
        scratch, occ = g(r'ScratchSize \[bytes/lane\]'), g(r'Occupancy \[waves/SIMD\]')
        print(f'{blk.split(" ")[0][:56]:<56} VGPR {g("VGPRs"):>3} scratch {scratch:>5} B  occupancy {occ}  '
              f'SGPR spills {g("SGPRs Spill")}  VGPR spills {g("VGPRs Spill")}')
    ok, seen = True, False
    for name, lines in kernels(open(out).read()):
        hits, sl, n = address_reloads(lines)
        print(f'{name[:56]:<56} {n:>6} instructions, {sl:>5} scratch loads, {hits:>4} address-reload sequences')
        lps = loops_of(lines)
        if 'osqp_instance_kernel' in name:
            hot = [(a, b) for a, b in sorted(lps, key=lambda x: x[1] - x[0])
                   if stats(lines, a, b)['mul'] >= 30 and stats(lines, a, b)['global_load'] == 0 and b - a < 6000]
            for a, b in hot[:1]:
                st = stats(lines, a, b)
                seen = True
                print('    ADMM loop:', st)
                ok &= st['scratch_load'] == 0 and st['scratch_store'] == 0
        elif 'osqp_refactor' in name:
            stream = [(a, b) for a, b in lps if 1000 < b - a < 2600 and stats(lines, a, b)['fma'] >= 8
                      and stats(lines, a, b)['global_load'] in (4, 8)]
            for sa, sb in sorted(stream)[:1]:
                cont = sorted([(a, b) for a, b in lps if a <= sa and sb <= b and (b - a) > (sb - sa) + 50], key=lambda x: x[1] - x[0])
                inner = stats(lines, sa, sb)
                for a, b in cont[:1]:
                    st = stats(lines, a, b)
                    print(f'    stream loop: {inner["scratch_load"]} scratch loads; rest of an iteration: '
                          f'{st["scratch_load"] - inner["scratch_load"]} loads, {st["scratch_store"] - inner["scratch_store"]} stores')
    if seen:
        print('generated instance kernel: ADMM loop free of scratch accesses' if ok else 'SCRATCH ACCESSES IN THE ADMM LOOP')
    sys.exit(0 if ok else 1)


if __name__ == '__main__':
    main()
