"""Register / scratch report of a family library's kernels (hipcc -Rpass-analysis=kernel-resource-usage) and the
scratch instructions per loop nest of the ISA.
Usage: [GEN_OPTS=cross=4,global_every=2] python scripts/kernel_resources.py mpc12|mpc6|portfolio [out.s]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cvxpygen_amd import codegen, families
from cvxpygen_amd.runtime import build_family_plan

FAMS = {'mpc12': lambda: families.mpc(12, 4, 10), 'mpc6': lambda: families.mpc(6, 3, 10),
        'portfolio': lambda: families.portfolio(100, 10)}


def main(name):
    out = os.path.join(ROOT, 'cvxpygen_amd', 'generated', name)
    kw = {k: (int(v) if v.lstrip('-').isdigit() else v) for k, v in (it.split('=') for it in filter(None, os.environ.get('GEN_OPTS', '').split(',')))}
    if kw:
        out = os.path.join(out, '..', 'variants', 'tmp')
    hdr, defs = codegen.family_library_defs(build_family_plan(FAMS[name]()), out, name, **kw)
    src, _ = codegen.source_files()
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, 'k.s')
        cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-Wno-unused-value', src, *defs,
               '-S', '--cuda-device-only', '-Rpass-analysis=kernel-resource-usage', '-o', asm]
        p = subprocess.run(cmd, capture_output=True, text=True)
        cur = None
        for line in p.stderr.splitlines():
            m = re.search(r'remark:\s+(.*?)\s+\[-Rpass', line)
            if not m:
                continue
            t = m.group(1)
            if t.startswith('Function Name:'):
                cur = t.split(':', 1)[1].strip()[:40]
                print(cur, end=' ')
            elif any(k in t for k in ('VGPRs:', 'AGPRs', 'ScratchSize', 'Occupancy', 'Spill', 'LDS Size')):
                print('|', t, end=' ')
            if 'LDS Size' in t:
                print()
        txt = open(asm).read()
        if len(sys.argv) > 2:
            open(sys.argv[2], 'w').write(txt)
        for kern in re.findall(r'^(_Z\w+):', txt, re.M):
            body = txt[txt.index(kern + ':'):]
            if 's_endpgm' not in body:
                continue
            body = body[:body.index('s_endpgm')]
            st = len(re.findall(r'scratch_store', body)); ld = len(re.findall(r'scratch_load', body))
            print(kern[:40], 'scratch_store', st, 'scratch_load', ld, 'lines', body.count('\n'))


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1 else 'mpc12')
