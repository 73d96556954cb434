"""Experiment helper: builds variants of a family library (generated-executor options) side by side under
cvxpygen_amd/generated/variants/<tag>/ for `bench.py --lib`.  Usage:
    python scripts/build_variants.py mpc12 tag:opt=val,opt=val ..."""
import os, sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cvxpygen_amd import codegen, families
from cvxpygen_amd.runtime import build_family_plan

FAMS = {'mpc12': lambda: families.mpc(12, 4, 10), 'mpc6': lambda: families.mpc(6, 3, 10),
        'portfolio': lambda: families.portfolio(100, 10)}


def main():
    fam = sys.argv[1]
    plan = build_family_plan(FAMS[fam]())

    def one(spec):
        tag, _, opts = spec.partition(':')
        kw = {}
        for item in filter(None, opts.split(',')):
            k, v = item.split('=')
            kw[k] = int(v) if v.lstrip('-').isdigit() else v
        waves = kw.pop('waves', None)
        flags = [f'-D{k}={kw.pop(k)}' for k in list(kw) if k.startswith('CPG_')]
        out = os.path.join(ROOT, 'cvxpygen_amd', 'generated', 'variants', tag)
        return codegen.build_family_library(plan, out, fam, g_list=(1,), waves={1: waves or 8}, verbose=False, extra_flags=flags, **kw)
    with ThreadPoolExecutor(max_workers=6) as ex:
        for p in ex.map(one, sys.argv[2:]):
            print(p)


if __name__ == '__main__':
    main()
