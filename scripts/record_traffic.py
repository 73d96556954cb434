#!/usr/bin/env python
"""Turns the per-kernel FETCH_SIZE / WRITE_SIZE totals of two rocprofv3 --pmc passes (scripts/rocpd_pmc.py output,
one line per kernel and counter) into a record of profiles/hbm_traffic.json that bench.py replays next to its
live timing: HBM-side bytes per STEP (all solve kernels of one step), stamped with the fingerprint of the kernel
sources the measurement was taken on (bench.source_fingerprint) -- bench.py refuses a record whose stamp differs.

    python scripts/record_traffic.py <key> <instances> <pmc_text_file> [<source note>]

Counters are in KiB per dispatch; FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950 (HBM section)."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(key, instances, pmc_file, source=''):
    from bench import source_fingerprint
    per_kernel = {}
    for ln in open(pmc_file):
        mt = re.match(r'\s*(\S.*?)\s+(FETCH_SIZE|WRITE_SIZE)\s+per-dispatch total\s+([0-9.eE+]+)', ln)
        if mt:
            per_kernel.setdefault(mt.group(1).strip(), {})[mt.group(2)] = float(mt.group(3))
    if not per_kernel:
        raise SystemExit(f'no FETCH_SIZE / WRITE_SIZE lines in {pmc_file}')
    fetch = sum(2.0 * 1024.0 * v.get('FETCH_SIZE', 0.0) for v in per_kernel.values())
    write = sum(1024.0 * v.get('WRITE_SIZE', 0.0) for v in per_kernel.values())
    path = os.path.join(ROOT, 'profiles', 'hbm_traffic.json')
    rec = json.load(open(path)) if os.path.exists(path) else {
        '_what': 'HBM-side bytes per bench step from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, KiB per dispatch, '
                 'FETCH_SIZE x 2 on gfx950 per MI355X_MICROARCH.md); Infinity-Cache hits are counted; replayed by bench.py only when '
                 'source_fingerprint matches the kernel sources it runs'}
    rec[key] = {'instances': int(instances), 'fetch_bytes': int(fetch), 'write_bytes': int(write),
                'kernel': ' + '.join(sorted(k.split('(')[0][:40] for k in per_kernel)),
                'per_kernel_KiB': per_kernel, 'source_fingerprint': source_fingerprint(), 'source': source or pmc_file}
    json.dump(rec, open(path, 'w'), indent=1)
    print(key, 'fetch', int(fetch), 'write', int(write), 'fingerprint', rec[key]['source_fingerprint'])


if __name__ == '__main__':
    main(*sys.argv[1:])
