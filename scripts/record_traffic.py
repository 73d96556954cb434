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
    # the SQ counters of the same session, per kernel: what fraction of its cycles each resource is busy -- the numbers behind
    # bench.py's `roofline.binding` (SQ_BUSY_CYCLES sums the 32 shader engines' busy cycles: x 8 CUs each = CU-cycles of the kernel;
    # quad-cycle counters x 4 / (4 SIMDs per CU) = the same denominator)
    sq = {}
    for ln in open(pmc_file):
        mt = re.match(r'\s*(\S.*?)\s+(SQ_\w+)\s+per-dispatch total\s+([0-9.eE+]+)', ln)
        if mt:
            sq.setdefault(mt.group(1).strip(), {})[mt.group(2)] = float(mt.group(3))
    binding = {}
    for k, c in sq.items():
        cu_cycles = 8.0 * c.get('SQ_BUSY_CYCLES', 0.0)
        if cu_cycles <= 0:
            continue
        b = {'lds_busy': c.get('SQ_LDS_IDX_ACTIVE', 0.0) / cu_cycles,
             'lds_conflict_share': c.get('SQ_LDS_BANK_CONFLICT', 0.0) / max(1.0, c.get('SQ_LDS_IDX_ACTIVE', 0.0)),
             'valu_busy': c.get('SQ_ACTIVE_INST_VALU', 0.0) / cu_cycles,
             'wave_wait_share': c.get('SQ_WAIT_ANY', 0.0) / max(1.0, c.get('SQ_WAVE_CYCLES', 0.0)),
             'waves_per_simd': c.get('SQ_WAVE_CYCLES', 0.0) / cu_cycles if c.get('SQ_WAVE_CYCLES') else None}   # (counted in quad-cycles, four SIMDs per CU)
        fk = per_kernel.get(k, {})
        secs = (c.get('SQ_BUSY_CYCLES', 0.0) / 32.0) / 2.4e9
        b['hbm_busy'] = (2.0 * 1024.0 * fk.get('FETCH_SIZE', 0.0) + 1024.0 * fk.get('WRITE_SIZE', 0.0)) / (secs * 8e12) if secs > 0 else None
        cand = {'lds': b['lds_busy'], 'valu': b['valu_busy'], 'hbm': b['hbm_busy'] or 0.0}
        top = max(cand, key=cand.get)
        # a kernel none of whose units is half busy is bound by the latency chain of its dependent phases: report the wait share
        if cand[top] < 0.5 and b['wave_wait_share'] > cand[top]:
            b.update(resource='latency (wave cycles waiting)', busy=b['wave_wait_share'], useful=1.0 - b['wave_wait_share'])
        else:
            b.update(resource=top, busy=cand[top], useful=cand[top] * (1.0 - b['lds_conflict_share']) if top == 'lds' else cand[top])
        binding[k.split('(')[0][:40]] = {kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in b.items()}
    fetch = sum(2.0 * 1024.0 * v.get('FETCH_SIZE', 0.0) for v in per_kernel.values())
    write = sum(1024.0 * v.get('WRITE_SIZE', 0.0) for v in per_kernel.values())
    path = os.path.join(ROOT, 'profiles', 'hbm_traffic.json')
    rec = json.load(open(path)) if os.path.exists(path) else {
        '_what': 'HBM-side bytes per bench step from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, KiB per dispatch, '
                 'FETCH_SIZE x 2 on gfx950 per MI355X_MICROARCH.md); Infinity-Cache hits are counted; replayed by bench.py only when '
                 'source_fingerprint matches the kernel sources it runs'}
    rec[key] = {'instances': int(instances), 'fetch_bytes': int(fetch), 'write_bytes': int(write),
                'kernel': ' + '.join(sorted(k.split('(')[0][:40] for k in per_kernel)),
                'per_kernel_KiB': per_kernel, 'binding': binding, 'source_fingerprint': source_fingerprint(), 'source': source or pmc_file,
                'recorded': __import__('datetime').datetime.now(__import__('datetime').timezone.utc).strftime('%Y-%m-%d %H:%M UTC')}
    json.dump(rec, open(path, 'w'), indent=1)
    print(key, 'fetch', int(fetch), 'write', int(write), 'fingerprint', rec[key]['source_fingerprint'])


if __name__ == '__main__':
    main(*sys.argv[1:])
