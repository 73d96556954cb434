"""Per-loop instruction statistics of a gfx950 assembly file (hipcc -S --cuda-device-only): scratch accesses,
AGPR moves, LDS operations, waits.  Usage: python scripts/isa_loops.py file.s [min_instructions]"""
import re, sys
from collections import Counter
lines = open(sys.argv[1]).read().split('\n')
mn = int(sys.argv[2]) if len(sys.argv) > 2 else 200
labels = {}
for i, l in enumerate(lines):
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
        labels[m.group(1)] = i
func = None
for i, l in enumerate(lines):
    m = re.match(r'^(_Z\w+):', l)
    if m:
        func = m.group(1)
        if len(sys.argv) > 3 and sys.argv[3] in func:
            print('FUNCTION', func[:90], 'at line', i)
    m = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        a = labels[m.group(1)]
        ins = [x.strip().split()[0] for x in lines[a:i + 1] if x.startswith('\t') and not x.strip().startswith(('.', ';'))]
        if len(ins) < mn:
            continue
        c = Counter(ins)
        g = lambda p: sum(v for k, v in c.items() if k.startswith(p))
        print(f'{func[:48]} loop@{a}-{i}: {len(ins)} instr, scratch ld/st {g("scratch_load")}/{g("scratch_store")}, acc rd/wr '
              f'{c.get("v_accvgpr_read_b32", 0)}/{c.get("v_accvgpr_write_b32", 0)}, ds {g("ds_")}, global {g("global_")}, '
              f'f64 {g("v_fma_f64") + g("v_mul_f64") + g("v_add_f64") + g("v_fmac_f64")}, dpp {sum(1 for x in lines[a:i+1] if "dpp" in x)}, waitcnt {c.get("s_waitcnt", 0)}')
