"""Micro-benchmark generator (round 4): the PER-INSTANCE substitution program of a family with matrix parameters
(portfolio: 172 steps, 144 coefficient registers after sharing) as a straight-line executor whose coefficients live in
registers -- at ONE wavefront per SIMD (512 VGPRs + AGPRs).  Writes the executor header and its LDS tables for
scripts/micro/resident_exec.hip."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from cvxpygen_amd import families, runtime, refactor_plan as rpl, resident_plan as rsp, codegen
from cvxpygen_amd.solve_program import execution_steps, GEN_DUMMY_SLOTS

out = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'out')
fam = sys.argv[1] if len(sys.argv) > 1 else 'portfolio'
ss = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
d = families.portfolio() if fam == 'portfolio' else families.mpc(12, 4, 10)
plan = runtime.build_family_plan(d)
merged = os.environ.get('MICRO_MERGED', '1') != '0'
if merged:
    pl = rsp.build_resident_plan(d.P, d.A, plan.osqp, stage_scale=ss)
    rp, sol = pl.base, pl.sol
else:
    rp = rpl.build_refactor_plan(d.P, d.A, plan.osqp, stage_scale=ss)
    sol = rp.sol
txt = codegen.emit_instance_program(sol, fam, prefetch_next=os.environ.get('MICRO_PREFETCH', '1') != '0',
                                    gather_batch=int(os.environ.get('MICRO_GATHER_BATCH', 0)), lookahead=int(os.environ.get('MICRO_LOOKAHEAD', 12)))
open(os.path.join(out, 'cpg_instance_micro.h'), 'w').write(txt)
steps = execution_steps(sol)
reg_of, shift_of, nregs = codegen.pack_step_registers(sol, steps)
T = len(steps); T4 = (T + 3) & ~3; C4 = (sol.n_chunks + 3) & ~3
zero_off = (sol.n_slots + GEN_DUMMY_SLOTS) * 8
gcols = np.full((T4 // 4, 64, 4), zero_off, dtype=np.uint16)
for t, (_, c, e, cnt) in enumerate(steps):
    sh = shift_of[c]
    gcols[t // 4, sh:sh + cnt, t % 4] = sol.cols[e:e + cnt]
grows = np.full((C4 // 4, 64, 4), sol.n_slots, dtype=np.uint16)
for c in range(sol.n_chunks):
    seg = int(sol.ctab[c, 3]) & 1
    sh = shift_of[c]
    dsh = np.full(64, 0xFFFF, dtype=np.int64)
    dsh[sh:] = sol.desc[c][:64 - sh]
    for g0 in range(0, 64, 16):
        used = set(int(x & 0xFFFF) % 16 for x in dsh[g0:g0 + 16] if (x & 0xFFFF) != 0xFFFF)
        nxt = 0
        for t in range(g0, g0 + 16):
            dd = int(dsh[t]); slot = dd & 0xFFFF
            if slot == 0xFFFF:
                while nxt < GEN_DUMMY_SLOTS and (sol.n_slots + nxt) % 16 in used:
                    nxt += 1
                if nxt < GEN_DUMMY_SLOTS:
                    j = nxt; nxt += 1
                else:
                    j = t & (GEN_DUMMY_SLOTS - 1)
                slot = sol.n_slots + j
            grows[c // 4, t, c % 4] = slot | (((dd >> 28) if seg else 0) << 13)
with open(os.path.join(out, 'micro_tables.h'), 'w') as f:
    f.write(f'#define MICRO_N {rp.n}\n#define MICRO_M {rp.m}\n#define MICRO_NEQ {d.n_eq}\n')
    f.write('static const unsigned short MICRO_GCOLS[] = {' + ','.join(str(int(v)) for v in gcols.ravel()) + '};\n')
    f.write('static const unsigned short MICRO_GROWS[] = {' + ','.join(str(int(v)) for v in grows.ravel()) + '};\n')
print('steps', T, 'chunks', sol.n_chunks, 'phases', sol.n_phases, 'nregs', nregs, 'slots', sol.n_slots, 'n', rp.n, 'm', rp.m)
