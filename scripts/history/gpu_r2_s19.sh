#!/bin/bash
# Round 2, GPU session 19: reduce-store bank conflicts in the slot-numbering cost (ds_write_b64: 16-lane groups, slots collide mod 16)
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
P="import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2), d['config'].get('mean_iter'), d['config'].get('solved'), d['config']['plan'].get('bank_conflict_cycles'))"
B="timeout 600 python $R/bench.py --no-cpu-baseline --no-wall"
V=$R/cvxpygen_amd/generated/variants
for g in 60 600; do echo "== sweeps $g"; CPG_BANK_SWEEPS=$g $B --lib $V/st$g/libcpg_mpc12.so 2>&1 | tail -1 | python -c "$P"; done
