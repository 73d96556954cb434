#!/bin/bash
# Round 2, GPU session 17: larger merged phases (fewer phases, more coefficients, fewer resident wavefronts)
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
P="import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2), d['config'].get('mean_iter'), d['config'].get('solved'), d['config']['plan'].get('phases'), d['config']['plan'].get('steps'))"
B="timeout 600 python $R/bench.py --no-cpu-baseline --no-wall"
V=$R/cvxpygen_amd/generated/variants
echo "== 128"; $B 2>&1 | tail -1 | python -c "$P"
for g in 160 192 256; do echo "== $g"; CPG_MAX_GROUP_ROWS=$g $B --lib $V/mg$g/libcpg_mpc12.so 2>&1 | tail -1 | python -c "$P"; done
