#!/bin/bash
# Round 2, GPU session 26: gathers of the next phase that do not read the open phase's results issued before it closes
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
P="import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2), d['config'].get('mean_iter'), d['config'].get('solved'), d.get('check'))"
B="timeout 600 python $R/bench.py --no-cpu-baseline --no-wall --check"
for v in e4 ef2 ef4 ef8; do echo "== $v"; $B --lib $R/cvxpygen_amd/generated/variants/$v/libcpg_mpc12.so 2>&1 | tail -1 | python -c "$P"; done
