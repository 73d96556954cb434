#!/bin/bash
# Round 2, GPU session 24: output-slot table read four chunks at a time; dummy store targets chosen per 16-lane group
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
P="import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2), d['config'].get('mean_iter'), d['config'].get('solved'))"
B="timeout 600 python $R/bench.py --no-cpu-baseline --no-wall"
echo "== rows quad"; $B --check --lib $R/cvxpygen_amd/generated/variants/rowsq/libcpg_mpc12.so 2>&1 | tail -1 | python -c "$P; print(d.get('check'))"
