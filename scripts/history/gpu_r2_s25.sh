#!/bin/bash
# Round 2, GPU session 25: pipeline parameters of the generated executor once more, on the end-of-round build
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
P="import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2), d['config'].get('mean_iter'), d['config'].get('solved'))"
B="timeout 600 python $R/bench.py --no-cpu-baseline --no-wall"
for v in nb2 nd3 nc8 nc2; do echo "== $v"; $B --lib $R/cvxpygen_amd/generated/variants/$v/libcpg_mpc12.so 2>&1 | tail -1 | python -c "$P"; done
