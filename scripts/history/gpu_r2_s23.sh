#!/bin/bash
# Round 2, GPU session 23: two instances per wavefront with the end-of-round generated executor
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
P="import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2), d['config'].get('mean_iter'), d['config'].get('solved'))"
B="timeout 600 python $R/bench.py --no-cpu-baseline --no-wall"
echo "== mpc12 ipw1"; $B 2>&1 | tail -1 | python -c "$P"
echo "== mpc12 ipw2"; $B --ipw 2 2>&1 | tail -1 | python -c "$P"
echo "== mpc6 ipw1"; $B --workload mpc6 2>&1 | tail -1 | python -c "$P"
echo "== mpc6 ipw2"; $B --workload mpc6 --ipw 2 2>&1 | tail -1 | python -c "$P"
