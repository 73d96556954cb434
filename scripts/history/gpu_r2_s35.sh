#!/bin/bash
# Round 2, GPU session 35: eight entries of a row in flight where the row-ordered copy of A is walked (portfolio)
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
P="import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2), d['config'].get('mean_iter'), d['config'].get('solved'))"
B="timeout 600 python $R/bench.py --no-cpu-baseline --no-wall --batch 20000 --steps 3 --warmup 1 --workload portfolio --lib $R/cvxpygen_amd/generated/variants/nb8/libcpg_portfolio.so"
echo "== config 3"; $B 2>&1 | tail -1 | python -c "$P"
echo "== config 3 max_iter 1"; $B --max-iter 1 2>&1 | tail -1 | python -c "$P"
echo "== config 3 200 its no tests"; $B --max-iter 200 --check-termination 1000 2>&1 | tail -1 | python -c "$P"
echo "== config 3 200 its test every 25"; $B --max-iter 200 --eps 1e-12 2>&1 | tail -1 | python -c "$P"
