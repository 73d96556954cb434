#!/bin/bash
# Round 2, GPU session 28: what the termination tests cost in the end-of-round headline kernel (100 iterations, fixed)
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
P="import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2), d['config'].get('mean_iter'), d['config'].get('solved'))"
B="timeout 600 python $R/bench.py --no-cpu-baseline --no-wall"
echo "== 100 its, one test at the end"; $B --max-iter 100 --check-termination 1000 --eps 1e-12 2>&1 | tail -1 | python -c "$P"
echo "== 100 its, test every 25"; $B --max-iter 100 --eps 1e-12 2>&1 | tail -1 | python -c "$P"
echo "== 100 its, test every 5"; $B --max-iter 100 --check-termination 5 --eps 1e-12 2>&1 | tail -1 | python -c "$P"
echo "== 200 its, one test at the end"; $B --max-iter 200 --check-termination 1000 --eps 1e-12 2>&1 | tail -1 | python -c "$P"
echo "== max_iter 1"; $B --max-iter 1 2>&1 | tail -1 | python -c "$P"
