#!/bin/bash
# Round 2, GPU session 18: entries requested together in the row products of the per-instance factor kernel
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
P="import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2), d['config'].get('mean_iter'), d['config'].get('solved'))"
B="timeout 600 python $R/bench.py --no-cpu-baseline --no-wall --workload portfolio --batch 20000 --steps 3 --warmup 1"
V=$R/cvxpygen_amd/generated/variants
for v in rb4 rb8 rb16; do
echo "== $v"; $B --lib $V/$v/libcpg_portfolio.so 2>&1 | tail -1 | python -c "$P"
echo "== $v max_iter 1"; $B --lib $V/$v/libcpg_portfolio.so --max-iter 1 2>&1 | tail -1 | python -c "$P"
done
