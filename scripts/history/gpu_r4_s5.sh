#!/bin/bash
# Round 4, session 5: iteration function inlined into the kernel vs as a call; cost of a call
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/${CPG_OUT:-r4s5}; mkdir -p $OUT; export TMPDIR=/tmp
P="import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2), d['config'].get('mean_iter'), d['config'].get('solved'), d.get('check'))"
for v in "" "_noinl"; do
L=$R/cvxpygen_amd/generated/portfolio/libcpg_portfolio$v.so
B="timeout 300 python $R/bench.py --no-cpu-baseline --no-wall --no-fixed-rho-leg --workload portfolio --batch 20000 --steps 2 --warmup 1 --lib $L"
echo "==== variant '$v'"
echo "== default + check"; $B --check 2>&1 | tail -1 | python -c "$P"
echo "== 100 iterations, one test"; $B --max-iter 100 --check-termination 100 --fixed-rho 2>&1 | tail -1 | python -c "$P"
echo "== 100 iterations, one test, one call per iteration"; $B --max-iter 100 --check-termination 100 --fixed-rho --debug-stage 7 2>&1 | tail -1 | python -c "$P"
echo "== 100 iterations, tests every 5"; $B --max-iter 100 --check-termination 5 --fixed-rho 2>&1 | tail -1 | python -c "$P"
done
echo "== done"
