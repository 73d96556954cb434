#!/bin/bash
# Round 4, session 3: where the resident kernel's time goes (stage stops, iteration / test split)
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/${CPG_OUT:-r4s3}; mkdir -p $OUT; export TMPDIR=/tmp
P="import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2), d['config'].get('mean_iter'), d['config'].get('solved'))"
B="timeout 300 python $R/bench.py --no-cpu-baseline --no-wall --no-fixed-rho-leg --workload portfolio --batch 20000 --steps 2 --warmup 1"
for st in 1 2 3 4 5; do echo "== stop after stage $st (max_iter 1)"; $B --max-iter 1 --debug-stage $st 2>&1 | tail -1 | python -c "$P"; done
echo "== 100 iterations, one test"; $B --max-iter 100 --check-termination 100 --fixed-rho 2>&1 | tail -1 | python -c "$P"
echo "== 100 iterations, tests every 25"; $B --max-iter 100 --fixed-rho 2>&1 | tail -1 | python -c "$P"
echo "== 100 iterations, tests every 5"; $B --max-iter 100 --check-termination 5 --fixed-rho 2>&1 | tail -1 | python -c "$P"
echo "== 200 iterations, one test"; $B --max-iter 200 --check-termination 200 --fixed-rho 2>&1 | tail -1 | python -c "$P"
echo "== default"; $B 2>&1 | tail -1 | tee $OUT/bench_pf_default.json | python -c "$P"
echo "== done"
