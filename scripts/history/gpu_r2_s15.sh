#!/bin/bash
# Round 2, GPU session 15: where the time of the per-instance factor kernel goes (config 3, 20 000 instances):
# set-up only (max_iter 1), 200 iterations without / with termination tests
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/r2s15; mkdir -p $OUT; export TMPDIR=/tmp
P="import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2), d['config'].get('mean_iter'), d['config'].get('solved'))"
B="timeout 600 python $R/bench.py --no-cpu-baseline --no-wall --workload portfolio --batch 20000 --steps 3 --warmup 1"
echo "== max_iter 1"; $B --max-iter 1 2>&1 | tail -1 | python -c "$P"
echo "== 200 its, no checks"; $B --max-iter 200 --check-termination 1000 2>&1 | tail -1 | python -c "$P"
echo "== 200 its, check every 25"; $B --max-iter 200 --eps 1e-12 2>&1 | tail -1 | python -c "$P"
echo "== 200 its, check every 5"; $B --max-iter 200 --eps 1e-12 --check-termination 5 2>&1 | tail -1 | python -c "$P"
echo "== 400 its, no checks"; $B --max-iter 400 --check-termination 1000 2>&1 | tail -1 | python -c "$P"
