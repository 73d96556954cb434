#!/bin/bash
# Round 4, session 4: resident kernel after the latency work on the termination test (scaling vectors in LDS, deeper product stream)
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/${CPG_OUT:-r4s4}; mkdir -p $OUT; export TMPDIR=/tmp
P="import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2), d['config'].get('mean_iter'), d['config'].get('solved'), d.get('check'))"
B="timeout 300 python $R/bench.py --no-cpu-baseline --no-wall --no-fixed-rho-leg --workload portfolio --batch 20000 --steps 2 --warmup 1"
echo "== default + check"; $B --check > $OUT/bench_pf_default.log 2>&1; tail -1 $OUT/bench_pf_default.log | python -c "$P" || tail -5 $OUT/bench_pf_default.log
for st in 1 2 3 4 5; do echo "== stop after stage $st (max_iter 1)"; $B --max-iter 1 --debug-stage $st 2>&1 | tail -1 | python -c "$P"; done
echo "== 100 iterations, one test"; $B --max-iter 100 --check-termination 100 --fixed-rho 2>&1 | tail -1 | python -c "$P"
echo "== 100 iterations, tests every 5"; $B --max-iter 100 --check-termination 5 --fixed-rho 2>&1 | tail -1 | python -c "$P"
echo "== 125000"; timeout 300 python $R/bench.py --no-cpu-baseline --no-wall --no-fixed-rho-leg --workload portfolio --batch 125000 --steps 2 --warmup 1 2>&1 | tail -1 | tee $OUT/bench_pf_125k.json | python -c "$P"
echo "== done"
