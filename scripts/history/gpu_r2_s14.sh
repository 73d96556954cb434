#!/bin/bash
# Round 2, GPU session 14: ADMM segment of the per-instance factor kernel as a real call (own register allocation);
# headline kernel with the new generated executor (default build)
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/r2s14; mkdir -p $OUT; export TMPDIR=/tmp
P="import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2), d['config'].get('mean_iter'), d['config'].get('solved'))"
echo "== pytest subset"; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "mpc_vs_oracle or full_size or generated_family or infeasible or nonneg or portfolio_config3 or refactor_path" 2>&1 | tail -3
B="timeout 600 python $R/bench.py --no-cpu-baseline --no-wall"
echo "== config 2"; $B 2>&1 | tail -1 | tee $OUT/bench_config2.json | python -c "$P"
echo "== mpc6"; $B --workload mpc6 2>&1 | tail -1 | tee $OUT/bench_mpc6.json | python -c "$P"
echo "== config 3 20k"; $B --workload portfolio --batch 20000 --steps 3 --warmup 1 2>&1 | tail -1 | tee $OUT/bench_config3.json | python -c "$P"
echo "== all params"; $B --all-params --batch 20000 --steps 3 --warmup 1 2>&1 | tail -1 | tee $OUT/bench_allparams.json | python -c "$P"
echo "== osqp1"; $B --osqp1 --batch 20000 --steps 3 --warmup 1 2>&1 | tail -1 | tee $OUT/bench_osqp1.json | python -c "$P"
echo "== config 3 125k"; $B --workload portfolio --batch 125000 --steps 2 --warmup 1 2>&1 | tail -1 | tee $OUT/bench_config3_125k.json | python -c "$P"
