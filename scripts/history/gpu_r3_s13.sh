#!/bin/bash
# Round 3, session 13: one-step-per-level LDL' also for per-instance matrices (config 3, all parameters); GPU tier
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/${CPG_OUT:-r3s13}; mkdir -p $OUT; export TMPDIR=/tmp
P="import sys,json; d=json.loads(sys.stdin.read()); ph=d.get('phases') or {}; print(round(d['value']), round(d['ms_per_step'],2), d['config'].get('mean_iter'), {k:(round(v['ms'],2), v['instances']) for k,v in ph.items()}, d.get('check'))"
B="timeout 600 python $R/bench.py --no-cpu-baseline --no-wall --no-fixed-rho-leg --steps 3 --warmup 1"
echo "== config 2 default"; $B --check 2>&1 | tail -1 | tee $OUT/bench_config2.json | python -c "$P"
echo "== config 3 portfolio 20k default"; $B --workload portfolio --batch 20000 2>&1 | tail -1 | tee $OUT/bench_config3_20k.json | python -c "$P"
echo "== config 3 portfolio 20k fixed rho"; $B --workload portfolio --batch 20000 --fixed-rho 2>&1 | tail -1 | tee $OUT/bench_config3_20k_fixed.json | python -c "$P"
echo "== mpc12 all params 20k default"; $B --all-params --batch 20000 2>&1 | tail -1 | tee $OUT/bench_allparams.json | python -c "$P"
echo "== mpc12 all params 20k fixed rho"; $B --all-params --batch 20000 --fixed-rho 2>&1 | tail -1 | tee $OUT/bench_allparams_fixed.json | python -c "$P"
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -8 | tee $OUT/pytest_gpu.txt
echo "== done"
