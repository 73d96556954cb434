#!/bin/bash
# Round 3, session 5: generated instance executor with the LDS-resident one-step-per-level LDL': breakdown, bench, GPU tier
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/${CPG_OUT:-r3s5}; mkdir -p $OUT; export TMPDIR=/tmp
P="import sys,json; d=json.loads(sys.stdin.read()); ph=d.get('phases') or {}; print(round(d['value']), round(d['ms_per_step'],2), d['config'].get('mean_iter'), {k:(round(v['ms'],2), v['instances']) for k,v in ph.items()}, (d.get('fixed_rho') or {}).get('value'), d.get('check'))"
B="timeout 600 python $R/bench.py --no-cpu-baseline --no-wall --no-fixed-rho-leg --steps 3 --warmup 1"
for mi in 51 76; do
  echo "== mpc12 generated max_iter=$mi"; $B --max-iter $mi 2>&1 | tail -1 | tee $OUT/bench_generated_mi$mi.json | python -c "$P"
done
echo "== config 2 default (check)"; timeout 900 python bench.py --check --no-cpu-baseline 2>&1 | tail -1 | tee $OUT/bench_config2.json | python -c "$P"
echo "== mpc6"; $B --workload mpc6 2>&1 | tail -1 | tee $OUT/bench_mpc6.json | python -c "$P"
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -8 | tee $OUT/pytest_gpu.txt
echo "== done"
