#!/bin/bash
# Round 3, session 6: one-step-per-level LDL' in both per-instance kernels (LDS in the generated one, global in the
# streaming one), generated kernel as one 8-wave workgroup per CU: breakdown of both, then the default bench
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/${CPG_OUT:-r3s6}; mkdir -p $OUT; export TMPDIR=/tmp
P="import sys,json; d=json.loads(sys.stdin.read()); ph=d.get('phases') or {}; print(round(d['value']), round(d['ms_per_step'],2), d['config'].get('mean_iter'), {k:(round(v['ms'],2), v['instances']) for k,v in ph.items()}, (d.get('fixed_rho') or {}).get('value'), d.get('check'))"
B="timeout 600 python $R/bench.py --no-cpu-baseline --no-wall --no-fixed-rho-leg --steps 3 --warmup 1"
for ex in generated stream; do for mi in 51 76; do
  echo "== mpc12 executor=$ex max_iter=$mi"; $B --instance-executor $ex --max-iter $mi 2>&1 | tail -1 | tee $OUT/bench_${ex}_mi$mi.json | python -c "$P"
done; done
for ex in generated stream; do
  echo "== mpc6 executor=$ex"; $B --workload mpc6 --instance-executor $ex 2>&1 | tail -1 | tee $OUT/bench_mpc6_${ex}.json | python -c "$P"
done
echo "== config 3 portfolio 20k"; $B --workload portfolio --batch 20000 2>&1 | tail -1 | tee $OUT/bench_config3_20k.json | python -c "$P"
echo "== done"
