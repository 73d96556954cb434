#!/bin/bash
# Round 3, session 10: pipelined factorisation schedule + literal row programs in the generated instance kernel
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/${CPG_OUT:-r3s10}; mkdir -p $OUT; export TMPDIR=/tmp
P="import sys,json; d=json.loads(sys.stdin.read()); ph=d.get('phases') or {}; print(round(d['value']), round(d['ms_per_step'],2), {k:(round(v['ms'],2), v['instances']) for k,v in ph.items()}, d.get('check'))"
B="timeout 600 python $R/bench.py --no-cpu-baseline --no-wall --no-fixed-rho-leg --steps 3 --warmup 1"
for ex in generated stream; do
  echo "== mpc12 executor=$ex debug_stage=3"; $B --instance-executor $ex --debug-stage 3 2>&1 | tail -1 | tee $OUT/bench_${ex}_st3.json | python -c "$P"
  echo "== mpc12 executor=$ex"; $B --instance-executor $ex --check 2>&1 | tail -1 | tee $OUT/bench_${ex}.json | python -c "$P"
  echo "== mpc6 executor=$ex"; $B --workload mpc6 --instance-executor $ex 2>&1 | tail -1 | tee $OUT/bench_mpc6_${ex}.json | python -c "$P"
done
echo "== done"
