#!/bin/bash
# Round 3, session 4: where the per-instance factor phase spends its time -- fixed part (canonicalise, numeric LDL',
# coefficient load, final test) against per-iteration part, generated instance executor against the streaming one.
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/${CPG_OUT:-r3s4}; mkdir -p $OUT; export TMPDIR=/tmp
P="import sys,json; d=json.loads(sys.stdin.read()); ph=d.get('phases') or {}; print(round(d['value']), round(d['ms_per_step'],2), d['config'].get('mean_iter'), {k:(round(v['ms'],2), v['instances']) for k,v in ph.items()})"
B="timeout 600 python $R/bench.py --no-cpu-baseline --no-wall --no-fixed-rho-leg --steps 3 --warmup 1"
for ex in generated stream; do for mi in 51 52 76 101; do
  echo "== mpc12 executor=$ex max_iter=$mi"; $B --instance-executor $ex --max-iter $mi 2>&1 | tail -1 | tee $OUT/bench_${ex}_mi$mi.json | python -c "$P"
done; done
echo "== done"
