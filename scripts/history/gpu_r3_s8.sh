#!/bin/bash
# Round 3, session 8: where the set-up of the per-instance factor phase goes (debug_stage: the kernel leaves an
# instance after canonicalisation / row classes / factorisation + coefficients)
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/${CPG_OUT:-r3s8}; mkdir -p $OUT; export TMPDIR=/tmp
P="import sys,json; d=json.loads(sys.stdin.read()); ph=d.get('phases') or {}; print(round(d['ms_per_step'],2), {k:(round(v['ms'],2), v['instances']) for k,v in ph.items()})"
B="timeout 600 python $R/bench.py --no-cpu-baseline --no-wall --no-fixed-rho-leg --steps 3 --warmup 1"
for ex in generated stream; do for st in 1 2 3; do
  echo "== mpc12 executor=$ex debug_stage=$st"; $B --instance-executor $ex --debug-stage $st 2>&1 | tail -1 | tee $OUT/bench_${ex}_st$st.json | python -c "$P"
done; done
echo "== done"
