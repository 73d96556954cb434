#!/bin/bash
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/${CPG_OUT:-r3s11}; mkdir -p $OUT; export TMPDIR=/tmp
P="import sys,json; d=json.loads(sys.stdin.read()); ph=d.get('phases') or {}; print(round(d['ms_per_step'],2), {k:(round(v['ms'],2), v['instances']) for k,v in ph.items()})"
B="timeout 600 python $R/bench.py --no-cpu-baseline --no-wall --no-fixed-rho-leg --steps 3 --warmup 1"
for st in 5 6 7 3; do
  echo "== mpc12 generated debug_stage=$st"; $B --debug-stage $st 2>&1 | tail -1 | tee $OUT/bench_generated_st$st.json | python -c "$P"
done
echo "== done"
