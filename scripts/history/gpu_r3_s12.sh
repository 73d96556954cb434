#!/bin/bash
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/${CPG_OUT:-r3s12}; mkdir -p $OUT; export TMPDIR=/tmp
P="import sys,json; d=json.loads(sys.stdin.read()); ph=d.get('phases') or {}; print(round(d['value']), round(d['ms_per_step'],2), {k:(round(v['ms'],2), v['instances']) for k,v in ph.items()}, d.get('check'))"
B="timeout 600 python $R/bench.py --no-cpu-baseline --no-wall --no-fixed-rho-leg --steps 3 --warmup 1"
echo "== mpc12 generated debug_stage=3"; $B --debug-stage 3 2>&1 | tail -1 | tee $OUT/bench_generated_st3.json | python -c "$P"
echo "== mpc12 generated"; $B --check 2>&1 | tail -1 | tee $OUT/bench_generated.json | python -c "$P"
echo "== mpc6 generated"; $B --workload mpc6 2>&1 | tail -1 | tee $OUT/bench_mpc6_generated.json | python -c "$P"
echo "== done"
