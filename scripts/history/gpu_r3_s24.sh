#!/bin/bash
# Round 3, session 24: per-instance program of the generated instance executor with fewer reduction stages (planner's stage
# cost 0.7: 165 steps / 56 stages in 63 register pairs) against the current one (86 steps / 135 stages in 55); the
# environment selects the variant, the library in the tree was built for it
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/${CPG_OUT:-r3s24}; mkdir -p $OUT; export TMPDIR=/tmp
P="import sys,json; d=json.loads(sys.stdin.read()); ph=d.get('phases') or {}; print(round(d['value']), round(d['ms_per_step'],2), d['config'].get('mean_iter'), d['config'].get('solved'), {k:(round(v['ms'],2), v['instances']) for k,v in ph.items()}, d.get('check'))"
B="timeout 600 python $R/bench.py --no-cpu-baseline --no-wall --no-fixed-rho-leg --steps 5 --warmup 2"
echo "== variant: stage scale ${CPG_STREAM_STAGE_SCALE:-default}"
echo "== config 2 default"; $B --check 2>&1 | tail -1 | tee $OUT/bench_config2_${CPG_STREAM_STAGE_SCALE:-default}.json | python -c "$P"
echo "== config 2 tight eps"; $B --eps 1e-6 2>&1 | tail -1 | tee $OUT/bench_config2_tight_${CPG_STREAM_STAGE_SCALE:-default}.json | python -c "$P"
echo "== max_iter 51"; $B --max-iter 51 2>&1 | tail -1 | python -c "$P"
echo "== done"
