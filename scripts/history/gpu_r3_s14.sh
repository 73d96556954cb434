#!/bin/bash
# Round 3, session 14: per-instance-matrix kernel as its own instantiation again (config 3, all parameters), adjoint on
# the pruned factor pattern (config 5), smoke with library verification
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/${CPG_OUT:-r3s14}; mkdir -p $OUT; export TMPDIR=/tmp
P="import sys,json; d=json.loads(sys.stdin.read()); ph=d.get('phases') or {}; print(round(d['value']), round(d['ms_per_step'],2), d['config'].get('mean_iter'), {k:(round(v['ms'],2), v['instances']) for k,v in ph.items()}, d.get('adjoint'), (d.get('cpu_baseline') or {}).get('value'))"
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $OUT/smoke.txt
B="timeout 600 python $R/bench.py --no-cpu-baseline --no-wall --no-fixed-rho-leg --steps 3 --warmup 1"
echo "== config 2 default + adjoint"; $B --adjoint 2>&1 | tail -1 | tee $OUT/bench_config5.json | python -c "$P"
echo "== mpc6"; $B --workload mpc6 2>&1 | tail -1 | tee $OUT/bench_mpc6.json | python -c "$P"
echo "== config 3 portfolio 20k default"; $B --workload portfolio --batch 20000 2>&1 | tail -1 | tee $OUT/bench_config3_20k.json | python -c "$P"
echo "== config 3 portfolio 20k fixed rho"; $B --workload portfolio --batch 20000 --fixed-rho 2>&1 | tail -1 | tee $OUT/bench_config3_20k_fixed.json | python -c "$P"
echo "== mpc12 all params 20k default"; $B --all-params --batch 20000 2>&1 | tail -1 | tee $OUT/bench_allparams.json | python -c "$P"
echo "== config 4 ADP (multi-core cpu baseline)"; timeout 600 python bench.py --no-wall --workload adp 2>&1 | tail -1 | tee $OUT/bench_config4.json | python -c "$P"
echo "== done"
