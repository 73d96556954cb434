#!/bin/bash
# Round 3, session 16: generated instance executor with narrow steps sharing coefficient registers (86 steps in 58
# register pairs on MPC 12/4/10: no coefficient left in scratch inside the ADMM loop)
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/${CPG_OUT:-r3s16}; mkdir -p $OUT; export TMPDIR=/tmp
P="import sys,json; d=json.loads(sys.stdin.read()); ph=d.get('phases') or {}; print(round(d['value']), round(d['ms_per_step'],2), d['config'].get('mean_iter'), d['config'].get('solved'), {k:(round(v['ms'],2), v['instances']) for k,v in ph.items()}, d.get('check'))"
B="timeout 600 python $R/bench.py --no-cpu-baseline --no-wall --no-fixed-rho-leg"
echo "== config 2 default"; $B --check 2>&1 | tail -1 | tee $OUT/bench_config2.json | python -c "$P"
echo "== mpc6"; $B --workload mpc6 --check 2>&1 | tail -1 | tee $OUT/bench_mpc6.json | python -c "$P"
echo "== config 2 tight eps"; $B --eps 1e-6 2>&1 | tail -1 | tee $OUT/bench_config2_tight.json | python -c "$P"
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -5 | tee $OUT/pytest_gpu.txt
echo "== done"
