#!/bin/bash
# Round 3, session 1: OSQP >= 1.0 default mode through the hybrid (two-kernel) execution: GPU tier, bench lines,
# kernel stats of the default bench.
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/${CPG_OUT:-r3s2}; mkdir -p $OUT; export TMPDIR=/tmp
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -15 | tee $OUT/pytest_gpu.txt
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $OUT/smoke.txt
P="import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2), d['config'].get('mean_iter'), d['config'].get('solved'), d.get('phases'), (d.get('fixed_rho') or {}).get('value'), d.get('wall_pcie',{}).get('value'), (d.get('cpu_baseline') or {}).get('value'), d.get('check'))"
echo "== config 2 default mode (cpu baseline, wall, fixed-rho leg, oracle check)"; timeout 900 python bench.py --check 2>&1 | tail -1 | tee $OUT/bench_config2.json | python -c "$P"
B="timeout 600 python $R/bench.py --no-cpu-baseline --no-wall --no-fixed-rho-leg"
echo "== mpc6 default mode"; $B --workload mpc6 2>&1 | tail -1 | tee $OUT/bench_mpc6.json | python -c "$P"
echo "== config 3 portfolio 20k default mode"; $B --workload portfolio --batch 20000 --steps 3 --warmup 1 2>&1 | tail -1 | tee $OUT/bench_config3_20k.json | python -c "$P"
cd /tmp
C="python $R/bench.py --no-cpu-baseline --no-wall --no-fixed-rho-leg"
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof -o trace -- $C > $R/$OUT/rocprof.log 2>&1
cd $R
f=$(find $OUT/prof -name "*.db" | head -1); [ -n "$f" ] && python scripts/rocpd_summary.py $f | tee $OUT/kernel_stats_config2.txt
rm -rf $OUT/prof
echo "== done"
