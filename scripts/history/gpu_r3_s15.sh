#!/bin/bash
# Round 3, session 15: conic kernel (config 4, ADP SOCP): one inlined copy of kkt_solve per loop, generated executor of
# the substitution program, family dimensions compiled in -- each against the previous form, same session
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/${CPG_OUT:-r3s15}; mkdir -p $OUT; export TMPDIR=/tmp
P="import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3), d['config'].get('mean_iter'))"
B="timeout 600 python $R/bench.py --no-cpu-baseline --no-wall --steps 5 --warmup 2 --workload adp"
echo "== gpu tests (conic)"; timeout 900 python -m pytest tests/test_conic.py -m gpu -x -q 2>&1 | tail -40 | tee $OUT/pytest_conic.txt | tail -3
echo "== config 4 family library (generated executor, dimensions compiled in)"; $B 2>&1 | tail -1 | tee $OUT/bench_config4_specialised.json | python -c "$P"
echo "== config 4 family library, generated executor, run-time dimensions"; CPG_CONIC_SPECIALISED=0 $B 2>&1 | tail -1 | tee $OUT/bench_config4_generated.json | python -c "$P"
echo "== config 4 family library, table-driven (tables in global memory)"; CPG_CONIC_GENERATED=0 $B 2>&1 | tail -1 | tee $OUT/bench_config4_tables.json | python -c "$P"
echo "== config 4 generic library"; $B --generic 2>&1 | tail -1 | tee $OUT/bench_config4_generic.json | python -c "$P"
for w in 8 12; do echo "== family library, waves $w"; $B --waves $w 2>&1 | tail -1 | python -c "$P"; done
echo "== done"
