#!/bin/bash
# A/B on ONE box, config 4 (ADP SOCP, 100 000 instances): the generated factorisation of the family library against its table walk
# (CPG_CONIC_FACTOR=0, same library).   gpurun --timeout 900 -- 'CPG_OUT=r6_s7 bash scripts/gpu_ab_conic.sh'
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/${CPG_OUT:-r6_s7}; mkdir -p $OUT; export TMPDIR=/tmp
P="import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3), d['config'].get('mean_iter'), d['config'].get('solved'), d['roofline']['kernel'], round(d['roofline']['kernel_ms'],3))"
B="timeout 600 python $R/bench.py --no-cpu-baseline --no-wall --workload adp"
for rep in 1 2; do
  echo "== table walk $rep"; CPG_CONIC_FACTOR=0 $B 2>&1 | tail -1 | tee $OUT/bench_config4_table_factor_$rep.json | python -c "$P"
  echo "== generated factorisation $rep"; $B 2>&1 | tail -1 | tee $OUT/bench_config4_generated_factor_$rep.json | python -c "$P"
done
echo "== conic GPU tests"; timeout 900 python -m pytest tests/test_conic.py -m gpu -q 2>&1 | tail -3
echo "== done"
