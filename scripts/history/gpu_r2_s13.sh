#!/bin/bash
# Round 2, GPU session 13: four steps' operand offsets per LDS read; pipeline depth / lookahead / batch variants
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/r2s13; mkdir -p $OUT; export TMPDIR=/tmp
P="import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2), d['config'].get('mean_iter'), d['config'].get('solved'))"
B="timeout 600 python $R/bench.py --no-cpu-baseline --no-wall"
for v in quad quadd3 quadc2 quadc6 quadd3c2 quadd1 quadb2 quadb2d1 quadb2c0 quadb3; do
  echo "== $v"; $B --lib $R/cvxpygen_amd/generated/variants/$v/libcpg_mpc12.so 2>&1 | tail -1 | tee $OUT/bench_$v.json | python -c "$P"
done
