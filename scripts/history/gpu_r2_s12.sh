#!/bin/bash
# Round 2, GPU session 12: generated executor with operand offsets stored for all 64 lanes (no masking of partial
# steps), and two steps' offsets per LDS read
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/r2s12; mkdir -p $OUT; export TMPDIR=/tmp
P="import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2), d['config'].get('mean_iter'), d['config'].get('solved'))"
B="timeout 600 python $R/bench.py --no-cpu-baseline --no-wall"
for v in nopad pad pair pairc8 paird3; do
  echo "== $v"; $B --lib $R/cvxpygen_amd/generated/variants/$v/libcpg_mpc12.so 2>&1 | tail -1 | tee $OUT/bench_$v.json | python -c "$P"
done
echo "== pair --check"; $B --check --lib $R/cvxpygen_amd/generated/variants/pair/libcpg_mpc12.so 2>&1 | tail -3
