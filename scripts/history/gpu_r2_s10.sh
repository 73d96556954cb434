#!/bin/bash
# Round 2, GPU session 10: generated executor variants (coefficient loads across phase boundaries; part of the
# coefficients from global memory instead of LDS), MPC 12/4/10
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/r2s10; mkdir -p $OUT; export TMPDIR=/tmp
P="import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2), d['config'].get('mean_iter'), d['config'].get('solved'))"
B="timeout 600 python $R/bench.py --no-cpu-baseline --no-wall"
for v in base c4 c8 c16 d3c6 g4 g3 g2 g1 g2c6 g2a40; do
  echo "== $v"; $B --lib $R/cvxpygen_amd/generated/variants/$v/libcpg_mpc12.so 2>&1 | tail -1 | tee $OUT/bench_$v.json | python -c "$P"
done
