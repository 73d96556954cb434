#!/bin/bash
# Round 2, GPU session 11: knock-outs of the LDS traffic of the generated executor (fixed work: 100 iterations,
# no termination test; results are wrong by construction)
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/r2s11; mkdir -p $OUT; export TMPDIR=/tmp
P="import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2), d['config'].get('mean_iter'), d['config'].get('solved'))"
B="timeout 600 python $R/bench.py --no-cpu-baseline --no-wall --max-iter 100 --check-termination 0"
for v in ko_conf; do
  echo "== $v"; $B --lib $R/cvxpygen_amd/generated/variants/$v/libcpg_mpc12.so 2>&1 | tail -1 | tee $OUT/bench_$v.json | python -c "$P"
done
