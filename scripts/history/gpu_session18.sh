#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
run() { echo "== $*"; timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), d['unit'], round(d['roofline']['kernel_ms'],2),'ms', d['config']['mean_iter'])"; }
for T in k1d1 k1d2 k1d3 k2d2; do
X=cvxpygen_amd/generated/mpc12$T/libcpg_mpc12$T.so
run --lib $X
run --lib $X --ipw 2 --waves 4
done
echo "== done"
