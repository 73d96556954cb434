#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
run() { echo "== $*"; timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), d['unit'], round(d['roofline']['kernel_ms'],2),'ms', d['config']['library'], d['config']['mean_iter'])"; }
X=cvxpygen_amd/generated/mpc12g/libcpg_mpc12g.so
run --lib $X --waves 8
run --lib $X --waves 12
run --lib $X --waves 10
run --lib $X --ipw 2 --waves 4
run --lib $X --ipw 2 --waves 8
run --lib $X --ipw 2 --waves 6
echo "== done"
