#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
run() { echo "== $*"; timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), d['unit'], round(d['roofline']['kernel_ms'],2),'ms', d['config']['mean_iter'])"; }
X=cvxpygen_amd/generated/mpc6w/libcpg_mpc6w.so
for w in 8 10 12 14 16; do run --workload mpc6 --lib $X --waves $w; done
echo "== done"
