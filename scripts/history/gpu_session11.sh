#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
for occ in 2 4 5 8; do for w in 0 4 8; do
echo "== adp occ=$occ waves=$w"; timeout 300 python bench.py --workload adp --steps 3 --warmup 1 --waves $w --lib cvxpygen_amd/csrc/libcpg_conic_w$occ.so 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), d['unit'], round(d['roofline']['kernel_ms'],2),'ms')"
done; done
echo "== done"
