#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
run() { echo "== $*"; timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), d['unit'], round(d['roofline']['kernel_ms'],2),'ms', d['config']['mean_iter'])"; }
run --workload portfolio --batch 20000
run --workload portfolio --batch 20000 --max-iter 25
run --workload portfolio --batch 20000 --max-iter 50
run --workload portfolio --batch 20000 --blocks-per-cu 1
run --workload portfolio --batch 20000 --blocks-per-cu 3
echo "== done"
