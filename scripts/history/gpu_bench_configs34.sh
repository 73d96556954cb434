#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
echo "== portfolio"; timeout 600 python bench.py --workload portfolio --batch 20000 --steps 2 --warmup 1 2>&1 | tail -1 | tee $OUT/s22_portfolio.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), d['unit'], d.get('cpu_baseline'))"
echo "== adp"; timeout 600 python bench.py --workload adp --steps 3 --warmup 1 2>&1 | tail -1 | tee $OUT/s22_adp.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), d['unit'], d.get('cpu_baseline'))"
echo "== done"
