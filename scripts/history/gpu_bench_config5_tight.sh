#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
echo "== config 5 adjoint"; timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --adjoint 2>&1 | tail -1 | tee $OUT/s23_adjoint.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), d['adjoint'])"
echo "== tight eps 1e-6"; timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --eps 1e-6 2>&1 | tail -1 | tee $OUT/s23_tight.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), d['roofline']['kernel_ms'], d['config']['mean_iter'], d['config']['solved'], d['config']['not_solved'])"
echo "== done"
