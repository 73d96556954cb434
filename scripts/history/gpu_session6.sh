#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
echo "== pytest -m gpu"; timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee $OUT/pytest_gpu6.log
echo "== all-params bench (refactor path)"
for wl in mpc6 mpc12; do
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --all-params --batch 20000 --workload $wl --generic 2>&1 | tail -1 | tee $OUT/s6_allparams_$wl.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), 'inst/s', round(d['roofline']['kernel_ms'],2),'ms', d['config']['mean_iter'], d['config']['solved'])" 2>&1
done
echo "== done"
