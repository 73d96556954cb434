#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
echo "== portfolio parity"; timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k portfolio 2>&1 | tail -8 | tee $OUT/pytest_gpu8.log
echo "== bench portfolio"; timeout 600 python bench.py --workload portfolio --batch 20000 --steps 2 --warmup 1 2>&1 | tail -1 | tee $OUT/s8_portfolio.json
echo "== done"
