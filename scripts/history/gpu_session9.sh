#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
echo "== conic gpu tests"; timeout 900 python -m pytest tests/test_conic.py -m gpu -x -q --durations=5 2>&1 | tail -25 | tee $OUT/pytest_gpu9.log
echo "== done"
