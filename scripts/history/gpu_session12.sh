#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
run() { echo "== $*"; timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), d['unit'], round(d['roofline']['kernel_ms'],2),'ms', d['config']['library'])"; }
run --ipw 1 --waves 8
run --ipw 1 --waves 7
run --ipw 1 --waves 6
run --ipw 2 --waves 4
run --ipw 2 --waves 3
python scripts/pcie_rate.py 2>&1 | tail -1 | tee $OUT/s12_pcie.txt
echo "== done"
