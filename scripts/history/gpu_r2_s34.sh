#!/bin/bash
# Round 2, GPU session 34: per-instance factor kernel with a row-ordered copy of A's entries (row walks of the
# equilibration sweeps and of the termination test without the entry-number indirection)
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
P="import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2), d['config'].get('mean_iter'), d['config'].get('solved'))"
B="timeout 600 python $R/bench.py --no-cpu-baseline --no-wall --batch 20000 --steps 3 --warmup 1"
echo "== config 3"; $B --workload portfolio 2>&1 | tail -1 | python -c "$P"
echo "== config 3 max_iter 1"; $B --workload portfolio --max-iter 1 2>&1 | tail -1 | python -c "$P"
echo "== config 3 200 its no tests"; $B --workload portfolio --max-iter 200 --check-termination 1000 2>&1 | tail -1 | python -c "$P"
echo "== config 3 200 its test every 25"; $B --workload portfolio --max-iter 200 --eps 1e-12 2>&1 | tail -1 | python -c "$P"
echo "== all params"; $B --all-params 2>&1 | tail -1 | python -c "$P"
echo "== osqp1"; $B --osqp1 2>&1 | tail -1 | python -c "$P"
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_surface.py -q -m gpu -x -k "refactor or portfolio or actuator or adjoint or row_class or adaptive" 2>&1 | tail -2
