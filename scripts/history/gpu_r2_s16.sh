#!/bin/bash
# Round 2, GPU session 16: ADMM segment as a call, waves per SIMD of the per-instance factor kernel
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/r2s16; mkdir -p $OUT; export TMPDIR=/tmp
P="import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2), d['config'].get('mean_iter'), d['config'].get('solved'))"
B="timeout 600 python $R/bench.py --no-cpu-baseline --no-wall --batch 20000 --steps 3 --warmup 1"
V=$R/cvxpygen_amd/generated/variants
echo "== all params w4"; $B --all-params 2>&1 | tail -1 | python -c "$P"
echo "== all params w3"; $B --all-params --lib $V/rw3/libcpg_mpc12.so 2>&1 | tail -1 | python -c "$P"
echo "== all params w2"; $B --all-params --lib $V/rw2/libcpg_mpc12.so 2>&1 | tail -1 | python -c "$P"
echo "== osqp1 w4"; $B --osqp1 2>&1 | tail -1 | python -c "$P"
echo "== osqp1 w3"; $B --osqp1 --lib $V/rw3/libcpg_mpc12.so 2>&1 | tail -1 | python -c "$P"
echo "== osqp1 w2"; $B --osqp1 --lib $V/rw2/libcpg_mpc12.so 2>&1 | tail -1 | python -c "$P"
echo "== config 3 w2"; $B --workload portfolio 2>&1 | tail -1 | python -c "$P"
echo "== config 3 w3"; $B --workload portfolio --lib $V/pw3/libcpg_portfolio.so 2>&1 | tail -1 | python -c "$P"
