#!/bin/bash
# Round 2, GPU session 7: per-instance factor kernel -- resident waves against register spills of the ADMM loop
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/r2s7; mkdir -p $OUT; export TMPDIR=/tmp
P="import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['mean_iter'], d['config']['solved'])"
B="timeout 600 python $R/bench.py --no-cpu-baseline --no-wall --workload portfolio --batch 20000 --steps 3 --warmup 1"
for v in w3 w2 w3qu; do echo "== portfolio $v"; $B --lib $R/cvxpygen_amd/generated/exp_portfolio_$v/libcpg_portfolio.so 2>&1 | tail -1 | tee $OUT/bench_portfolio_$v.json | python -c "$P"; done
B="timeout 600 python $R/bench.py --no-cpu-baseline --no-wall --all-params --batch 20000 --steps 3 --warmup 1"
for v in w4 w3 w2; do echo "== mpc12 all params $v"; $B --lib $R/cvxpygen_amd/generated/exp_mpc12_$v/libcpg_mpc12.so 2>&1 | tail -1 | tee $OUT/bench_allparams_$v.json | python -c "$P"; done
B="timeout 600 python $R/bench.py --no-cpu-baseline --no-wall --osqp1 --batch 20000 --steps 3 --warmup 1"
for v in w4 w3 w2; do echo "== mpc12 osqp1 $v"; $B --lib $R/cvxpygen_amd/generated/exp_mpc12_$v/libcpg_mpc12.so 2>&1 | tail -1 | tee $OUT/bench_osqp1_$v.json | python -c "$P"; done
echo "== done"
