#!/bin/bash
# Round 2, GPU session 5: per-instance factor path with the GENERATED substitution executor (config 3)
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/r2s5; mkdir -p $OUT; export TMPDIR=/tmp
B="timeout 600 python $R/bench.py --no-cpu-baseline --no-wall --workload portfolio --batch 20000 --steps 3 --warmup 1"
P="import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['mean_iter'], d['config']['solved'])"
echo "== table-driven (generic library)"; $B --generic 2>&1 | tail -1 | tee $OUT/bench_generic.json | python -c "$P"
for v in w3p6 w3p10 w2p16 w2p24; do
  echo "== generated $v"; $B --lib $R/cvxpygen_amd/generated/exp_$v/libcpg_portfolio.so 2>&1 | tail -1 | tee $OUT/bench_$v.json | python -c "$P"
done
echo "== parity (family library, both rho modes)"; timeout 900 python -m pytest tests/test_gpu_surface.py -m gpu -x -q -k "portfolio_family_library or config3" 2>&1 | tail -3 | tee $OUT/pytest.txt
echo "== done"
