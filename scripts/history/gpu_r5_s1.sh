#!/bin/bash
# round 5, session 1: the team kernel's first contact with the GPU -- parity on 48 instances, stage times, kernel time of 20 000
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/r5s1; mkdir -p $OUT
V=cvxpygen_amd/generated/variants
for v in mpc12_t4 mpc12_t8 mpc12_t4g600 mpc12_t4tab; do
  echo "== $v"; timeout 420 python scripts/gpu_probe_team.py mpc12 $V/$v/libcpg_mpc12.so 20000 2048 2>&1 | tail -16 | tee $OUT/$v.txt
done
echo "== portfolio_t4"; timeout 420 python scripts/gpu_probe_team.py portfolio $V/portfolio_t4/libcpg_portfolio.so 20000 2048 2>&1 | tail -16 | tee $OUT/portfolio_t4.txt
echo "== done"
