#!/bin/bash
# round 5, session 8: the block inverses follow the LDL' chain level by level on wavefronts 1 - 3 (no barrier); partial sums per chunk on / off
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/r5s8; mkdir -p $OUT; export TMPDIR=/tmp
V=cvxpygen_amd/generated/variants
for v in mpc12_t4 mpc12_t4ns; do
  echo "== $v"; CPG_PROBE_CHECK=$([ $v = mpc12_t4 ] && echo 1 || echo 0) timeout 120 python scripts/gpu_probe_team.py mpc12 $V/$v/libcpg_mpc12.so 20000 2048 2>&1 | tail -14 | tee $OUT/$v.txt
done
echo "== done"
