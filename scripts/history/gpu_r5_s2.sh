#!/bin/bash
# round 5, session 2: team kernel with the batched factorisation (steps per batch 8 / 4, group limit 64 / 600, W = 4 / 8), MFMA micro-benchmark
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/r5s2; mkdir -p $OUT
V=cvxpygen_amd/generated/variants
echo "== mfma micro-benchmark"; (cd scripts/micro && timeout 100 out/mfma_shared out/mfma_program.bin 2>&1 | tail -8) | tee $OUT/mfma_shared.txt
for v in mpc12_t4 mpc12_t4g600 mpc12_t4s4 mpc12_t8; do
  echo "== $v"; timeout 200 python scripts/gpu_probe_team.py mpc12 $V/$v/libcpg_mpc12.so 20000 2048 2>&1 | tail -14 | tee $OUT/$v.txt
done
echo "== done"
