#!/bin/bash
# round 5, session 9: where does a level of the factorisation chain spend its time?  builds that leave a piece out (results are garbage on purpose)
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/r5s9; mkdir -p $OUT; export TMPDIR=/tmp
V=cvxpygen_amd/generated/variants
echo "== mpc12_t4 (complete)"; CPG_PROBE_CHECK=0 timeout 120 python scripts/gpu_probe_team.py mpc12 $V/mpc12_t4/libcpg_mpc12.so 20000 2048 2>&1 | tail -12 | tee $OUT/mpc12_t4.txt
for e in 1 2 3; do
  echo "== experiment $e (1 no division, 2 no reduction, 3 one operand read per step instead of three)"
  CPG_PROBE_TIMING_ONLY=1 CPG_PROBE_CHECK=0 timeout 120 python scripts/gpu_probe_team.py mpc12 $V/mpc12_x$e/libcpg_mpc12.so 20000 2048 2>&1 | grep -E "factor|setup|store" | tee $OUT/mpc12_x$e.txt
done
echo "== done"
