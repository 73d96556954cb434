#!/bin/bash
# round 5, session 5: no scalar loads inside the factorisation loop (flags in the destination words, LDS base pinned)
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/r5s5; mkdir -p $OUT; export TMPDIR=/tmp
V=cvxpygen_amd/generated/variants
for v in mpc12_t4 mpc12_t4g600; do
  echo "== $v"; CPG_PROBE_CHECK=$([ $v = mpc12_t4 ] && echo 1 || echo 0) timeout 200 python scripts/gpu_probe_team.py mpc12 $V/$v/libcpg_mpc12.so 20000 2048 2>&1 | tail -14 | tee $OUT/$v.txt
done
C="python $R/bench.py --no-cpu-baseline --no-wall --no-fixed-rho-leg --all-params --batch 20000 --steps 2 --warmup 1 --lib $R/$V/mpc12_t4/libcpg_mpc12.so"
for cn in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 300 rocprofv3 --pmc $cn -d $R/$OUT/pmc_$cn -o pmc -- $C > $R/$OUT/pmc_$cn.log 2>&1 )
  f=$(find $OUT/pmc_$cn -name "*.db" | head -1); [ -n "$f" ] && python scripts/rocpd_pmc.py $f '%osqp%'
  rm -rf $OUT/pmc_$cn
done | tee $OUT/pmc_allparams_t4.txt
echo "== done"
