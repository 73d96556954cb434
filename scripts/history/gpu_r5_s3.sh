#!/bin/bash
# round 5, session 3: batched factorisation with byte-offset tables (reduction by switch / branch-free, prefetch depth 4 / 8), MFMA micro-benchmark
# with the operand layout found at run time, HBM-side traffic of the team kernel
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/r5s3; mkdir -p $OUT; export TMPDIR=/tmp
V=cvxpygen_amd/generated/variants
echo "== mfma micro-benchmark"; (cd scripts/micro && timeout 100 out/mfma_shared out/mfma_program.bin 2>&1 | tail -8) | tee $OUT/mfma_shared.txt
for v in mpc12_t4 mpc12_t4flat mpc12_t4d8 mpc12_t4g600; do
  echo "== $v"; CPG_PROBE_CHECK=$([ $v = mpc12_t4 ] && echo 1 || echo 0) timeout 200 python scripts/gpu_probe_team.py mpc12 $V/$v/libcpg_mpc12.so 20000 2048 2>&1 | tail -14 | tee $OUT/$v.txt
done
C="python $R/bench.py --no-cpu-baseline --no-wall --no-fixed-rho-leg --all-params --batch 20000 --steps 2 --warmup 1 --lib $R/$V/mpc12_t4/libcpg_mpc12.so"
echo "== bench line"; timeout 200 $C 2>&1 | tail -1 | tee $OUT/bench_allparams_t4.json | cut -c1-400
for cn in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 300 rocprofv3 --pmc $cn -d $R/$OUT/pmc_$cn -o pmc -- $C > $R/$OUT/pmc_$cn.log 2>&1 )
  f=$(find $OUT/pmc_$cn -name "*.db" | head -1); [ -n "$f" ] && python scripts/rocpd_pmc.py $f '%osqp%'
  rm -rf $OUT/pmc_$cn
done | tee $OUT/pmc_allparams_t4.txt
echo "== done"
