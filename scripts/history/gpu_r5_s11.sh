#!/bin/bash
# round 5, session 11: non-temporal hints on the team kernel's coefficient images: kernel time and HBM-side traffic, with / without
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/r5s11; mkdir -p $OUT; export TMPDIR=/tmp
V=cvxpygen_amd/generated/variants
for v in mpc12_t4nt mpc12_t4; do
  C="python $R/bench.py --no-cpu-baseline --no-wall --no-fixed-rho-leg --all-params --batch 20000 --steps 2 --warmup 1 --lib $R/$V/$v/libcpg_mpc12.so"
  echo "== $v"; timeout 200 $C 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), d['roofline']['kernel'])"
  for cn in FETCH_SIZE WRITE_SIZE; do
    ( cd /tmp && timeout 300 rocprofv3 --pmc $cn -d $R/$OUT/pmc_$cn -o pmc -- $C > $R/$OUT/pmc_$cn.log 2>&1 )
    f=$(find $OUT/pmc_$cn -name "*.db" | head -1); [ -n "$f" ] && python scripts/rocpd_pmc.py $f '%osqp%'
    rm -rf $OUT/pmc_$cn
  done | tee $OUT/pmc_$v.txt
done
echo "== done"
