#!/bin/bash
# Round 3, session 22: config 3 with the streaming executor's entry words in a block-shared LDS copy (eight wavefronts per
# workgroup) against the same library reading them through L2 (--placement 0); FETCH_SIZE of both
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/${CPG_OUT:-r3s22}; mkdir -p $OUT; export TMPDIR=/tmp
P="import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2), d['config'].get('mean_iter'), d['config'].get('solved'))"
B="timeout 600 python $R/bench.py --no-cpu-baseline --no-wall --no-fixed-rho-leg --steps 3 --warmup 1 --workload portfolio"
echo "== config 3 20k, entry words in LDS"; $B --batch 20000 2>&1 | tail -1 | tee $OUT/bench_config3_20k.json | python -c "$P"
echo "== config 3 20k, entry words through L2"; $B --batch 20000 --placement 0 2>&1 | tail -1 | tee $OUT/bench_config3_20k_l2.json | python -c "$P"
echo "== config 3 20k fixed rho, LDS"; $B --batch 20000 --fixed-rho 2>&1 | tail -1 | tee $OUT/bench_config3_20k_fixed.json | python -c "$P"
echo "== config 3 125k, LDS"; $B --batch 125000 --steps 2 2>&1 | tail -1 | tee $OUT/bench_config3_125k.json | python -c "$P"
echo "== gpu tests (portfolio)"; timeout 900 python -m pytest tests -m gpu -q -k "portfolio or config3" 2>&1 | tail -3
cd /tmp
for pl in -1 0; do
C="python $R/bench.py --no-cpu-baseline --no-wall --no-fixed-rho-leg --workload portfolio --batch 20000 --steps 2 --warmup 1 --placement $pl"
timeout 400 rocprofv3 --pmc FETCH_SIZE -d $R/$OUT/pmc_f$pl -o pmc -- $C > $R/$OUT/pmc_f$pl.log 2>&1
f=$(find $R/$OUT/pmc_f$pl -name "*.db" | head -1); [ -n "$f" ] && (cd $R; python scripts/rocpd_pmc.py $f '%refactor%' | tee -a $OUT/pmc_config3.txt)
rm -rf $R/$OUT/pmc_f$pl
done
echo "== done"
