#!/bin/bash
# Round 2, GPU session 8: per-call opaque lane ids in the cold code (check, finalize): spill traffic of the headline kernel
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/r2s8; mkdir -p $OUT; export TMPDIR=/tmp
P="import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2), d['config'].get('mean_iter'), d['config'].get('solved'))"
echo "== pytest subset"; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "mpc_vs_oracle or full_size or generated_family or infeasible or nonneg or portfolio_config3 or refactor_path" 2>&1 | tail -3
B="timeout 600 python $R/bench.py --no-cpu-baseline --no-wall"
echo "== config 2"; $B 2>&1 | tail -1 | tee $OUT/bench_config2.json | python -c "$P"
echo "== mpc6"; $B --workload mpc6 2>&1 | tail -1 | python -c "$P"
echo "== config 3 20k"; $B --workload portfolio --batch 20000 --steps 3 --warmup 1 2>&1 | tail -1 | tee $OUT/bench_config3.json | python -c "$P"
echo "== all params"; $B --all-params --batch 20000 --steps 3 --warmup 1 2>&1 | tail -1 | python -c "$P"
echo "== osqp1"; $B --osqp1 --batch 20000 --steps 3 --warmup 1 2>&1 | tail -1 | python -c "$P"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof -o trace -- $B > $R/$OUT/rocprof.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $R/$OUT/pmc_f -o pmc -- $B --steps 2 --warmup 1 > $R/$OUT/pmc_f.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $R/$OUT/pmc_w -o pmc -- $B --steps 2 --warmup 1 > $R/$OUT/pmc_w.log 2>&1
cd $R
f=$(find $OUT/prof -name "*.db" | head -1); [ -n "$f" ] && python scripts/rocpd_summary.py $f | tee $OUT/kernel_stats.txt
for d in f w; do f=$(find $OUT/pmc_$d -name "*.db" | head -1); [ -n "$f" ] && python scripts/rocpd_pmc.py $f | cut -c62-; done | tee $OUT/pmc.txt
rm -rf $OUT/prof $OUT/pmc_f $OUT/pmc_w
