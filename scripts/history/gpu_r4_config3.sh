#!/bin/bash
# Round 4: the record of config 3 on the resident per-instance factor kernel -- GPU tests, bench lines (20 000 / 125 000 instances,
# both forks, the streaming kernel beside it), rocprofv3 kernel stats, FETCH / WRITE passes and the stamped traffic record
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/${CPG_OUT:-r4c3}; mkdir -p $OUT; export TMPDIR=/tmp
P="import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2), d['config'].get('mean_iter'), d['config'].get('solved'), d.get('check'), d['roofline']['kernel'], d['roofline'].get('traffic'))"
echo "== gpu tests (resident, config 3)"; timeout 600 python -m pytest tests/test_resident.py tests/test_gpu_parity.py -m gpu -q -x -k "resident or portfolio or config3" 2>&1 | tail -3 | tee $OUT/pytest_gpu_config3.txt
B="timeout 400 python $R/bench.py --no-wall --no-fixed-rho-leg --workload portfolio"
echo "== 20000 (default mode, check, cpu baseline)"; $B --batch 20000 --steps 3 --warmup 1 --check 2>&1 | tail -1 | tee $OUT/bench_config3_20k.json | python -c "$P"
echo "== 125000 (config 3's shard)"; $B --batch 125000 --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | tee $OUT/bench_config3_125k.json | python -c "$P"
echo "== 20000 fixed rho"; $B --batch 20000 --steps 3 --warmup 1 --no-cpu-baseline --fixed-rho 2>&1 | tail -1 | tee $OUT/bench_config3_20k_fixed_rho.json | python -c "$P"
echo "== 20000 streaming kernel (placement 0)"; $B --batch 20000 --steps 2 --warmup 1 --no-cpu-baseline --placement 0 2>&1 | tail -1 | tee $OUT/bench_config3_20k_streaming.json | python -c "$P"
cd /tmp
C3="python $R/bench.py --no-cpu-baseline --no-wall --no-fixed-rho-leg --workload portfolio --batch 20000 --steps 2 --warmup 1"
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof -o trace -- $C3 > $R/$OUT/rocprof.log 2>&1
cd $R; f=$(find $OUT/prof -name "*.db" | head -1); [ -n "$f" ] && python scripts/rocpd_summary.py $f | tee $OUT/kernel_stats_config3.txt; rm -rf $OUT/prof
cd /tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $R/$OUT/pmc_f3 -o pmc -- $C3 > $R/$OUT/pmc_f3.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $R/$OUT/pmc_w3 -o pmc -- $C3 > $R/$OUT/pmc_w3.log 2>&1
cd $R
for d in f3 w3; do f=$(find $OUT/pmc_$d -name "*.db" | head -1); [ -n "$f" ] && python scripts/rocpd_pmc.py $f '%resident%'; done | tee $OUT/pmc_config3.txt
python scripts/record_traffic.py portfolio 20000 $OUT/pmc_config3.txt "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on python bench.py --no-cpu-baseline --no-wall --no-fixed-rho-leg --workload portfolio --batch 20000 --steps 2 --warmup 1, session $OUT" && cp profiles/hbm_traffic.json $OUT/hbm_traffic.json
rm -rf $OUT/pmc_f3 $OUT/pmc_w3
echo "== bench line with the stamped traffic"; $B --batch 20000 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | tee $OUT/bench_config3_20k_traffic.json | python -c "$P"
echo "== done"
