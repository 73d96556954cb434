#!/bin/bash
# Round 2 final measurement session: GPU tier, bench lines of every BASELINE config, kernel stats and PMC
# passes (SQ, instruction mix, FETCH_SIZE, WRITE_SIZE -- separate passes) of the headline workload.
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/${CPG_OUT:-r2final}; mkdir -p $OUT; export TMPDIR=/tmp
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee $OUT/pytest_gpu.txt
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $OUT/smoke.txt
P="import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2), d['config'].get('mean_iter'), d['config'].get('solved'), d.get('wall_pcie',{}).get('value'), (d.get('cpu_baseline') or {}).get('value'), d.get('adjoint'))"
echo "== config 2 default (cpu baseline, wall)"; timeout 600 python bench.py 2>&1 | tail -1 | tee $OUT/bench_config2.json | python -c "$P"
B="timeout 600 python $R/bench.py --no-cpu-baseline --no-wall"
echo "== config 2 tight eps 1e-6"; $B --eps 1e-6 2>&1 | tail -1 | tee $OUT/bench_config2_tight.json | python -c "$P"
echo "== config 2, OSQP >= 1.0 build options"; $B --osqp1 --steps 2 --warmup 1 2>&1 | tail -1 | tee $OUT/bench_config2_osqp1.json | python -c "$P"
echo "== config 2 at 1M instances"; $B --batch 1000000 --steps 2 --warmup 1 2>&1 | tail -1 | tee $OUT/bench_config2_1M.json | python -c "$P"
echo "== mpc6 (notebook dims)"; $B --workload mpc6 2>&1 | tail -1 | tee $OUT/bench_mpc6.json | python -c "$P"
echo "== generic table-driven (streamed program)"; $B --generic 2>&1 | tail -1 | tee $OUT/bench_config2_generic.json | python -c "$P"
echo "== config 3 portfolio 20k (cpu baseline)"; timeout 900 python bench.py --no-wall --workload portfolio --batch 20000 --steps 3 --warmup 1 2>&1 | tail -1 | tee $OUT/bench_config3_20k.json | python -c "$P"
echo "== config 3 portfolio 125k shard"; $B --workload portfolio --batch 125000 --steps 2 --warmup 1 2>&1 | tail -1 | tee $OUT/bench_config3_125k.json | python -c "$P"
echo "== config 3 with OSQP >= 1.0 options"; $B --workload portfolio --batch 20000 --steps 2 --warmup 1 --osqp1 2>&1 | tail -1 | tee $OUT/bench_config3_osqp1.json | python -c "$P"
echo "== config 4 ADP"; timeout 600 python bench.py --no-wall --workload adp 2>&1 | tail -1 | tee $OUT/bench_config4.json | python -c "$P"
echo "== config 5 adjoint"; $B --adjoint 2>&1 | tail -1 | tee $OUT/bench_config5.json | python -c "$P"
echo "== mpc12 all parameters"; $B --all-params --batch 20000 --steps 3 --warmup 1 2>&1 | tail -1 | tee $OUT/bench_allparams.json | python -c "$P"
cd /tmp
C="python $R/bench.py --no-cpu-baseline --no-wall"
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof -o trace -- $C > $R/$OUT/rocprof.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $R/$OUT/pmc_f -o pmc -- $C --steps 2 --warmup 1 > $R/$OUT/pmc_f.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $R/$OUT/pmc_w -o pmc -- $C --steps 2 --warmup 1 > $R/$OUT/pmc_w.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $R/$OUT/pmc_a -o pmc -- $C --steps 2 --warmup 1 > $R/$OUT/pmc_a.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_ACTIVE_INST_VALU -d $R/$OUT/pmc_b -o pmc -- $C --steps 2 --warmup 1 > $R/$OUT/pmc_b.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof3 -o trace -- $C --workload portfolio --batch 20000 --steps 3 --warmup 1 > $R/$OUT/rocprof3.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE -d $R/$OUT/pmc_f3 -o pmc -- $C --workload portfolio --batch 20000 --steps 2 --warmup 1 > $R/$OUT/pmc_f3.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE -d $R/$OUT/pmc_w3 -o pmc -- $C --workload portfolio --batch 20000 --steps 2 --warmup 1 > $R/$OUT/pmc_w3.log 2>&1
cd $R
f=$(find $OUT/prof -name "*.db" | head -1); [ -n "$f" ] && python scripts/rocpd_summary.py $f | tee $OUT/kernel_stats_config2.txt
f=$(find $OUT/prof3 -name "*.db" | head -1); [ -n "$f" ] && python scripts/rocpd_summary.py $f | tee $OUT/kernel_stats_config3.txt
for d in f w a b; do f=$(find $OUT/pmc_$d -name "*.db" | head -1); [ -n "$f" ] && python scripts/rocpd_pmc.py $f | cut -c62-; done | tee $OUT/pmc_config2.txt
for d in f3 w3; do f=$(find $OUT/pmc_$d -name "*.db" | head -1); [ -n "$f" ] && python scripts/rocpd_pmc.py $f '%refactor%' | cut -c62-; done | tee $OUT/pmc_config3.txt
rm -rf $OUT/prof $OUT/prof3 $OUT/pmc_f $OUT/pmc_w $OUT/pmc_a $OUT/pmc_b $OUT/pmc_f3 $OUT/pmc_w3
echo "== done"
