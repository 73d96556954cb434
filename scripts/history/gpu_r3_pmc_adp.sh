#!/bin/bash
# Round 3: kernel stats + PMC passes (FETCH_SIZE, WRITE_SIZE, SQ activity, instruction mix -- separate passes) of the
# conic kernel on config 4 (ADP SOCP, family library: generated executor, dimensions compiled in), and the GPU tier of the conic tests
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/${CPG_OUT:-r3pmc_adp}; mkdir -p $OUT; export TMPDIR=/tmp
echo "== gpu tests (conic)"; timeout 900 python -m pytest tests/test_conic.py tests/test_ecos_front.py tests/test_two_stage.py -m gpu -q 2>&1 | tail -15 | tee $OUT/pytest_conic.txt | tail -3
P="import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3), d['config'].get('mean_iter'), (d.get('cpu_baseline') or {}).get('value'))"
echo "== config 4"; timeout 600 python bench.py --no-wall --workload adp 2>&1 | tail -1 | tee $OUT/bench_config4.json | python -c "$P"
echo "== config 4 generic library"; timeout 600 python bench.py --no-wall --no-cpu-baseline --workload adp --generic 2>&1 | tail -1 | tee $OUT/bench_config4_generic.json | python -c "$P"
cd /tmp
C="python $R/bench.py --no-cpu-baseline --no-wall --workload adp"
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof -o trace -- $C > $R/$OUT/rocprof.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $R/$OUT/pmc_f -o pmc -- $C --steps 2 --warmup 1 > $R/$OUT/pmc_f.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $R/$OUT/pmc_w -o pmc -- $C --steps 2 --warmup 1 > $R/$OUT/pmc_w.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $R/$OUT/pmc_a -o pmc -- $C --steps 2 --warmup 1 > $R/$OUT/pmc_a.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_ACTIVE_INST_VALU -d $R/$OUT/pmc_b -o pmc -- $C --steps 2 --warmup 1 > $R/$OUT/pmc_b.log 2>&1
cd $R
f=$(find $OUT/prof -name "*.db" | head -1); [ -n "$f" ] && python scripts/rocpd_summary.py $f | tee $OUT/kernel_stats_config4.txt
for d in f w a b; do f=$(find $OUT/pmc_$d -name "*.db" | head -1); [ -n "$f" ] && python scripts/rocpd_pmc.py $f '%clarabel%'; done | tee $OUT/pmc_config4.txt
rm -rf $OUT/prof $OUT/pmc_f $OUT/pmc_w $OUT/pmc_a $OUT/pmc_b
echo "== done"
