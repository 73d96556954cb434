#!/bin/bash
# Evidence for the per-instance refactorisation / adjoint / conic paths after the streaming executor:
# bench lines (with CPU baselines), rocprofv3 kernel-trace stats and PMC passes for config 3.
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
echo "== portfolio (config 3)"; timeout 600 python bench.py --workload portfolio --batch 20000 --steps 3 --warmup 1 2>&1 | tail -1 | tee $OUT/s24_portfolio.json | cut -c1-300
echo "== portfolio 40000"; timeout 600 python bench.py --workload portfolio --batch 40000 --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | tee $OUT/s24_portfolio_40k.json | cut -c1-200
echo "== adp (config 4)"; timeout 600 python bench.py --workload adp --steps 3 --warmup 1 2>&1 | tail -1 | tee $OUT/s24_adp.json | cut -c1-300
echo "== adjoint (config 5)"; timeout 600 python bench.py --adjoint --batch 20000 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | tee $OUT/s24_adjoint.json | cut -c1-200
echo "== mpc12 all parameters"; timeout 600 python bench.py --all-params --batch 20000 --steps 3 --warmup 1 2>&1 | tail -1 | tee $OUT/s24_mpc12_all.json | cut -c1-200
cd /tmp
CMD="python $R/bench.py --workload portfolio --batch 20000 --steps 3 --warmup 1 --no-cpu-baseline"
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof_s24 -o trace -- $CMD > $R/$OUT/s24_rocprof.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $R/$OUT/pmc_s24a -o pmc -- $CMD > $R/$OUT/s24a.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_ACTIVE_INST_VALU -d $R/$OUT/pmc_s24b -o pmc -- $CMD > $R/$OUT/s24b.log 2>&1
timeout 300 rocprofv3 --pmc TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum -d $R/$OUT/pmc_s24c -o pmc -- $CMD > $R/$OUT/s24c.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $R/$OUT/pmc_s24f -o pmc -- $CMD > $R/$OUT/s24f.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $R/$OUT/pmc_s24w -o pmc -- $CMD > $R/$OUT/s24w.log 2>&1
cd $R
f=$(find $OUT/prof_s24 -name "*.db" | head -1); python scripts/rocpd_summary.py $f | tee $OUT/s24_kernel_stats.txt
for d in a b c f w; do f=$(find $OUT/pmc_s24$d -name "*.db" | head -1); [ -n "$f" ] && python scripts/rocpd_pmc.py $f '%refactor%' | cut -c62-; done | tee $OUT/s24_pmc.txt
echo "== done"
