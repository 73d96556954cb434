#!/bin/bash
# PMC counters of the per-instance refactorisation kernel (config 3: portfolio), separate passes.
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
CMD="python $R/bench.py --workload portfolio --batch 20000 --steps 2 --warmup 1 --no-cpu-baseline"
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $R/$OUT/pmc_ref_a -o pmc -- $CMD > $R/$OUT/ref_a.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_ACTIVE_INST_VALU -d $R/$OUT/pmc_ref_b -o pmc -- $CMD > $R/$OUT/ref_b.log 2>&1
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TCC_HIT_sum TCC_MISS_sum -d $R/$OUT/pmc_ref_c -o pmc -- $CMD > $R/$OUT/ref_c.log 2>&1
cd $R
for d in a b c; do f=$(find $OUT/pmc_ref_$d -name "*.db" | head -1); [ -n "$f" ] && python scripts/rocpd_pmc.py $f '%refactor%' | cut -c62-; done | tee $OUT/refactor_portfolio_pmc.txt
tail -2 $OUT/ref_c.log | head -1 | cut -c1-300
