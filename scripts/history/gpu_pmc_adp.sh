#!/bin/bash
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $R/$OUT/pmc_adp_a -o pmc -- python $R/bench.py --workload adp --steps 2 --warmup 1 --no-cpu-baseline > $R/$OUT/adp_a.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_ACTIVE_INST_VALU -d $R/$OUT/pmc_adp_b -o pmc -- python $R/bench.py --workload adp --steps 2 --warmup 1 --no-cpu-baseline > $R/$OUT/adp_b.log 2>&1
cd $R
for d in a b; do f=$(find $OUT/pmc_adp_$d -name "*.db" | head -1); python scripts/rocpd_pmc.py $f '%clarabel%' | cut -c62-; done | tee $OUT/adp_pmc.txt
