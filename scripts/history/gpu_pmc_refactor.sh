#!/bin/bash
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY -d $R/$OUT/pmc_rf_a -o pmc -- python $R/bench.py --workload portfolio --batch 20000 --steps 1 --warmup 1 --no-cpu-baseline > $R/$OUT/rf_a.log 2>&1
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum -d $R/$OUT/pmc_rf_b -o pmc -- python $R/bench.py --workload portfolio --batch 20000 --steps 1 --warmup 1 --no-cpu-baseline > $R/$OUT/rf_b.log 2>&1
cd $R
for d in a b; do f=$(find $OUT/pmc_rf_$d -name "*.db" | head -1); python scripts/rocpd_pmc.py $f '%refactor%' | cut -c62-; done | tee $OUT/rf_pmc.txt
tail -2 $OUT/rf_b.log
