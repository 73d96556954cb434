#!/bin/bash
# Round 3: kernel stats + PMC passes (FETCH_SIZE, WRITE_SIZE, SQ activity, instruction mix -- separate passes) of the
# default-mode config 2 bench; optional $1 = extra bench flags
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/${CPG_OUT:-r3pmc}; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
C="python $R/bench.py --no-cpu-baseline --no-wall --no-fixed-rho-leg ${1:-}"
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof -o trace -- $C > $R/$OUT/rocprof.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $R/$OUT/pmc_f -o pmc -- $C --steps 2 --warmup 1 > $R/$OUT/pmc_f.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $R/$OUT/pmc_w -o pmc -- $C --steps 2 --warmup 1 > $R/$OUT/pmc_w.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $R/$OUT/pmc_a -o pmc -- $C --steps 2 --warmup 1 > $R/$OUT/pmc_a.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_ACTIVE_INST_VALU -d $R/$OUT/pmc_b -o pmc -- $C --steps 2 --warmup 1 > $R/$OUT/pmc_b.log 2>&1
cd $R
f=$(find $OUT/prof -name "*.db" | head -1); [ -n "$f" ] && python scripts/rocpd_summary.py $f | tee $OUT/kernel_stats.txt
for d in f w a b; do f=$(find $OUT/pmc_$d -name "*.db" | head -1); [ -n "$f" ] && python scripts/rocpd_pmc.py $f; done | tee $OUT/pmc.txt
rm -rf $OUT/prof $OUT/pmc_f $OUT/pmc_w $OUT/pmc_a $OUT/pmc_b
echo "== done"
