#!/bin/bash
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
echo "== bench default (with cpu baseline)"; timeout 600 python bench.py 2>&1 | tail -1 | tee $OUT/s20_bench.json | cut -c1-400
echo "== bench mpc6"; timeout 300 python bench.py --workload mpc6 --no-cpu-baseline --steps 3 2>&1 | tail -1 | tee $OUT/s20_mpc6.json | cut -c1-200
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof_s20 -o trace -- python $R/bench.py --no-cpu-baseline > $R/$OUT/s20_rocprof.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $R/$OUT/pmc_s20a -o pmc -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/$OUT/s20a.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_ACTIVE_INST_VALU -d $R/$OUT/pmc_s20b -o pmc -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/$OUT/s20b.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $R/$OUT/pmc_s20f -o pmc -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/$OUT/s20f.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $R/$OUT/pmc_s20w -o pmc -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/$OUT/s20w.log 2>&1
cd $R
f=$(find $OUT/prof_s20 -name "*.db" | head -1); python scripts/rocpd_summary.py $f | tee $OUT/s20_kernel_stats.txt
for d in a b f w; do f=$(find $OUT/pmc_s20$d -name "*.db" | head -1); python scripts/rocpd_pmc.py $f | cut -c62-; done | tee $OUT/s20_pmc.txt
echo "== done"
