#!/bin/bash
# Round 2, GPU session 3: full GPU tier on the final headline kernel, default bench with CPU baseline and wall rate,
# kernel stats and HBM-side traffic (FETCH_SIZE / WRITE_SIZE passes) for the default (plain, 8 waves) and the
# dictionary-compressed (12 waves) builds.
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/r2s3; mkdir -p $OUT; export TMPDIR=/tmp
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 2>&1 | tail -16 | tee $OUT/pytest_gpu.txt
echo "== default bench (cpu baseline, wall)"; timeout 600 python bench.py 2>&1 | tail -1 | tee $OUT/bench_default.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d.get('wall_pcie',{}).get('value'), d.get('cpu_baseline'))"
B="timeout 300 python $R/bench.py --no-cpu-baseline --no-wall"
echo "== comp12"; $B --lib $R/cvxpygen_amd/generated/exp/libcpg_mpc12_comp.so 2>&1 | tail -1 | tee $OUT/bench_comp12.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
echo "== mpc6"; $B --workload mpc6 2>&1 | tail -1 | tee $OUT/bench_mpc6.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
echo "== tight eps 1e-6"; $B --eps 1e-6 2>&1 | tail -1 | tee $OUT/bench_tight.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['mean_iter'])"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof -o trace -- python $R/bench.py --no-cpu-baseline --no-wall > $R/$OUT/rocprof.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $R/$OUT/pmc_f -o pmc -- $B --steps 2 --warmup 1 > $R/$OUT/pmc_f.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $R/$OUT/pmc_w -o pmc -- $B --steps 2 --warmup 1 > $R/$OUT/pmc_w.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $R/$OUT/pmc_a -o pmc -- $B --steps 2 --warmup 1 > $R/$OUT/pmc_a.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_ACTIVE_INST_VALU -d $R/$OUT/pmc_b -o pmc -- $B --steps 2 --warmup 1 > $R/$OUT/pmc_b.log 2>&1
cd $R
f=$(find $OUT/prof -name "*.db" | head -1); [ -n "$f" ] && python scripts/rocpd_summary.py $f | tee $OUT/kernel_stats.txt
for d in f w a b; do f=$(find $OUT/pmc_$d -name "*.db" | head -1); [ -n "$f" ] && python scripts/rocpd_pmc.py $f | cut -c62-; done | tee $OUT/pmc.txt
rm -rf $OUT/prof $OUT/pmc_f $OUT/pmc_w $OUT/pmc_a $OUT/pmc_b
echo "== done"
