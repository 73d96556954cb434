#!/bin/bash
# Round 2, GPU session 1: full GPU test tier on the restructured kernels, then the headline-kernel variants
# (termination test out of the hot loop + register stash; plain vs dictionary-compressed program; 8 vs 12 waves),
# kernel stats and PMC passes of the default build.
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/r2s1; mkdir -p $OUT; export TMPDIR=/tmp
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $OUT/pytest_gpu.txt
echo "== smoke"; timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3 | tee $OUT/smoke.txt
B="timeout 300 python bench.py --no-cpu-baseline --steps 5 --warmup 2"
echo "== V1 plain, 8 waves";  $B --lib cvxpygen_amd/generated/exp/libcpg_mpc12_plain.so 2>&1 | tail -1 | tee $OUT/bench_plain8.json | cut -c1-260
echo "== V2 compressed, 8 waves"; $B --waves 8 2>&1 | tail -1 | tee $OUT/bench_comp8.json | cut -c1-260
echo "== V2 compressed, 12 waves (default)"; $B 2>&1 | tail -1 | tee $OUT/bench_comp12.json | cut -c1-260
echo "== V2 compressed, 10 waves"; $B --waves 10 2>&1 | tail -1 | tee $OUT/bench_comp10.json | cut -c1-260
echo "== plain G=2 x 4 waves"; $B --lib cvxpygen_amd/generated/exp/libcpg_mpc12_plain.so --ipw 2 2>&1 | tail -1 | tee $OUT/bench_plain_g2.json | cut -c1-260
echo "== check vs oracle (default build)"; timeout 300 python bench.py --no-cpu-baseline --steps 2 --warmup 1 --check 2>&1 | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['check'])" | tee $OUT/check.txt
echo "== mpc6"; $B --workload mpc6 2>&1 | tail -1 | tee $OUT/bench_mpc6.json | cut -c1-200
echo "== portfolio 20k"; $B --workload portfolio --batch 20000 --steps 3 2>&1 | tail -1 | tee $OUT/bench_portfolio.json | cut -c1-260
echo "== mpc12 all params 20k"; $B --all-params --batch 20000 --steps 3 2>&1 | tail -1 | tee $OUT/bench_allparams.json | cut -c1-260
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof -o trace -- python $R/bench.py --no-cpu-baseline > $R/$OUT/rocprof.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $R/$OUT/pmc_a -o pmc -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/$OUT/pmc_a.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_ACTIVE_INST_VALU -d $R/$OUT/pmc_b -o pmc -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/$OUT/pmc_b.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $R/$OUT/pmc_f -o pmc -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/$OUT/pmc_f.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $R/$OUT/pmc_w -o pmc -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/$OUT/pmc_w.log 2>&1
cd $R
f=$(find $OUT/prof -name "*.db" | head -1); [ -n "$f" ] && python scripts/rocpd_summary.py $f | tee $OUT/kernel_stats.txt
for d in a b f w; do f=$(find $OUT/pmc_$d -name "*.db" | head -1); [ -n "$f" ] && python scripts/rocpd_pmc.py $f | cut -c62-; done | tee $OUT/pmc.txt
rm -rf $OUT/prof $OUT/pmc_a $OUT/pmc_b $OUT/pmc_f $OUT/pmc_w
echo "== done"
