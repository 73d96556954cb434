#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
echo "== pytest -m gpu"; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee $OUT/pytest_gpu4.log
echo "== default bench"; timeout 600 python bench.py --check 2>&1 | tail -1 | tee $OUT/bench_default4.json | cut -c1-300
echo "== rocprof kernel trace"
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OLDPWD/$OUT/prof_s4 -o bench -- python $OLDPWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OLDPWD/$OUT/s4_rocprof.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OLDPWD/$OUT/pmc_s4 -o pmc -- python $OLDPWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OLDPWD/$OUT/s4_pmc.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $OLDPWD/$OUT/pmc_s4f -o pmc -- python $OLDPWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OLDPWD/$OUT/s4_pmcf.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $OLDPWD/$OUT/pmc_s4w -o pmc -- python $OLDPWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OLDPWD/$OUT/s4_pmcw.log 2>&1
cd $OLDPWD; python scripts/rocpd_summary.py $OUT/prof_s4/bench_results.db | head -4; echo "== done"
