#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest -m gpu"
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $OUT/pytest_gpu2.log
echo "== sweep LDS-resident"
for waves in 4 6 8; do
  echo "-- placement=1 waves=$waves"
  timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --placement 1 --waves $waves 2>&1 | tail -1 | tee $OUT/s2_lds_w${waves}.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), 'inst/s', round(d['roofline']['kernel_ms'],2),'ms', d['config']['mean_iter'])" 2>&1
done
echo "-- placement=0 (stream)"
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --placement 0 2>&1 | tail -1 | tee $OUT/s2_stream.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), 'inst/s', round(d['roofline']['kernel_ms'],2),'ms')" 2>&1
echo "== mpc6 workload"
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --workload mpc6 2>&1 | tail -1 | tee $OUT/s2_mpc6.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), 'inst/s', round(d['roofline']['kernel_ms'],2),'ms', d['config']['mean_iter'])" 2>&1
echo "== default bench --check"
timeout 600 python bench.py --check 2>&1 | tail -1 | tee $OUT/bench_default2.json | cut -c1-400
echo "== rocprof kernel trace + counters"
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OLDPWD/$OUT/prof_r1b -o bench -- python $OLDPWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OLDPWD/$OUT/rocprof2.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OLDPWD/$OUT/pmc_r1b -o pmc -- python $OLDPWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OLDPWD/$OUT/rocprof2_pmc.log 2>&1
cd $OLDPWD
python scripts/rocpd_summary.py $OUT/prof_r1b/bench_results.db 2>&1 | head -5
ls $OUT/pmc_r1b | head
echo "== done"
