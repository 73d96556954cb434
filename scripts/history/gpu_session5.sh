#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
run() { local label=$1; shift; echo "-- $label: $*"
  timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" 2>&1 | tail -1 | tee $OUT/s5_$label.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), 'inst/s', round(d['roofline']['kernel_ms'],2),'ms', d['config']['mean_iter'], d['config'].get('library'))" 2>&1; }
run gen_g1_w8 --waves 8
run gen_g1_w7 --waves 7
run gen_g2_w4 --waves 4 --ipw 2
run generic --generic
echo "== check"; timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --check 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['check'], d['config']['library'])"
echo "== pmc"
cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OLDPWD/$OUT/pmc_s5 -o pmc -- python $OLDPWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OLDPWD/$OUT/s5_pmc.log 2>&1
cd $OLDPWD; python scripts/rocpd_pmc.py $OUT/pmc_s5/pmc_results.db | cut -c60-; echo "== done"
