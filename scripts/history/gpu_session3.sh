#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
run() { # label, args...
  local label=$1; shift
  echo "-- $label: $*"
  timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" 2>&1 | tail -1 | tee $OUT/s3_$label.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), 'inst/s', round(d['roofline']['kernel_ms'],2),'ms', d['config']['mean_iter'])" 2>&1
}
echo "== pytest -m gpu"; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee $OUT/pytest_gpu3.log
run lds_g1_w8 --placement 1 --waves 8
run lds_g1_w6 --placement 1 --waves 6
run lds_g2_w4 --placement 1 --waves 4 --ipw 2
run lds_g2_w3 --placement 1 --waves 3 --ipw 2
run u8_g1_w8 --placement 1 --waves 8 --lib cvxpygen_amd/csrc/libcpg_hip_u8.so
run u8_g2_w4 --placement 1 --waves 4 --ipw 2 --lib cvxpygen_amd/csrc/libcpg_hip_u8.so
run stream --placement 0
run mpc6_auto --workload mpc6
run mpc6_g2 --workload mpc6 --ipw 2
echo "== pmc for lds_g2_w4"
cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OLDPWD/$OUT/pmc_s3 -o pmc -- python $OLDPWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline --placement 1 --waves 4 --ipw 2 > $OLDPWD/$OUT/s3_pmc.log 2>&1
cd $OLDPWD; echo "== done"
