#!/bin/bash
# Round 4, step 8: generated straight-line factorisation of shared-matrix mode (numeric_ldl_gen) in the instance kernel
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/${CPG_OUT:-r4s8}; mkdir -p $OUT; export TMPDIR=/tmp
P="import sys,json; d=json.loads(sys.stdin.read()); ph=d.get('phases') or {}; print(round(d['value']), round(d['ms_per_step'],2), d['config'].get('mean_iter'), d['config'].get('solved'), d['roofline']['kernel'], {k:(round(v['ms'],2), v['instances']) for k,v in ph.items()}, d.get('check'))"
echo "== probes"; timeout 300 python scripts/gpu_probe_instance.py 100000 2>&1 | tail -12 | tee $OUT/probe_instance.txt
B="timeout 400 python $R/bench.py --no-cpu-baseline --no-wall --no-fixed-rho-leg"
echo "== config 2"; $B --check 2>&1 | tail -1 | tee $OUT/bench_config2.json | python -c "$P"
echo "== gpu tests touched"; timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "mpc or hybrid or generated or adaptive" 2>&1 | tail -3 | tee $OUT/pytest_gpu_subset.txt
echo "== done"
