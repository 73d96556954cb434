#!/bin/bash
# Round 4, step 6: where the per-instance phase of config 2 spends its time (probes inside osqp_instance_kernel), the bench
# lines of configs 2 and 3 on the current sources, the GPU tests of what changed since step 5
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/${CPG_OUT:-r4s6}; mkdir -p $OUT; export TMPDIR=/tmp
P="import sys,json; d=json.loads(sys.stdin.read()); ph=d.get('phases') or {}; print(round(d['value']), round(d['ms_per_step'],2), d['config'].get('mean_iter'), d['config'].get('solved'), d['roofline']['kernel'], {k:(round(v['ms'],2), v['instances']) for k,v in ph.items()})"
echo "== probes"; timeout 300 python scripts/gpu_probe_instance.py 100000 2>&1 | tail -12 | tee $OUT/probe_instance.txt
B="timeout 400 python $R/bench.py --no-cpu-baseline --no-wall --no-fixed-rho-leg"
echo "== config 2"; $B 2>&1 | tail -1 | tee $OUT/bench_config2.json | python -c "$P"
echo "== config 3 20k"; $B --workload portfolio --batch 20000 --steps 3 --warmup 1 2>&1 | tail -1 | tee $OUT/bench_config3_20k.json | python -c "$P"
echo "== gpu tests touched"; timeout 600 python -m pytest tests/test_resident.py tests/test_sequential.py tests/test_gpu_surface.py -m gpu -q -x 2>&1 | tail -3 | tee $OUT/pytest_gpu_subset.txt
echo "== done"
