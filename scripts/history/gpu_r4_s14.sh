#!/bin/bash
# Round 4, step 14: per-slot step sizes as per-lane values in resident_iterate (no select between stack addresses in the loop)
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/${CPG_OUT:-r4s14}; mkdir -p $OUT; export TMPDIR=/tmp
P="import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2), d['config'].get('mean_iter'), d['config'].get('solved'), d['roofline']['kernel'], d.get('check'))"
echo "== probes"; CPG_PROBE_STAGE=20 timeout 300 python scripts/gpu_probe_resident.py 20000 2>&1 | tail -9 | tee $OUT/probe_resident.txt
B="timeout 400 python $R/bench.py --no-cpu-baseline --no-wall --no-fixed-rho-leg --workload portfolio"
echo "== config 3 20k"; $B --batch 20000 --steps 3 --warmup 1 --check 2>&1 | tail -1 | tee $OUT/bench_config3_20k.json | python -c "$P"
echo "== config 3 125k"; $B --batch 125000 --steps 2 --warmup 1 2>&1 | tail -1 | tee $OUT/bench_config3_125k.json | python -c "$P"
echo "== gpu tests config 3"; timeout 600 python -m pytest tests/test_resident.py tests/test_gpu_parity.py -m gpu -q -x -k "resident or portfolio or config3" 2>&1 | tail -3 | tee $OUT/pytest_gpu_config3.txt
echo "== done"
