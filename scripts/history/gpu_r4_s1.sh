#!/bin/bash
# Round 4, session 1: first run of the resident per-instance factor kernel (cpg_osqp_resident.h) on config 3
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/${CPG_OUT:-r4s1}; mkdir -p $OUT; export TMPDIR=/tmp
P="import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2), d['config'].get('mean_iter'), d['config'].get('solved'), d.get('check'), d.get('roofline',{}).get('kernel'))"
B="timeout 300 python $R/bench.py --no-cpu-baseline --no-wall --no-fixed-rho-leg --workload portfolio"
echo "== resident, 20000, check"; $B --batch 20000 --steps 2 --warmup 1 --check 2>&1 | tail -3 | tee $OUT/bench_pf_res.json | tail -1 | python -c "$P"
echo "== streaming (placement 0), 20000"; $B --batch 20000 --steps 2 --warmup 1 --placement 0 2>&1 | tail -1 | tee $OUT/bench_pf_stream.json | python -c "$P"
echo "== resident, 125000"; $B --batch 125000 --steps 2 --warmup 1 2>&1 | tail -1 | tee $OUT/bench_pf_res125.json | python -c "$P"
echo "== resident, 3 waves"; $B --batch 20000 --steps 2 --warmup 1 --waves 3 2>&1 | tail -1 | tee $OUT/bench_pf_res_w3.json | python -c "$P"
echo "== resident, fixed rho"; $B --batch 20000 --steps 2 --warmup 1 --fixed-rho 2>&1 | tail -1 | tee $OUT/bench_pf_res_fixed.json | python -c "$P"
for mi in 1 25 100; do echo "== max_iter $mi"; $B --batch 20000 --steps 2 --warmup 1 --max-iter $mi 2>&1 | tail -1 | python -c "$P"; done
echo "== done"
