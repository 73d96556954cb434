#!/bin/bash
# Round 3, session 17: time split of the per-instance phase after the register packing (debug stages, max_iter cuts)
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/${CPG_OUT:-r3s17}; mkdir -p $OUT; export TMPDIR=/tmp
P="import sys,json; d=json.loads(sys.stdin.read()); ph=d.get('phases') or {}; print(round(d['value']), round(d['ms_per_step'],2), d['config'].get('mean_iter'), {k:(round(v['ms'],2), v['instances']) for k,v in ph.items()})"
B="timeout 600 python $R/bench.py --no-cpu-baseline --no-wall --no-fixed-rho-leg --steps 5 --warmup 2"
for st in 1 2 3; do echo "== debug_stage=$st"; $B --debug-stage $st 2>&1 | tail -1 | tee $OUT/bench_st$st.json | python -c "$P"; done
for mi in 51 52 75 76; do echo "== max_iter=$mi"; $B --max-iter $mi 2>&1 | tail -1 | tee $OUT/bench_mi$mi.json | python -c "$P"; done
echo "== done"
