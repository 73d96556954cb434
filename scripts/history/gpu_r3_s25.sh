#!/bin/bash
# Round 3, session 25: fixed costs of the per-instance phase for the two per-instance programs (see session 24)
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/${CPG_OUT:-r3s25}; mkdir -p $OUT; export TMPDIR=/tmp
P="import sys,json; d=json.loads(sys.stdin.read()); ph=d.get('phases') or {}; print(round(d['ms_per_step'],2), {k:(round(v['ms'],2)) for k,v in ph.items()})"
B="timeout 600 python $R/bench.py --no-cpu-baseline --no-wall --no-fixed-rho-leg --steps 5 --warmup 2"
echo "== variant: stage scale ${CPG_STREAM_STAGE_SCALE:-default}"
for st in 2 3; do echo "== debug_stage=$st"; $B --debug-stage $st 2>&1 | tail -1 | python -c "$P"; done
for mi in 51 52 75; do echo "== max_iter=$mi"; $B --max-iter $mi 2>&1 | tail -1 | python -c "$P"; done
echo "== done"
