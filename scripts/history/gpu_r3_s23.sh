#!/bin/bash
# Round 3, session 23: shared-factor kernel with two instances per wavefront (G = 2, four wavefronts per CU) against one
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/${CPG_OUT:-r3s23}; mkdir -p $OUT; export TMPDIR=/tmp
P="import sys,json; d=json.loads(sys.stdin.read()); ph=d.get('phases') or {}; print(round(d['value']), round(d['ms_per_step'],2), d['config'].get('mean_iter'), {k:(round(v['ms'],2), v['instances']) for k,v in ph.items()})"
B="timeout 600 python $R/bench.py --no-cpu-baseline --no-wall --no-fixed-rho-leg --steps 5 --warmup 2"
echo "== fixed rho, G=1"; $B --fixed-rho 2>&1 | tail -1 | python -c "$P"
echo "== fixed rho, G=2"; $B --fixed-rho --ipw 2 2>&1 | tail -1 | python -c "$P"
echo "== fixed rho, G=2, 4 waves"; $B --fixed-rho --ipw 2 --waves 4 2>&1 | tail -1 | python -c "$P"
echo "== default, G=2"; $B --ipw 2 2>&1 | tail -1 | python -c "$P"
echo "== done"
