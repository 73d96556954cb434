#!/bin/bash
# Round 2, GPU session 27: what the packer charges for a stage of the segmented reduction (fewer stages, more steps)
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
P="import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2), d['config'].get('mean_iter'), d['config'].get('solved'), d.get('check'))"
B="timeout 600 python $R/bench.py --no-cpu-baseline --no-wall --check"
V=$R/cvxpygen_amd/generated/variants
echo "== 1.0"; $B 2>&1 | tail -1 | python -c "$P"
for g in 2.5:sc25 6:sc6 10:sc10; do c=${g%%:*}; v=${g##*:}; echo "== $c"; CPG_STAGE_COST=$c $B --lib $V/$v/libcpg_mpc12.so 2>&1 | tail -1 | python -c "$P"; done
