#!/bin/bash
# Round 3, session 19: conic family library compiled for three wavefronts per SIMD (168 VGPRs, no scratch) against four
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/${CPG_OUT:-r3s19}; mkdir -p $OUT; export TMPDIR=/tmp
P="import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3), d['config'].get('mean_iter'))"
B="timeout 600 python $R/bench.py --no-cpu-baseline --no-wall --steps 5 --warmup 2 --workload adp"
echo "== 4 waves per SIMD (default build)"; $B 2>&1 | tail -1 | tee $OUT/bench_w4.json | python -c "$P"
echo "== 3 waves per SIMD"; $B --lib $R/cvxpygen_amd/generated/adp_w3/libcpg_adp.so 2>&1 | tail -1 | tee $OUT/bench_w3.json | python -c "$P"
for w in 4 5 6; do echo "== 3 waves per SIMD, workgroups of $w"; $B --lib $R/cvxpygen_amd/generated/adp_w3/libcpg_adp.so --waves $w 2>&1 | tail -1 | python -c "$P"; done
echo "== done"
