#!/bin/bash
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/r2s6; mkdir -p $OUT; export TMPDIR=/tmp
B="timeout 600 python $R/bench.py --no-cpu-baseline --no-wall --workload portfolio --batch 20000 --steps 3 --warmup 1"
P="import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['mean_iter'], d['config']['solved'])"
echo "== family lib, table-driven substitution"; $B --lib $R/cvxpygen_amd/generated/exp_nosub/libcpg_portfolio.so 2>&1 | tail -1 | tee $OUT/bench_nosub.json | python -c "$P"
echo "== family lib, generated w2p16"; $B --lib $R/cvxpygen_amd/generated/exp_w2p16/libcpg_portfolio.so 2>&1 | tail -1 | python -c "$P"
echo "== knock-out: max_iter 25 (setup + 25 iterations)"; $B --lib $R/cvxpygen_amd/generated/exp_nosub/libcpg_portfolio.so --max-iter 25 2>&1 | tail -1 | python -c "$P"
echo "== knock-out: max_iter 25 generated"; $B --lib $R/cvxpygen_amd/generated/exp_w2p16/libcpg_portfolio.so --max-iter 25 2>&1 | tail -1 | python -c "$P"
echo "== knock-out: max_iter 125 nosub"; $B --lib $R/cvxpygen_amd/generated/exp_nosub/libcpg_portfolio.so --max-iter 125 2>&1 | tail -1 | python -c "$P"
echo "== knock-out: max_iter 125 generated"; $B --lib $R/cvxpygen_amd/generated/exp_w2p16/libcpg_portfolio.so --max-iter 125 2>&1 | tail -1 | python -c "$P"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof -o trace -- $B --lib $R/cvxpygen_amd/generated/exp_w2p16/libcpg_portfolio.so > $R/$OUT/rocprof.log 2>&1
cd $R
f=$(find $OUT/prof -name "*.db" | head -1); [ -n "$f" ] && python scripts/rocpd_summary.py $f | tee $OUT/kernel_stats.txt
rm -rf $OUT/prof
