#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
for w in 0 2 4 8; do
echo "== bench adp waves=$w"; timeout 300 python bench.py --workload adp --steps 3 --warmup 1 --waves $w 2>&1 | tail -1 | tee $OUT/s10_adp_w$w.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), d['unit'], round(d['roofline']['kernel_ms'],2),'ms', d['config']['mean_iter'], d['config']['solved'])"
done
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof_s10 -- python $GRAFT_REPO_ROOT/bench.py --workload adp --steps 3 --warmup 1 > $GRAFT_REPO_ROOT/$OUT/s10_rocprof.log 2>&1
cd $GRAFT_REPO_ROOT; python scripts/rocpd_summary.py $OUT/prof_s10 2>&1 | tail -8 | tee $OUT/s10_kernel_stats.txt
echo "== done"
