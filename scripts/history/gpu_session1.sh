#!/bin/bash
# One GPU session: parity tests, smoke, launch-geometry sweep, default bench, rocprof kernel trace.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
echo "== rocm-smi"; rocm-smi --showproductname 2>/dev/null | head -8
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee $OUT/smoke.log
echo "== pytest -m gpu"
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 | tee $OUT/pytest_gpu.log
echo "== sweep"
for lib in "" _w2 _w4; do
  for ipw in 1 2; do
    for waves in 4 8; do
      L=cvxpygen_amd/csrc/libcpg_hip${lib}.so
      echo "-- lib=$lib ipw=$ipw waves=$waves"
      timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --lib $L --ipw $ipw --waves $waves 2>&1 | tail -1 | tee $OUT/sweep${lib}_ipw${ipw}_w${waves}.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), 'inst/s', round(d['roofline']['kernel_ms'],2),'ms', d['config']['mean_iter'])" 2>&1
    done
  done
done
echo "== default bench (with cpu baseline + oracle check)"
timeout 600 python bench.py --check 2>&1 | tail -1 | tee $OUT/bench_default.json
echo "== rocprof kernel trace"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OLDPWD/$OUT/prof_r1 -o bench -- python $OLDPWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OLDPWD/$OUT/rocprof_bench.log 2>&1
cd $OLDPWD
find $OUT/prof_r1 -name "*stats*" | head; for f in $(find $OUT/prof_r1 -name "*kernel_stats.csv"); do head -5 $f; done
echo "== done"
