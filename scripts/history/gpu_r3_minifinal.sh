#!/bin/bash
# Round 3, last session (little GPU time left): the shared-mode plan with its own stage scale (165-step per-instance program
# on MPC 12/4/10) -- GPU tests that touch the generated instance executor, headline bench, kernel stats, FETCH / WRITE
# passes and the stamped traffic record, most important first
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/${CPG_OUT:-r3mini}; mkdir -p $OUT; export TMPDIR=/tmp
echo "== gpu tests"; timeout 400 python -m pytest tests/test_gpu_surface.py tests/test_gpu_parity.py -m gpu -q -x -k "hybrid or generated or family or sequence or drop_in or forward_backward" 2>&1 | tail -3 | tee $OUT/pytest_gpu_subset.txt
P="import sys,json; d=json.loads(sys.stdin.read()); ph=d.get('phases') or {}; print(round(d['value']), round(d['ms_per_step'],2), d['config'].get('mean_iter'), d['config'].get('solved'), {k:(round(v['ms'],2), v['instances']) for k,v in ph.items()}, d.get('check'))"
echo "== config 2"; timeout 300 python bench.py --check --no-wall 2>&1 | tail -1 | tee $OUT/bench_config2.json | python -c "$P"
B="timeout 300 python $R/bench.py --no-cpu-baseline --no-wall --no-fixed-rho-leg"
echo "== mpc6"; $B --workload mpc6 2>&1 | tail -1 | tee $OUT/bench_mpc6.json | python -c "$P"
echo "== tight"; $B --eps 1e-6 2>&1 | tail -1 | tee $OUT/bench_config2_tight.json | python -c "$P"
cd /tmp
C="python $R/bench.py --no-cpu-baseline --no-wall --no-fixed-rho-leg --steps 2 --warmup 1"
timeout 200 rocprofv3 --pmc FETCH_SIZE -d $R/$OUT/pmc_f -o pmc -- $C > $R/$OUT/pmc_f.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE -d $R/$OUT/pmc_w -o pmc -- $C > $R/$OUT/pmc_w.log 2>&1
cd $R
for d in f w; do f=$(find $OUT/pmc_$d -name "*.db" | head -1); [ -n "$f" ] && python scripts/rocpd_pmc.py $f; done | tee $OUT/pmc_config2.txt
python scripts/record_traffic.py mpc12 100000 $OUT/pmc_config2.txt "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on python bench.py --no-cpu-baseline --no-wall --no-fixed-rho-leg --steps 2 --warmup 1, session $OUT" && cp profiles/r3_hbm_traffic.json $OUT/r3_hbm_traffic.json
rm -rf $OUT/pmc_f $OUT/pmc_w
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof -o trace -- python $R/bench.py --no-cpu-baseline --no-wall --no-fixed-rho-leg > $R/$OUT/rocprof.log 2>&1
cd $R; f=$(find $OUT/prof -name "*.db" | head -1); [ -n "$f" ] && python scripts/rocpd_summary.py $f | tee $OUT/kernel_stats_config2.txt; rm -rf $OUT/prof
cd /tmp
C3="python $R/bench.py --no-cpu-baseline --no-wall --no-fixed-rho-leg --workload portfolio --batch 20000 --steps 2 --warmup 1"
timeout 200 rocprofv3 --pmc FETCH_SIZE -d $R/$OUT/pmc_f3 -o pmc -- $C3 > $R/$OUT/pmc_f3.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE -d $R/$OUT/pmc_w3 -o pmc -- $C3 > $R/$OUT/pmc_w3.log 2>&1
cd $R
for d in f3 w3; do f=$(find $OUT/pmc_$d -name "*.db" | head -1); [ -n "$f" ] && python scripts/rocpd_pmc.py $f '%refactor%'; done | tee $OUT/pmc_config3.txt
python scripts/record_traffic.py portfolio 20000 $OUT/pmc_config3.txt "same command with --workload portfolio --batch 20000, session $OUT" && cp profiles/r3_hbm_traffic.json $OUT/r3_hbm_traffic.json
rm -rf $OUT/pmc_f3 $OUT/pmc_w3
echo "== done"
