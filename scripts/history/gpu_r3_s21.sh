#!/bin/bash
# Round 3, session 21: non-temporal loads for the per-instance coefficient stream (configs 3 / all parameters / streaming
# instance executor), FETCH_SIZE of config 3 beside the timing
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/${CPG_OUT:-r3s21}; mkdir -p $OUT; export TMPDIR=/tmp
P="import sys,json; d=json.loads(sys.stdin.read()); ph=d.get('phases') or {}; print(round(d['value']), round(d['ms_per_step'],2), d['config'].get('mean_iter'), d['config'].get('solved'), {k:(round(v['ms'],2), v['instances']) for k,v in ph.items()})"
B="timeout 600 python $R/bench.py --no-cpu-baseline --no-wall --no-fixed-rho-leg --steps 3 --warmup 1"
echo "== config 3 portfolio 20k default"; $B --workload portfolio --batch 20000 2>&1 | tail -1 | tee $OUT/bench_config3_20k.json | python -c "$P"
echo "== config 3 portfolio 20k fixed rho"; $B --workload portfolio --batch 20000 --fixed-rho 2>&1 | tail -1 | tee $OUT/bench_config3_20k_fixed.json | python -c "$P"
echo "== mpc12 all params 20k"; $B --all-params --batch 20000 2>&1 | tail -1 | tee $OUT/bench_allparams.json | python -c "$P"
echo "== config 2, streaming instance executor"; $B --instance-executor stream 2>&1 | tail -1 | tee $OUT/bench_config2_stream.json | python -c "$P"
echo "== config 2 generic"; $B --generic 2>&1 | tail -1 | tee $OUT/bench_config2_generic.json | python -c "$P"
echo "== config 5 adjoint"; $B --adjoint 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d.get('adjoint'))"
cd /tmp
C="python $R/bench.py --no-cpu-baseline --no-wall --no-fixed-rho-leg --workload portfolio --batch 20000 --steps 2 --warmup 1"
timeout 400 rocprofv3 --pmc FETCH_SIZE -d $R/$OUT/pmc_f3 -o pmc -- $C > $R/$OUT/pmc_f3.log 2>&1
cd $R
f=$(find $OUT/pmc_f3 -name "*.db" | head -1); [ -n "$f" ] && python scripts/rocpd_pmc.py $f '%refactor%' | tee $OUT/pmc_config3.txt
rm -rf $OUT/pmc_f3
echo "== done"
