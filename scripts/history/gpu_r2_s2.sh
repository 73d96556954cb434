#!/bin/bash
# Round 2, GPU session 2: termination-test products with literal chunk tables (loads of a whole chunk in flight),
# plain vs compressed program; PCIe-inclusive pipelined rate; OSQP >= 1.0 build options on config 2.
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/r2s2; mkdir -p $OUT; export TMPDIR=/tmp
echo "== pytest subset"; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "mpc_vs_oracle or full_size or generated_family or infeasible" 2>&1 | tail -4 | tee $OUT/pytest_gpu.txt
B="timeout 300 python bench.py --no-cpu-baseline --no-wall --steps 5 --warmup 2"
for v in plain plain_nb25 comp_nb25 comp_nb6; do
  echo "== exp $v"; $B --lib cvxpygen_amd/generated/exp/libcpg_mpc12_$v.so 2>&1 | tail -1 | tee $OUT/bench_$v.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['mean_iter'])"
done
echo "== default (comp, 12 waves) with wall"; timeout 300 python bench.py --no-cpu-baseline --steps 5 --warmup 2 2>&1 | tail -1 | tee $OUT/bench_default.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('wall_pcie'))"
echo "== default --waves 8"; $B --waves 8 2>&1 | tail -1 | tee $OUT/bench_default_w8.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
echo "== knock-out: one check"; $B --max-iter 100 --check-termination 100 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
echo "== knock-out: four checks"; $B --max-iter 100 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
echo "== osqp1 mode (adaptive rho + dualgap), config 2"; timeout 600 python bench.py --no-cpu-baseline --no-wall --steps 2 --warmup 1 --osqp1 2>&1 | tail -1 | tee $OUT/bench_osqp1.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['mean_iter'], d['config']['solved'])"
echo "== check vs oracle"; timeout 300 python bench.py --no-cpu-baseline --no-wall --steps 2 --warmup 1 --check 2>&1 | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['check'])" | tee $OUT/check.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof -o trace -- python $R/bench.py --no-cpu-baseline --no-wall > $R/$OUT/rocprof.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $R/$OUT/pmc_f -o pmc -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-wall > $R/$OUT/pmc_f.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $R/$OUT/pmc_w -o pmc -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-wall > $R/$OUT/pmc_w.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $R/$OUT/pmc_f8 -o pmc -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-wall --lib cvxpygen_amd/generated/exp/libcpg_mpc12_plain.so > $R/$OUT/pmc_f8.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $R/$OUT/pmc_w8 -o pmc -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-wall --lib cvxpygen_amd/generated/exp/libcpg_mpc12_plain.so > $R/$OUT/pmc_w8.log 2>&1
cd $R
f=$(find $OUT/prof -name "*.db" | head -1); [ -n "$f" ] && python scripts/rocpd_summary.py $f | tee $OUT/kernel_stats.txt
for d in f w f8 w8; do f=$(find $OUT/pmc_$d -name "*.db" | head -1); [ -n "$f" ] && python scripts/rocpd_pmc.py $f | cut -c1-130; done | tee $OUT/pmc.txt
rm -rf $OUT/prof $OUT/pmc_f $OUT/pmc_w $OUT/pmc_f8 $OUT/pmc_w8
echo "== done"
