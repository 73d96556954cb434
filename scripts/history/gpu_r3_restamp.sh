#!/bin/bash
# Round 3, very last GPU seconds: the guard test on the re-built libraries (plan lock), FETCH / WRITE passes of the
# headline step and its stamped traffic record on the final sources
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/${CPG_OUT:-r3restamp}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 60 python -m pytest tests/test_gpu_surface.py -m gpu -q -k "generated_instance_executor" 2>&1 | tail -2 | tee $OUT/pytest_guard.txt
cd /tmp
C="python $R/bench.py --no-cpu-baseline --no-wall --no-fixed-rho-leg --steps 2 --warmup 1"
timeout 100 rocprofv3 --pmc FETCH_SIZE -d $R/$OUT/pmc_f -o pmc -- $C > $R/$OUT/pmc_f.log 2>&1
timeout 100 rocprofv3 --pmc WRITE_SIZE -d $R/$OUT/pmc_w -o pmc -- $C > $R/$OUT/pmc_w.log 2>&1
cd $R
for d in f w; do f=$(find $OUT/pmc_$d -name "*.db" | head -1); [ -n "$f" ] && python scripts/rocpd_pmc.py $f; done | tee $OUT/pmc_config2.txt
python scripts/record_traffic.py mpc12 100000 $OUT/pmc_config2.txt "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on python bench.py --no-cpu-baseline --no-wall --no-fixed-rho-leg --steps 2 --warmup 1, session $OUT" && cp profiles/r3_hbm_traffic.json $OUT/r3_hbm_traffic.json
rm -rf $OUT/pmc_f $OUT/pmc_w
tail -1 $OUT/pmc_f.log | cut -c1-300
echo "== done"
