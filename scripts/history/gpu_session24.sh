#!/bin/bash
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $R/$OUT/pmc_s24f -o pmc -- python $R/bench.py --workload portfolio --batch 20000 --steps 1 --warmup 1 --no-cpu-baseline > $R/$OUT/s24f.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $R/$OUT/pmc_s24w -o pmc -- python $R/bench.py --workload portfolio --batch 20000 --steps 1 --warmup 1 --no-cpu-baseline > $R/$OUT/s24w.log 2>&1
cd $R
for d in f w; do f=$(find $OUT/pmc_s24$d -name "*.db" | head -1); python scripts/rocpd_pmc.py $f '%refactor%' | cut -c62-; done
echo "== done"
