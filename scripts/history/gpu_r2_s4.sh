#!/bin/bash
# Round 2, GPU session 4: bank-aware slot numbering of the work vector (cvxpygen_amd/slot_layout.py)
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/r2s4; mkdir -p $OUT; export TMPDIR=/tmp
echo "== pytest subset"; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "mpc_vs_oracle or full_size or generated_family or infeasible or nonneg" 2>&1 | tail -3 | tee $OUT/pytest_gpu.txt
B="timeout 300 python $R/bench.py --no-cpu-baseline --no-wall"
echo "== default mpc12"; $B 2>&1 | tail -1 | tee $OUT/bench_default.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['plan'])"
echo "== mpc6"; $B --workload mpc6 2>&1 | tail -1 | tee $OUT/bench_mpc6.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['plan'])"
echo "== generic streamed"; $B --generic 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
echo "== check"; $B --steps 2 --warmup 1 --check 2>&1 | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['check'])"
cd /tmp
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $R/$OUT/pmc_a -o pmc -- $B --steps 2 --warmup 1 > $R/$OUT/pmc_a.log 2>&1
cd $R
for d in a; do f=$(find $OUT/pmc_$d -name "*.db" | head -1); [ -n "$f" ] && python scripts/rocpd_pmc.py $f | cut -c62-; done | tee $OUT/pmc.txt
rm -rf $OUT/pmc_a
echo "== done"
