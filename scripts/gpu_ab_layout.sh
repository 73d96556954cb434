#!/bin/bash
# A/B on ONE box: the slot numbering alone (round 5: CPG_ENTRY_ORDER=0, library under generated/variants/base) against the joint
# numbering + entry order of round 6 (the default library).   gpurun --timeout 900 -- 'bash scripts/gpu_ab_layout.sh'
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/${CPG_OUT:-r6_s1}; mkdir -p $OUT; export TMPDIR=/tmp
P="import sys,json; d=json.loads(sys.stdin.read()); ph=d.get('phases') or {}; print(round(d['value']), round(d['ms_per_step'],2), d['config'].get('mean_iter'), d['config'].get('solved'), {k:(round(v['ms'],2), v['instances']) for k,v in ph.items()}, d['config']['plan'].get('bank_conflict_cycles'), d.get('check'))"
B="timeout 600 python $R/bench.py --no-cpu-baseline --no-wall --no-fixed-rho-leg"
for rep in 1 2; do
  echo "== base (slot numbering only) $rep"; CPG_ENTRY_ORDER=0 $B --lib cvxpygen_amd/generated/variants/base/libcpg_mpc12.so 2>&1 | tail -1 | tee $OUT/bench_config2_base_$rep.json | python -c "$P"
  echo "== new (numbering + entry order) $rep"; $B $([ $rep = 1 ] && echo --check) 2>&1 | tail -1 | tee $OUT/bench_config2_new_$rep.json | python -c "$P"
done
prof() { local tag=$1; shift; ( cd /tmp && timeout 400 rocprofv3 "$@" > $R/$OUT/$tag.log 2>&1 ); }
C="python $R/bench.py --no-cpu-baseline --no-wall --no-fixed-rho-leg"
prof pmc_a --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $R/$OUT/pmc_a -o pmc -- $C --steps 2 --warmup 1
f=$(find $OUT/pmc_a -name "*.db" | head -1); [ -n "$f" ] && python scripts/rocpd_pmc.py $f '%_kernel%' | tee $OUT/pmc_config2_new.txt
rm -rf $OUT/pmc_a
echo "== done"
