#!/bin/bash
# A/B on ONE box: the squad executor (solve program in registers, placement 3 / automatic) against the LDS-program executor
# (placement 1) of the same family library.   gpurun --timeout 900 -- 'CPG_OUT=r6_s2 bash scripts/gpu_ab_squad.sh'
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/${CPG_OUT:-r6_s2}; mkdir -p $OUT; export TMPDIR=/tmp
P="import sys,json; d=json.loads(sys.stdin.read()); ph=d.get('phases') or {}; print(round(d['value']), round(d['ms_per_step'],2), d['config'].get('mean_iter'), d['config'].get('solved'), {k:(v['kernel'], round(v['ms'],2), v['instances']) for k,v in ph.items()}, d.get('check'))"
B="timeout 600 python $R/bench.py --no-cpu-baseline --no-wall --no-fixed-rho-leg"
for rep in 1 2; do
  echo "== squad $rep"; $B --placement 3 $([ $rep = 1 ] && echo --check) 2>&1 | tail -1 | tee $OUT/bench_config2_squad_$rep.json | python -c "$P"
  echo "== lds program $rep"; $B --placement 1 $([ $rep = 1 ] && echo --check) 2>&1 | tail -1 | tee $OUT/bench_config2_lds_$rep.json | python -c "$P"
done
echo "== squad, fixed-rho fork"; $B --placement 3 --fixed-rho 2>&1 | tail -1 | tee $OUT/bench_config2_squad_fixed_rho.json | python -c "$P"
echo "== lds, fixed-rho fork"; $B --placement 1 --fixed-rho 2>&1 | tail -1 | tee $OUT/bench_config2_lds_fixed_rho.json | python -c "$P"
if [[ " ${CPG_SKIP:-} " != *" pmc "* ]]; then
prof() { local tag=$1; shift; ( cd /tmp && timeout 400 rocprofv3 "$@" > $R/$OUT/$tag.log 2>&1 ); }
C="python $R/bench.py --no-cpu-baseline --no-wall --no-fixed-rho-leg --placement 3"
prof pmc_a --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $R/$OUT/pmc_a -o pmc -- $C --steps 2 --warmup 1
prof pmc_b --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_ACTIVE_INST_VALU -d $R/$OUT/pmc_b -o pmc -- $C --steps 2 --warmup 1
for d in a b; do f=$(find $OUT/pmc_$d -name "*.db" | head -1); [ -n "$f" ] && python scripts/rocpd_pmc.py $f '%_kernel%'; done | tee $OUT/pmc_config2_squad.txt
rm -rf $OUT/pmc_a $OUT/pmc_b
fi
echo "== done"
