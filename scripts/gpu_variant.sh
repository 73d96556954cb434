#!/bin/bash
# bench lines of family-library variants built under cvxpygen_amd/generated/variants/<tag> (scripts/build_variants.py or by hand):
#   gpurun --timeout 900 -- 'CPG_OUT=r6_s3 bash scripts/gpu_variant.sh "tag:ENV=val,ENV=val:bench args" ...'
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/${CPG_OUT:-r6_variant}; mkdir -p $OUT; export TMPDIR=/tmp
P="import sys,json; d=json.loads(sys.stdin.read()); ph=d.get('phases') or {}; print(round(d['value']), round(d['ms_per_step'],2), d['config'].get('mean_iter'), d['config'].get('solved'), {k:(v['kernel'], round(v['ms'],2), v['instances']) for k,v in ph.items()}, d.get('check'))"
for spec in "$@"; do
  tag="${spec%%:*}"; rest="${spec#*:}"; envs="${rest%%:*}"; bargs="${rest#*:}"
  lib=cvxpygen_amd/generated/variants/$tag/libcpg_mpc12.so
  [ "$tag" = default ] && lib=cvxpygen_amd/generated/mpc12/libcpg_mpc12.so
  echo "== $tag [$envs] $bargs"
  env $(echo $envs | tr ',' ' ') timeout 600 python $R/bench.py --no-cpu-baseline --no-wall --no-fixed-rho-leg --lib $lib $bargs 2>&1 | tail -1 | tee $OUT/bench_${tag}_$(echo $bargs | tr -d ' -').json | python -c "$P"
done
echo "== done"
