set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r5s18
timeout 300 python -m pytest tests -m gpu -q -k "conic or adp or socp or clarabel or ecos" 2>&1 | tail -3
for i in 1 2; do timeout 300 python bench.py --no-wall --no-cpu-baseline --workload adp 2>&1 | tail -1 | tee gpurun_out/r5s18/bench_config4_$i.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config'].get('mean_iter'), d['config'].get('solved'))"; done
