set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r5s14
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "conic or adp or socp or clarabel" 2>&1 | tail -5
timeout 300 python scripts/gpu_probe_conic.py 2>&1 | tail -12 | tee gpurun_out/r5s14/conic_probe.txt
timeout 300 python bench.py --no-wall --no-cpu-baseline --workload adp 2>&1 | tail -1 | tee gpurun_out/r5s14/bench_config4.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config'].get('mean_iter'), d['config'].get('solved'))"
