cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 300 python bench.py --workload portfolio --batch 20000 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('portfolio', round(d['value']), d['ms_per_step'])"
timeout 300 python bench.py --all-params --batch 20000 --steps 3 --warmup 1 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mpc12 all', round(d['value']), d['ms_per_step'])"
timeout 300 python bench.py --adjoint --batch 20000 --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('adjoint', d['adjoint']['kernel_ms'])"
timeout 900 python -m pytest tests -q -x -m gpu 2>&1 | tail -2
