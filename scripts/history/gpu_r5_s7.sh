#!/bin/bash
# round 5, session 7: the integrated default path -- family library of MPC 12/4/10 with the team kernel chosen automatically:
# the team's GPU tests, the all-parameters bench line through bench.py's own library selection, smoke
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/r5s7; mkdir -p $OUT; export TMPDIR=/tmp
echo "== team GPU tests"; timeout 600 python -m pytest tests/test_team.py -m gpu -q -x 2>&1 | tail -5 | tee $OUT/pytest_team.txt
echo "== bench all parameters (default library)"; timeout 300 python bench.py --no-cpu-baseline --no-wall --no-fixed-rho-leg --all-params --batch 20000 --steps 3 --warmup 1 2>&1 | tail -1 | tee $OUT/bench_allparams.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), d['roofline']['kernel'], d['config'].get('mean_iter'), d['config'].get('solved'), d['config']['plan'])"
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $OUT/smoke.txt
echo "== done"
