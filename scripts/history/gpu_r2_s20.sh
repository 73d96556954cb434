#!/bin/bash
# Round 2, GPU session 20: time split of the per-instance factor kernel in the MID-ROUND build (commit 4f6d55e checked out
# as a git worktree under tmp_old/, its portfolio library built there) -- the counterpart of gpu_r2_s15.sh
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R/tmp_old
python -c "import __graft_entry__" 2>/dev/null
(cd oracle && make -s 2>/dev/null | tail -1)
P="import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2), d['config'].get('mean_iter'), d['config'].get('solved'))"
B="timeout 600 python bench.py --no-cpu-baseline --no-wall --workload portfolio --batch 20000 --steps 3 --warmup 1"
echo "== OLD default"; $B 2>&1 | tail -1 | python -c "$P"
echo "== OLD max_iter 1"; $B --max-iter 1 2>&1 | tail -1 | python -c "$P"
echo "== OLD 200 its, no checks"; $B --max-iter 200 --check-termination 1000 2>&1 | tail -1 | python -c "$P"
echo "== OLD 200 its, check every 25"; $B --max-iter 200 --eps 1e-12 2>&1 | tail -1 | python -c "$P"
echo "== OLD 400 its, no checks"; $B --max-iter 400 --check-termination 1000 2>&1 | tail -1 | python -c "$P"
