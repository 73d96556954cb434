#!/bin/bash
# Round 4, step 9: what the termination test of the resident kernel costs with and without its three products (debug_stage 21)
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd $R
OUT=gpurun_out/${CPG_OUT:-r4s9}; mkdir -p $OUT; export TMPDIR=/tmp
echo "== probes, full test"; CPG_PROBE_STAGE=20 timeout 300 python scripts/gpu_probe_resident.py 20000 2>&1 | tail -9 | tee $OUT/probe_stage20.txt
echo "== probes, test without products"; CPG_PROBE_STAGE=21 timeout 300 python scripts/gpu_probe_resident.py 20000 2>&1 | tail -9 | tee $OUT/probe_stage21.txt
echo "== done"
