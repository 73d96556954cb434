set -x
mkdir -p gpurun_out/r6ns
timeout 900 python -m pytest tests/test_nonsym_cones.py -m gpu -q > gpurun_out/r6ns/pytest_nonsym.txt 2>&1; echo "rc=$?" >> gpurun_out/r6ns/pytest_nonsym.txt
tail -5 gpurun_out/r6ns/pytest_nonsym.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r6ns/smoke.txt 2>&1; echo "rc=$?" >> gpurun_out/r6ns/smoke.txt
tail -5 gpurun_out/r6ns/smoke.txt
