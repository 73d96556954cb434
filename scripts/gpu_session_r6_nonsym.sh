set -x
mkdir -p gpurun_out/r6ns
timeout 900 python -m pytest tests/test_nonsym_cones.py -m gpu -x -q > gpurun_out/r6ns/pytest_nonsym.txt 2>&1; echo "rc=$?" >> gpurun_out/r6ns/pytest_nonsym.txt
tail -5 gpurun_out/r6ns/pytest_nonsym.txt
timeout 600 python scripts/gpu_probe_nonsym.py 100000 > gpurun_out/r6ns/probe_nonsym.txt 2>&1
cat gpurun_out/r6ns/probe_nonsym.txt | tail -8
timeout 1200 python -m pytest tests/test_conic.py -m gpu -x -q > gpurun_out/r6ns/pytest_conic.txt 2>&1; echo "rc=$?" >> gpurun_out/r6ns/pytest_conic.txt
tail -3 gpurun_out/r6ns/pytest_conic.txt
timeout 300 python bench.py --workload adp --steps 10 --warmup 2 > gpurun_out/r6ns/bench_adp.txt 2>&1
tail -2 gpurun_out/r6ns/bench_adp.txt
