# round 6, exponential / power cones: GPU tests, probe, and the unchanged symmetric path beside them
set -x
mkdir -p gpurun_out/r6ns
timeout 900 python -m pytest tests/test_nonsym_cones.py -m gpu -q > gpurun_out/r6ns/pytest_nonsym.txt 2>&1; echo "rc=$?" >> gpurun_out/r6ns/pytest_nonsym.txt
tail -5 gpurun_out/r6ns/pytest_nonsym.txt
timeout 600 python scripts/gpu_probe_nonsym.py 100000 > gpurun_out/r6ns/probe_nonsym.txt 2>&1
tail -8 gpurun_out/r6ns/probe_nonsym.txt
timeout 1200 python -m pytest tests/test_conic.py tests/test_gpu_parity.py -m gpu -q -k "conic or squad or gpu" > gpurun_out/r6ns/pytest_conic_parity.txt 2>&1; echo "rc=$?" >> gpurun_out/r6ns/pytest_conic_parity.txt
tail -3 gpurun_out/r6ns/pytest_conic_parity.txt
timeout 300 python bench.py --steps 10 --warmup 2 > gpurun_out/r6ns/bench_default.txt 2>&1
tail -1 gpurun_out/r6ns/bench_default.txt | cut -c1-300
