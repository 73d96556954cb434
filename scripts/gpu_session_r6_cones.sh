# round 6: exponential / power / PSD cones on the GPU -- tests, probe, the symmetric path beside them
set -x
mkdir -p gpurun_out/r6cones
timeout 1200 python -m pytest tests/test_nonsym_cones.py tests/test_psd_cones.py -m gpu -q > gpurun_out/r6cones/pytest_cones.txt 2>&1; echo "rc=$?" >> gpurun_out/r6cones/pytest_cones.txt
tail -5 gpurun_out/r6cones/pytest_cones.txt
timeout 900 python scripts/gpu_probe_nonsym.py 100000 > gpurun_out/r6cones/probe_cones.txt 2>&1
tail -12 gpurun_out/r6cones/probe_cones.txt
timeout 300 python bench.py --workload adp --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r6cones/bench_adp.txt 2>&1
tail -1 gpurun_out/r6cones/bench_adp.txt | cut -c1-200
