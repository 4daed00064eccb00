set -o pipefail
mkdir -p gpurun_out/r04d
timeout -k 10 900 python -m pytest tests/ -x -q -m gpu > gpurun_out/r04d/pytest_gpu.log 2>&1; echo "pytest rc $?" >> gpurun_out/r04d/pytest_gpu.log
tail -5 gpurun_out/r04d/pytest_gpu.log
timeout -k 10 600 python3 bench.py > gpurun_out/r04d/bench_default.json 2>gpurun_out/r04d/bench_default.err; echo "bench rc $?"
tail -3 gpurun_out/r04d/bench_default.err
