set -o pipefail
mkdir -p gpurun_out/r04x
timeout -k 10 1100 python3 -m pytest tests -x -q -m gpu > gpurun_out/r04x/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -5 gpurun_out/r04x/pytest_gpu.log
