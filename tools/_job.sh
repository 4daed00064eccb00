set -o pipefail
mkdir -p gpurun_out/r04l
export TMPDIR=/tmp
timeout -k 10 1100 python -m pytest tests/ -q -m gpu > gpurun_out/r04l/pytest_gpu.log 2>&1; echo "pytest rc $?" >> gpurun_out/r04l/pytest_gpu.log
tail -6 gpurun_out/r04l/pytest_gpu.log
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p5 -- python3 $GRAFT_REPO_ROOT/bench.py --workload sites --sites 125 --steps 5 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r04l/bench_cfg4_125_prof.json 2>$GRAFT_REPO_ROOT/gpurun_out/r04l/e2.txt; f=$(find /tmp/p5 -name "*kernel_stats.csv" | head -1); cp $f $GRAFT_REPO_ROOT/gpurun_out/r04l/cfg4_125_kernel_stats.csv
cd $GRAFT_REPO_ROOT
cut -c1-150 gpurun_out/r04l/cfg4_125_kernel_stats.csv | head -8
