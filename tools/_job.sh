set -o pipefail
mkdir -p gpurun_out/r04ac
P=$PWD/build/libpgbp_prev.so
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "network_differential_fuzz or level3 or loopy or cfg5" > gpurun_out/r04ac/pytest.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -3 gpurun_out/r04ac/pytest.log
[ $rc -eq 0 ] || exit 1
bash tools/sweep_env.sh r04ac/jg "--workload network --no-cpu-baseline" "PGBP_LIB=$P" "-" "PGBP_LIB=$P" "-" || exit 1
bash tools/sweep_env.sh r04ac/be "--workload network --graph bethe --no-cpu-baseline" "PGBP_LIB=$P" "-" || exit 1
