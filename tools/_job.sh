set -o pipefail
mkdir -p gpurun_out/r04ag
C=$PWD/build/libpgbp_new.so
PGBP_LIB=$C timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "network_differential_fuzz or level3 or loopy or cfg5 or failure" > gpurun_out/r04ag/pytest.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -3 gpurun_out/r04ag/pytest.log
[ $rc -eq 0 ] || exit 1
bash tools/sweep_env.sh r04ag/jg "--workload network --no-cpu-baseline" "-" "PGBP_LIB=$C" "-" "PGBP_LIB=$C" || exit 1
bash tools/sweep_env.sh r04ag/be "--workload network --graph bethe --no-cpu-baseline" "-" "PGBP_LIB=$C" || exit 1
