set -o pipefail
mkdir -p gpurun_out/r04w
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "network_differential_fuzz or level3 or loopy or cfg5" > gpurun_out/r04w/pytest.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -5 gpurun_out/r04w/pytest.log
[ $rc -eq 0 ] || exit 1
PGBP_LIB=$PWD/build/libpgbp_pstamp.so timeout -k 10 300 python3 tools/stamp_pair.py joingraph > gpurun_out/r04w/stamps_jg.txt 2>gpurun_out/r04w/err.txt; echo rc $?; cat gpurun_out/r04w/stamps_jg.txt; tail -3 gpurun_out/r04w/err.txt
B=$PWD/build/libpgbp_base.so
bash tools/sweep_env.sh r04w/jg "--workload network --no-cpu-baseline" "PGBP_LIB=$B" "-" "PGBP_TUNING=pair=0" "-" || exit 1
bash tools/sweep_env.sh r04w/be "--workload network --graph bethe --no-cpu-baseline" "PGBP_LIB=$B" "-" || exit 1
