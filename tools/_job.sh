set -o pipefail
mkdir -p gpurun_out/r04ad
PGBP_LIB=$PWD/build/libpgbp_fwdcheck.so timeout -k 10 900 python3 tests/fuzz_gpu_vs_c_oracle_networks.py 60 17 > gpurun_out/r04ad/fuzz_chk.log 2>&1; echo "fuzz with check rc $?"; grep -c "FWD MISMATCH" gpurun_out/r04ad/fuzz_chk.log; tail -2 gpurun_out/r04ad/fuzz_chk.log
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "network_differential_fuzz or level3 or loopy or cfg5 or failure" > gpurun_out/r04ad/pytest.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -5 gpurun_out/r04ad/pytest.log
[ $rc -eq 0 ] || exit 1
PGBP_LIB=$PWD/build/libpgbp_pstamp.so timeout -k 10 300 python3 tools/stamp_pair.py joingraph > gpurun_out/r04ad/stamps_jg.txt 2>gpurun_out/r04ad/err.txt; echo rc $?; cat gpurun_out/r04ad/stamps_jg.txt; tail -3 gpurun_out/r04ad/err.txt
B=$PWD/build/libpgbp_prev.so
bash tools/sweep_env.sh r04ad/jg "--workload network --no-cpu-baseline" "PGBP_LIB=$B" "-" "PGBP_LIB=$B" "-" || exit 1
bash tools/sweep_env.sh r04ad/be "--workload network --graph bethe --no-cpu-baseline" "PGBP_LIB=$B" "-" "PGBP_LIB=$B" "-" || exit 1
