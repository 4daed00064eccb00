set -o pipefail
mkdir -p gpurun_out/r04p
timeout -k 10 1100 python3 -m pytest tests -x -q -m gpu > gpurun_out/r04p/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -3 gpurun_out/r04p/pytest_gpu.log
[ $rc -eq 0 ] || exit 1
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r04p/smoke.txt 2>&1; tail -1 gpurun_out/r04p/smoke.txt
python3 bench.py > gpurun_out/r04p/bench_default.json 2>gpurun_out/r04p/e1.txt || exit 1
python3 bench.py --workload sites > gpurun_out/r04p/bench_cfg4.json 2>gpurun_out/r04p/e2.txt || exit 1
python3 bench.py --workload network > gpurun_out/r04p/bench_cfg5_joingraph.json 2>gpurun_out/r04p/e3.txt || exit 1
python3 bench.py --workload network --graph bethe > gpurun_out/r04p/bench_cfg5_bethe.json 2>gpurun_out/r04p/e4.txt || exit 1
PGBP_BENCH_REHEARSAL=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29655 bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r04p/bench_default_2rank_rehearsal_one_gpu.json 2>gpurun_out/r04p/e5.txt; echo "rehearsal rc $?"
tail -c 200 gpurun_out/r04p/bench_default_2rank_rehearsal_one_gpu.json
