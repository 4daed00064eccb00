set -o pipefail
mkdir -p gpurun_out/r04s
python3 bench.py > gpurun_out/r04s/bench_default.json 2>gpurun_out/r04s/e1.txt || exit 1
python3 bench.py --workload sites > gpurun_out/r04s/bench_cfg4.json 2>gpurun_out/r04s/e2.txt || exit 1
python3 bench.py --workload network > gpurun_out/r04s/bench_cfg5_joingraph.json 2>gpurun_out/r04s/e3.txt || exit 1
python3 bench.py --workload network --graph bethe > gpurun_out/r04s/bench_cfg5_bethe.json 2>gpurun_out/r04s/e4.txt || exit 1
PGBP_BENCH_REHEARSAL=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29655 bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r04s/bench_default_2rank_rehearsal_one_gpu.json 2>gpurun_out/r04s/e5.txt; echo "rehearsal rc $?"
tail -c 600 gpurun_out/r04s/bench_default_2rank_rehearsal_one_gpu.json
