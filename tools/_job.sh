set -o pipefail
mkdir -p gpurun_out/r04ap
timeout -k 10 1100 python3 -m pytest tests -x -q -m gpu > gpurun_out/r04ap/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -3 gpurun_out/r04ap/pytest_gpu.log
[ $rc -eq 0 ] || exit 1
for v in 1 2 3; do
  python3 bench.py --no-cpu-baseline --no-network-block --no-sites-block --no-alt-reading > gpurun_out/r04ap/ll_$v.json 2>gpurun_out/r04ap/e_$v.txt || exit 1
  python3 -c "
import json
d=json.loads(open('gpurun_out/r04ap/ll_$v.json').read().strip().splitlines()[-1])
print('run $v', 'll/s', round(d['ll_evals_per_s'],1), 'ms/eval', round(d['ll_eval']['roofline']['ms_per_eval'],4), 'batched', round(d['ll_evals_per_s_batched']['value'],1), 'cal', round(d['ms_per_step'],4), d['loglik'], d['loglik_rel_err_vs_pruning'])
"
done
