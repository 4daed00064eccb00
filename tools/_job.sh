set -o pipefail
mkdir -p gpurun_out/r04q
timeout -k 10 1100 python -m pytest tests/ -q -m gpu -x > gpurun_out/r04q/pytest_gpu.log 2>&1; echo "pytest rc $?" >> gpurun_out/r04q/pytest_gpu.log
tail -4 gpurun_out/r04q/pytest_gpu.log
for s in 125 125 125 1000; do
  python3 bench.py --workload sites --sites $s --steps 30 --warmup 3 --no-cpu-baseline > gpurun_out/r04q/sites_$s.json 2>gpurun_out/r04q/err_$s.txt || exit 1
  python3 - gpurun_out/r04q/sites_$s.json $s <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); n=int(sys.argv[2])*8
print(sys.argv[2], "step ms", round(d['ms_per_step'],4), "cal ms", round(d['calibrate_only']['ms_per_step'],4), "ll/s", round(d['ll_evals_per_s']))
PY
done
