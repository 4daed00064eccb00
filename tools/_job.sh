set -o pipefail
mkdir -p gpurun_out/r04y
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fused_into_the_postorder or device_factor_fill" > gpurun_out/r04y/pytest.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -15 gpurun_out/r04y/pytest.log
[ $rc -eq 0 ] || exit 1
bash tools/sweep_env.sh r04y/ll "--no-cpu-baseline --no-network-block --no-sites-block" "PGBP_TUNING=fused_fill=0" "-" "PGBP_TUNING=fused_fill=2" "PGBP_TUNING=fused_fill=0" "-" || exit 1
python3 - <<'PY'
import json
for i in (1,2,3,4,5):
    d=json.loads(open(f'gpurun_out/r04y/ll_{i}.json').read().strip().splitlines()[-1])
    print(i, d['ms_per_step'], d['ll_evals_per_s'], d['ll_eval']['roofline']['ms_per_eval'], d['ll_evals_per_s_batched']['value'], d['loglik'])
PY
